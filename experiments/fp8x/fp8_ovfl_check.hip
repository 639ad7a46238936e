// Checks the assumptions the fp8 cross-term mode would rest on: v_cvt_pk_fp8_f32 produces OCP e4m3 bytes (RNE), and
// mfma_scale_f32_32x32x64_f8f6f4 with format 0/0 computes D[i][j] = 2^(sa-127) 2^(sb-127) sum_k A[i][k] B[k][j] when lane l
// holds row/column l & 31 and the 32 consecutive k = 32 * (l >> 5) ... in its 8 operand registers (byte order = k order).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* __restrict__ a, const float* __restrict__ b, unsigned char* __restrict__ a8, unsigned char* __restrict__ b8,
                  float* __restrict__ d) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // MODE.FP16_OVFL = 1: does it make the fp8 conversion saturate?
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 fa, fb;
    for (int v = 0; v < 8; v++) {
        int wa = 0, wb = 0;
        const float* pa = a + r * 64 + h * 32 + v * 4;  // A[r][k]
        const float* pb = b + r * 64 + h * 32 + v * 4;  // B^T[r][k]  (column r of B)
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(pa[0], pa[1], wa, false);
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(pa[2], pa[3], wa, true);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(pb[0], pb[1], wb, false);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(pb[2], pb[3], wb, true);
        fa[v] = wa; fb[v] = wb;
        reinterpret_cast<int*>(a8)[(r * 64 + h * 32) / 4 + v] = wa;
        reinterpret_cast<int*>(b8)[(r * 64 + h * 32) / 4 + v] = wb;
    }
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    // weights-as-rows convention of conv_igemm: first operand = B fragment, second = A fragment -> D rows = B's columns j, D cols = A's rows i
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, acc, 0, 0, 0, 126, 0, 127);
    for (int e = 0; e < 16; e++) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = lane & 31;  // row = j (B column), col = i (A row)
        d[col * 32 + row] = acc[e];                                                // d[i][j]
    }
}

static float dec(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 15 && m == 7) x = NAN;
    else if (e == 0) x = std::ldexp((float)m, -9);
    else x = std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

int main() {
    std::vector<float> a(32 * 64), b(32 * 64);
    srand(3);
    for (auto& x : a) x = ((rand() % 20001) - 10000) * (1.0f / 10000.0f) * (rand() % 4 == 0 ? 200.f : 3.f);
    for (auto& x : b) x = ((rand() % 20001) - 10000) * (1.0f / 10000.0f) * (rand() % 5 == 0 ? 0.01f : 1.f);
    a[5] = 447.0f; a[6] = 460.0f; a[7] = 1000.0f; a[8] = -1e-4f; a[9] = 0.0019f;  // near max, beyond max, subnormals
    float *da, *db, *dd; unsigned char *da8, *db8;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dd, 32 * 32 * 4); hipMalloc(&da8, 32 * 64); hipMalloc(&db8, 32 * 64);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, da8, db8, dd);
    std::vector<float> d(32 * 32); std::vector<unsigned char> a8(32 * 64), b8(32 * 64);
    hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(a8.data(), da8, a8.size(), hipMemcpyDeviceToHost); hipMemcpy(b8.data(), db8, b8.size(), hipMemcpyDeviceToHost);
    printf("bytes of 447, 460, 1000, -1e-4, 0.0019: %02x %02x %02x %02x %02x -> %g %g %g %g %g\n", a8[5], a8[6], a8[7], a8[8], a8[9], dec(a8[5]), dec(a8[6]), dec(a8[7]), dec(a8[8]), dec(a8[9]));
    double worst = 0, worst_q = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            double s = 0;
            for (int kk = 0; kk < 64; kk++) {
                const float x = dec(a8[i * 64 + kk]), y = dec(b8[j * 64 + kk]);
                if (std::isnan(x) || std::isnan(y)) continue;
                s += (double)x * y;
            }
            s *= 0.5;
            if (i == 0 && (a8[7] == 0x7f || a8[7] == 0xff)) continue;  // row 0 holds the NaN probe
            worst = fmax(worst, fabs(s - d[i * 32 + j]) / (fabs(s) + 1e-3));
        }
    for (int i = 1; i < 32 * 64; i++) { const float q = dec(a8[i]); if (!std::isnan(q) && fabs(a[i]) < 440 && fabs(a[i]) > 0.02) worst_q = fmax(worst_q, fabs(q - a[i]) / fabs(a[i])); }
    printf("mfma vs decoded-bytes dot product: worst rel diff %.3g; e4m3 quantisation worst rel err %.3g (expect <= 0.0625)\n", worst, worst_q);
    for (int i = 1; i < 4; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0, s1 = 0, s2 = 0;
            for (int kk = 0; kk < 64; kk++) { const double p = (double)dec(a8[i * 64 + kk]) * dec(b8[j * 64 + kk]); s += p; (kk < 32 ? s1 : s2) += p; }
            printf("i %d j %d: d[i][j] %g d[j][i] %g  full %g  first half %g second half %g\n", i, j, d[i * 32 + j], d[j * 32 + i], s, s1, s2);
        }
    return 0;
}
