// Round 4: the cross terms of INFUR_DTYPE_F32_SPLIT_FP8 moved from e4m3 to bf8 (OCP e5m2).  This probe pins what the kernel
// rests on: v_cvt_pk_bf8_f32 produces e5m2 bytes with round-to-nearest-even, what it does beyond +-57344 with and without
// MODE.FP16_OVFL (the kernel does NOT rely on it: it converts hi / 2), and v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 1
// computes D[i][j] = 2^(sa-127) 2^(sb-127) sum_k A[i][k] B[k][j] on e5m2 operands in the same lane layout as the e4m3 form
// (lane l: row / column l & 31, the 32 consecutive k = 32 * (l >> 5) ... in its 8 operand registers, byte order = k order).
//   hipcc --offload-arch=gfx950 -O2 experiments/fp8x/bf8_mfma_check.hip -o experiments/fp8x/bf8_mfma_check && experiments/fp8x/bf8_mfma_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* __restrict__ a, const float* __restrict__ b, unsigned char* __restrict__ a8, unsigned char* __restrict__ b8,
                  float* __restrict__ d, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // MODE.FP16_OVFL = 1
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 fa, fb;
    for (int v = 0; v < 8; v++) {
        int wa = 0, wb = 0;
        const float* pa = a + r * 64 + h * 32 + v * 4;
        const float* pb = b + r * 64 + h * 32 + v * 4;
        wa = __builtin_amdgcn_cvt_pk_bf8_f32(pa[0], pa[1], wa, false);
        wa = __builtin_amdgcn_cvt_pk_bf8_f32(pa[2], pa[3], wa, true);
        wb = __builtin_amdgcn_cvt_pk_bf8_f32(pb[0], pb[1], wb, false);
        wb = __builtin_amdgcn_cvt_pk_bf8_f32(pb[2], pb[3], wb, true);
        fa[v] = wa; fb[v] = wb;
        reinterpret_cast<int*>(a8)[(r * 64 + h * 32) / 4 + v] = wa;
        reinterpret_cast<int*>(b8)[(r * 64 + h * 32) / 4 + v] = wb;
    }
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, acc, 1, 1, 0, 127 - 10, 0, 127);
    for (int e = 0; e < 16; e++) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = lane & 31;
        d[col * 32 + row] = acc[e];
    }
}

static float dec(unsigned char v) {  // OCP e5m2
    const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
    float x;
    if (e == 31) x = m ? NAN : INFINITY;
    else if (e == 0) x = std::ldexp((float)m, -16);
    else x = std::ldexp(1.0f + m / 4.0f, e - 15);
    return s ? -x : x;
}

int main() {
    std::vector<float> a(32 * 64), b(32 * 64);
    srand(3);
    for (auto& x : a) x = ((rand() % 20001) - 10000) * (1.0f / 10000.0f) * (rand() % 4 == 0 ? 2000.f : 3.f);
    for (auto& x : b) x = ((rand() % 20001) - 10000) * (1.0f / 10000.0f) * (rand() % 5 == 0 ? 1e-3f : 1.f);
    // row 0 of A holds the probes: near max, between max and f16's max, beyond, ties (1.125 -> 1.0, 1.375 -> 1.5), subnormals
    const float probes[10] = {57344.0f, 60000.0f, 65504.0f, 1e6f, 1.125f, 1.375f, -3e-5f, 2e-5f, 7e-6f, INFINITY};
    for (int i = 0; i < 10; i++) a[i] = probes[i];
    float *da, *db, *dd; unsigned char *da8, *db8;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dd, 32 * 32 * 4); hipMalloc(&da8, 32 * 64); hipMalloc(&db8, 32 * 64);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ovfl++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, da8, db8, dd, ovfl);
        std::vector<float> d(32 * 32); std::vector<unsigned char> a8(32 * 64), b8(32 * 64);
        hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(a8.data(), da8, a8.size(), hipMemcpyDeviceToHost); hipMemcpy(b8.data(), db8, b8.size(), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d  probes:", ovfl);
        for (int i = 0; i < 10; i++) printf("  %g -> %02x (%g)", probes[i], a8[i], dec(a8[i]));
        printf("\n");
        double worst = 0, worst_q = 0;
        for (int i = 1; i < 32; i++)  // row 0 holds the probes (inf / NaN)
            for (int j = 0; j < 32; j++) {
                double s = 0;
                for (int kk = 0; kk < 64; kk++) s += (double)dec(a8[i * 64 + kk]) * dec(b8[j * 64 + kk]);
                s *= 1.0 / 1024.0;
                worst = fmax(worst, fabs(s - d[i * 32 + j]) / (fabs(s) + 1e-6));
            }
        for (int i = 64; i < 32 * 64; i++) { const float q = dec(a8[i]); if (fabs(a[i]) > 1e-3) worst_q = fmax(worst_q, fabs(q - a[i]) / fabs(a[i])); }
        printf("FP16_OVFL=%d  mfma (cbsz = blgp = 1, scale 2^-10) vs decoded-bytes dot product: worst rel diff %.3g; e5m2 quantisation worst rel err %.3g (expect <= 0.125)\n", ovfl, worst, worst_q);
    }
    return 0;
}
