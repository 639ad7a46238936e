// Micro-benchmark: MFMA instruction mix of the split mode (3 f16 MFMAs per product pair) against a mix with the two cross
// terms on the fp8 MX MFMA (2 f16 + 1 f8f6f4 32x32x64 per 32 k), registers only, random operands (DVFS: random data).
// hipcc --offload-arch=gfx950 -O3 mix_bench.hip -o mix_bench && ./mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) mix(const int* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    f32x16 acc[4][2];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    // operands: random bit patterns reinterpreted as f16 in [-2,2) would be better; use small-magnitude f16 built from src
    f16x8 ah[4][2], al[4][2], bh[2][2], bl[2][2];
    i32x8 a8[4], b8[2];
    const int* p = src + (blockIdx.x * 512 + lane) * 64;
    int q = 0;
    auto h8 = [&](int o) { f16x8 v; for (int e = 0; e < 8; e++) v[e] = (_Float16)(((p[(o + e) & 63] >> 8) & 1023) * (1.0f / 512.0f) - 1.0f); return v; };
    for (int i = 0; i < 4; i++) for (int s = 0; s < 2; s++) { ah[i][s] = h8(q); q += 3; al[i][s] = h8(q); q += 5; }
    for (int j = 0; j < 2; j++) for (int s = 0; s < 2; s++) { bh[j][s] = h8(q); q += 7; bl[j][s] = h8(q); q += 11; }
    for (int i = 0; i < 4; i++) for (int e = 0; e < 8; e++) a8[i][e] = p[(e * 5 + i) & 63] & 0x3f3f3f3f;  // fp8 bytes with small exponents
    for (int j = 0; j < 2; j++) for (int e = 0; e < 8; e++) b8[j][e] = p[(e * 3 + j + 9) & 63] & 0x3f3f3f3f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (MODE == 0) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j][s], al[i][s], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j][s], ah[i][s], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j][s], ah[i][s], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j][s], ah[i][s], acc[i][j], 0, 0, 0);
                        if (s == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j], a8[i], acc[i][j], 0, 0, 0, 116, 0, 127);
                    }
                }
        // keep operands "live" and data changing a little
        ah[it & 3][0][0] += (_Float16)0.001f;
    }
    float sum = 0.f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) sum += acc[i][j][e];
    out[blockIdx.x * 512 + lane] = sum;
}

int main() {
    const int blocks = 256 * 2, iters = 20000;
    std::vector<int> h(blocks * 512 * 64);
    srand(1);
    for (auto& x : h) x = rand();
    int* d; float* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(mix<0>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(mix<1>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // f32-equivalent flops: per iteration per wave: 8 blocks x 32x32 x 32 k x 2
            const double fl = (double)blocks * 8 * iters * 8.0 * 32 * 32 * 32 * 2;
            printf("mode %d (%s): %.2f ms  %.1f TFLOP/s f32-equivalent\n", mode, mode ? "2 f16 + 1 fp8-MX per 32 k" : "6 f16 per 32 k", ms, fl / ms / 1e9);
        }
    }
    return 0;
}
