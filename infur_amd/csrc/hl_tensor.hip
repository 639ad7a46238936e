// hl_tensor.hip -- conversions of the three-byte tensor format of INFUR_DTYPE_F16_HL (hl_format.h) that are not fused into a
// producer: f32 -> hi / lo planes (the unfused stem + max-pool path of keep_activations) and hi / lo planes -> planar f32 (the
// per-layer read-back, infur_debug_read_activation).  Neither runs on the frame path.
#include "hl_format.h"
#include "kernels.h"

namespace infur {

__global__ void __launch_bounds__(256) hl_from_f32_kernel(const float* __restrict__ in, size_t n8, _Float16* __restrict__ hi, unsigned char* __restrict__ lo) {
    hl_set_fp16_ovfl();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        hl_f16x8 hv;
        hl_u32x2 lv;
        hl_split8(x, hv, lv);
        reinterpret_cast<hl_f16x8*>(hi)[i] = hv;
        reinterpret_cast<hl_u32x2*>(lo)[i] = lv;
    }
}

hipError_t launch_hl_from_f32(const float* in, size_t n, void* hi, void* lo, hipStream_t s) {
    if (n & 7) return hipErrorInvalidValue;
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(hl_from_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, n / 8, static_cast<_Float16*>(hi), static_cast<unsigned char*>(lo));
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) hl_nhwc_to_planar_kernel(const _Float16* __restrict__ hi, const unsigned char* __restrict__ lo, int HW, int C,
                                                                float* __restrict__ out) {
    const size_t total = (size_t)HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / HW);
        const size_t p = i - (size_t)c * HW;
        const size_t e = p * C + c;
        out[i] = (float)hi[e] + __builtin_amdgcn_cvt_f32_bf8((int)lo[e], 0) * kHlLoInv;
    }
}

hipError_t launch_hl_nhwc_to_planar(const void* hi, const void* lo, int H, int W, int C, float* out, hipStream_t s) {
    const size_t total = (size_t)H * W * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(hl_nhwc_to_planar_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const _Float16*>(hi), static_cast<const unsigned char*>(lo),
                       H * W, C, out);
    return hipGetLastError();
}

}  // namespace infur
