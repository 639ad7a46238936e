// hl_format.h -- the three-byte tensor format of INFUR_DTYPE_F16_HL (conv_hl.hip): x ~= hi + lo8 / kHlLoScale with
//   hi  = rne16(x)                          f16 plane   [pixels][C]
//   lo8 = rne8((x - hi) * kHlLoScale)       e5m2 plane  [pixels][C]  (OCP bf8: f16's exponent range, no tensor scale)
// kHlLoScale = 2^kHlLoShift * kHlDebias: the power of two is undone by the bf8 MFMA's block scale; kHlDebias stays in the
// product on purpose -- the lo plane of one operand always meets the TRUNCATED top byte t(.) of the other operand's f16, whose
// magnitude is short by that factor on average (least squares over f16 mantissas: 1.087 uniform / 1.088 log-uniform, mean
// ratio 1.095; scripts/sim_hl_assign.py).  Readers that want the value (residual add, Winograd input transform, read-back)
// divide it out again: kHlLoInv.  Device helpers shared by every producer / consumer of the format.
#pragma once
#include <hip/hip_runtime.h>

namespace infur {

constexpr int kHlLoShift = 11;
constexpr float kHlDebias = 1.09f;
constexpr float kHlLoScale = 2048.0f * kHlDebias;
constexpr float kHlLoInv = 1.0f / kHlLoScale;
constexpr float kHlLoMax = 57344.0f;  // largest finite e5m2

typedef _Float16 hl_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hl_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned hl_u32x2 __attribute__((ext_vector_type(2)));
typedef float hl_f32x2 __attribute__((ext_vector_type(2)));

// four remainders (x - hi) -> four e5m2 bytes of rem * kHlLoScale, clamped to the finite range
__device__ __forceinline__ unsigned hl_pack_lo4(const float* rem) {
    float s[4];
#pragma unroll
    for (int t = 0; t < 4; t++) s[t] = fminf(fmaxf(rem[t] * kHlLoScale, -kHlLoMax), kHlLoMax);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(s[0], s[1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(s[2], s[3], w, true);
    return (unsigned)w;
}

// four e5m2 bytes -> the four remainders they stand for
__device__ __forceinline__ void hl_lo8_to_f32(const unsigned w, float* out) {
    const hl_f32x2 a = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, true);
    out[0] = a[0] * kHlLoInv;
    out[1] = a[1] * kHlLoInv;
    out[2] = b[0] * kHlLoInv;
    out[3] = b[1] * kHlLoInv;
}

// eight values -> hi (f16 x 8) and lo (8 bytes).  The caller runs with MODE.FP16_OVFL = 1 (hl_set_fp16_ovfl()).
// Round 6: the value is clamped FIRST -- one v_med3 to [lb, kHlHiMax], lb = 0 doing the caller's ReLU in the same instruction (a NaN
// comes out as lb, like fmaxf(x, 0)) -- so hi cannot overflow, |x - hi| <= 16 and the lo byte needs no clamp of its own; hi comes from
// the packed conversion alone.  5 VALU instructions per value instead of 8.25 (+ 2 for a separate ReLU); identical results for every
// |x| <= 65504 (the epilogues' split was ~3.5 % of the f16hl frame's kernel time).
constexpr float kHlHiMax = 65520.0f;  // halfway to 2^16: rounds to the f16 maximum under FP16_OVFL
typedef float hl_f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 hl_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hl_pack_lo4_nc(const float* s) {  // (no clamp: the caller bounds its argument)
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(s[0], s[1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(s[2], s[3], w, true);
    return (unsigned)w;
}
__device__ __forceinline__ void hl_split8(const float* x, hl_f16x8& hv, hl_u32x2& lv, const float lb = -kHlHiMax) {
    float c[8], s[8];
#pragma unroll
    for (int t = 0; t < 8; t++) c[t] = __builtin_amdgcn_fmed3f(x[t], lb, kHlHiMax);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const hl_f32x2v v = {c[2 * p], c[2 * p + 1]};
        const hl_f16x2 h = __builtin_convertvector(v, hl_f16x2);
        hv[2 * p] = h[0];
        hv[2 * p + 1] = h[1];
    }
#pragma unroll
    for (int t = 0; t < 8; t++) s[t] = (c[t] - (float)hv[t]) * kHlLoScale;
    lv.x = hl_pack_lo4_nc(s);
    lv.y = hl_pack_lo4_nc(s + 4);
}
__device__ __forceinline__ void hl_split4(const float* x, hl_f16x4& hv, unsigned& lv, const float lb = -kHlHiMax) {
    float c[4], s[4];
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_fmed3f(x[t], lb, kHlHiMax);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const hl_f32x2v v = {c[2 * p], c[2 * p + 1]};
        const hl_f16x2 h = __builtin_convertvector(v, hl_f16x2);
        hv[2 * p] = h[0];
        hv[2 * p + 1] = h[1];
    }
#pragma unroll
    for (int t = 0; t < 4; t++) s[t] = (c[t] - (float)hv[t]) * kHlLoScale;
    lv = hl_pack_lo4_nc(s);
}
// eight values back
__device__ __forceinline__ void hl_join8(const hl_f16x8 hv, const hl_u32x2 lv, float* x) {
    float lo[8];
    hl_lo8_to_f32(lv.x, lo);
    hl_lo8_to_f32(lv.y, lo + 4);
#pragma unroll
    for (int t = 0; t < 8; t++) x[t] = (float)hv[t] + lo[t];
}
struct HlTag {  // element-type tag of kernels templated on their output type: sizeof == 3
    unsigned char b[3];
};
__device__ __forceinline__ void hl_set_fp16_ovfl() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

}  // namespace infur
