// qepilogue.h -- the requantisation arithmetic of the quantised operators (device code shared by conv_igemm.hip mode 4,
// conv1x1_areg8.hip and quant.hip).  Every floating-point step is ONE f32 operation, never contracted: the results are defined
// bit for bit (oracle/infur_qoracle.py restates them from the ONNX / com.microsoft operator text).
//
// The values travel as integer-valued FLOATS between the steps -- the operator text converts to u8 and back, which is the identity
// on integers in 0..255 -- and the clamp of a QLinearConv output is applied to the value CENTRED on its zero point:
//     sat_u8(r + zp) - zp  ==  med3(r, -zp, 255 - zp)          (r = round_half_even(f32(acc) * mult), an integer-valued float;
//                                                               |r| >= 2^24 saturates on both sides)
// which is what QLinearAdd / DequantizeLinear subtract first anyway.  Per output that is 14 VALU operations with a residual
// sum and 7 without, against 24 / 11 for the literal convert-clamp-convert form: the quantised 1x1 expansions are VALU-bound in
// their epilogue (33 M requantisations per launch at 1080p), so this is a third of their time.
#pragma once

namespace infur {

// QLinearConv: round_half_even(f32(acc) * mult), clamped to [lo, hi] = [-y_zp, 255 - y_zp]: the output minus its zero point
__device__ __forceinline__ float q_requant_c(const int acc, const float mult, const float lo, const float hi) {
#pragma clang fp contract(off)
    const float t = (float)acc * mult;
    return __builtin_amdgcn_fmed3f(__builtin_rintf(t), lo, hi);
}

// QLinearAdd: sat_u8(round(f32(a - a_zp) * ra + f32(b - b_zp) * rb) + c_zp), a - a_zp given (q_requant_c), b the residual byte
__device__ __forceinline__ float q_add_c(const float a_c, const float ra, const float b, const float bzp, const float rb, const float czp) {
#pragma clang fp contract(off)
    const float ta = a_c * ra;
    const float tb = (b - bzp) * rb;
    float t = ta + tb;
    t = __builtin_rintf(t) + czp;
    return __builtin_amdgcn_fmed3f(t, 0.f, 255.f);
}

// byte `sel` of `word` <- the integer-valued float v in [0, 255]
__device__ __forceinline__ unsigned q_pack(const float v, const unsigned sel, const unsigned word) { return __builtin_amdgcn_cvt_pk_u8_f32(v, sel, word); }

__device__ __forceinline__ float q_byte(const unsigned word, const int sel) { return (float)((word >> (8 * sel)) & 0xffu); }  // v_cvt_f32_ubyteN

}  // namespace infur
