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
// their epilogue (33 M requantisations per launch at 1080p; ablation of conv1x1_q8.hip on layer3 conv3: 13.7 of 38.6 us are this
// arithmetic, and nothing overlaps it), so this is a third of their time.  q_word below is the packed form of the same steps.
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

// ---- four outputs -> one packed word (the form the u8-output epilogues use) ----
// Two facts measured on gfx950 (experiments/q8epi/cvt_probe.hip) take the per-output count from 14 / 7 (residual sum / none) to
// 9 / 4 instruction issues:
//   * v_cvt_pk_u8_f32 SATURATES to 0..255 (and rounds half to even; -inf -> 0, +inf -> 255): the clamp in front of the final
//     conversion is the instruction itself.  Only the clamp of a QLinearConv output that feeds a QLinearAdd stays (its value is
//     arithmetic input);
//   * v_pk_mul_f32 / v_pk_add_f32 are the scalar IEEE operations on two values per lane (bitwise, probed over 8192 pairs), at the
//     same issue cost as one: the f32 multiplies and adds run on PAIRS.  (A subnormal intermediate, should the packed form flush
//     it, becomes 0 under the rint that follows every product / sum either way.)
struct QEpi {
    float yzp, lo, hi;      // QLinearConv: output zero point, clamp of the centred value
    float ra, rb, bzp, czp;  // QLinearAdd: a_scale / c_scale, b_scale / c_scale, residual zero point, sum zero point
};
typedef float qf2 __attribute__((ext_vector_type(2)));
typedef int qi4 __attribute__((ext_vector_type(4)));
typedef float qf4 __attribute__((ext_vector_type(4)));

template <bool RES>
__device__ __forceinline__ unsigned q_word(const qi4 acc, const qf4 mult, const unsigned resw, const QEpi& e) {
#pragma clang fp contract(off)
    qf2 t0 = {(float)acc[0], (float)acc[1]}, t1 = {(float)acc[2], (float)acc[3]};
    t0 = t0 * qf2{mult[0], mult[1]};
    t1 = t1 * qf2{mult[2], mult[3]};
    t0 = qf2{__builtin_rintf(t0[0]), __builtin_rintf(t0[1])};
    t1 = qf2{__builtin_rintf(t1[0]), __builtin_rintf(t1[1])};
    if constexpr (RES) {
        t0 = qf2{__builtin_amdgcn_fmed3f(t0[0], e.lo, e.hi), __builtin_amdgcn_fmed3f(t0[1], e.lo, e.hi)};
        t1 = qf2{__builtin_amdgcn_fmed3f(t1[0], e.lo, e.hi), __builtin_amdgcn_fmed3f(t1[1], e.lo, e.hi)};
        t0 = t0 * e.ra;
        t1 = t1 * e.ra;
        qf2 b0 = {q_byte(resw, 0), q_byte(resw, 1)}, b1 = {q_byte(resw, 2), q_byte(resw, 3)};
        b0 = (b0 - e.bzp) * e.rb;
        b1 = (b1 - e.bzp) * e.rb;
        t0 = t0 + b0;
        t1 = t1 + b1;
        t0 = qf2{__builtin_rintf(t0[0]), __builtin_rintf(t0[1])} + e.czp;
        t1 = qf2{__builtin_rintf(t1[0]), __builtin_rintf(t1[1])} + e.czp;
    } else {
        t0 = t0 + e.yzp;
        t1 = t1 + e.yzp;
    }
    unsigned w = 0;
    w = q_pack(t0[0], 0, w);
    w = q_pack(t0[1], 1, w);
    w = q_pack(t1[0], 2, w);
    w = q_pack(t1[1], 3, w);
    return w;
}

}  // namespace infur
