// Internal launcher interface between the host runtime (infur_capi.cpp) and the
// gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace infur {

// NHWC activations (f32 or f16); weights OHWI ([Cout][KH][KW][Cin], K contiguous per output
// channel) in the same element type; bias always f32.
struct ConvArgs {
    const void* in;     // [H][W][Cin]
    const void* wt;     // [Cout][KH*KW*Cin]
    const float* bias;  // [Cout], may be null (no bias)
    const void* res;    // optional residual, same shape and type as the input activations
    void* out;          // [OH][OW][Cout]
    int H, W, Cin;
    int OH, OW, Cout;
    int KH, KW, stride, pad, dil;
    int relu;
    // batched use (Winograd-domain GEMMs): `batch` independent problems of the same shape,
    // operand b starts at base + b * stride (bytes).  batch <= 1: plain convolution.
    int batch = 1;
    size_t in_bs = 0, wt_bs = 0, out_bs = 0;
    // mode 2 (f32 split into f16 pairs) only: activations are multiplied by a_scale (a power of two
    // that centres them in the f16 range) while they are staged, the weights were scaled and split at
    // load time (launch_split_weights), and the accumulators are multiplied by acc_scale =
    // 1 / (a_scale * weight scale) before bias / residual / ReLU.  Powers of two: exact.
    float a_scale = 1.0f, acc_scale = 1.0f;
    // batched use in the split modes: one accumulator scale PER PROBLEM (device array of `batch` floats) instead of acc_scale --
    // the (m+2)^2 Winograd-domain weight planes differ by orders of magnitude (G's entries run from 1/180 to 1), so each
    // plane is scaled into the f16 pair's range on its own
    const float* acc_scale_b = nullptr;
    // mode 2, optional: atomicMax target (bit pattern of a non-negative float) for max |output| of this launch --
    // the range monitor of the split mode (infur_split_range)
    unsigned* amax = nullptr;
    // mode 4 (quantised models: u8 activations, s8 weights, exact i32 accumulation on v_mfma_i32_32x32x32_i8): the requantisation
    // of ONNX QLinearConv in the epilogue -- y = sat_u8(round(f32(acc + q_bias[o]) * q_mult[o]) + q_yzp) -- then, with `res`, the
    // com.microsoft QLinearAdd of the residual sum -- c = sat_u8(round(f32(y - q_yzp) * q_ra + f32(res - q_bzp) * q_rb) + q_czp)
    // -- and, for an f32 output (the logits), DequantizeLinear: f32(y - q_yzp) * q_dq.  q_bias already holds the operator's
    // bias + (128 - x_zp) * sum_k w[o][k]: the kernel feeds the MFMA x - 128 (one XOR per fragment; an out-of-range tap loads 0,
    // i.e. x = 0 = the zero point of every padded tensor).
    const float* q_mult = nullptr;
    const int32_t* q_bias = nullptr;
    int q_yzp = 0, q_bzp = 0, q_czp = 0;
    float q_ra = 0.f, q_rb = 0.f, q_dq = 0.f;
    // f32 output: (y - q_yzp + q_dq_off) * q_dq.  DequantizeLinear: off 0, dq = y_scale; the codes themselves (models that Resize
    // before they dequantise): off = q_yzp, dq = 1
    float q_dq_off = 0.f;
    // two-source 1x1 GEMM (a bottleneck's conv3 and its downsample branch as ONE launch, no residual tensor):
    // out = W[:, :Cin] * in  +  W[:, Cin:] * in2(stride2)  + bias.  in2 == nullptr: ordinary convolution.
    // Requires KH = KW = 1, pad = 0, stride = 1; wt rows are Cin + Cin2 long; OH x OW = ceil(H2/stride2) x ...
    const void* in2 = nullptr;  // [H2][W2][Cin2]
    int H2 = 0, W2 = 0, Cin2 = 0, stride2 = 1;
    // mode 5 (INFUR_DTYPE_F16_HL, conv_hl.hip): every tensor is an f16 hi plane (in / wt / res / out above) plus an e5m2 lo plane of
    // the same shape (hl_format.h); batched use: lo plane b starts at base + b * (in_bs / 2).  out_f32 launches write plain f32.
    const void* in_lo = nullptr;
    const void* wt_lo = nullptr;
    const void* res_lo = nullptr;
    void* out_lo = nullptr;
    const void* in2_lo = nullptr;
};

// conv as implicit GEMM on the matrix cores.  mode 0: f32 operands on the f32 MFMA (Cin % 32 == 0);
// mode 1: f16 operands, f32 accumulation (Cin % 64 == 0), output f16 or (out_f32) f32;
// mode 2: f32 tensors, each value split into an f16 hi + lo pair while it is staged, three f16
// MFMAs per product, f32 accumulation (Cin % 32 == 0) -- f32-grade results at f16 matrix rate / 3.
// mode 3: as mode 2, but the two cross terms hi * lo run on the bf8 (OCP e5m2) MX MFMA: 2 MFMA units per product instead of
// 3, products exact to ~2^-13 whatever the tensors' dynamic range (e5m2 has f16's exponent range: no scales; round 3 used
// e4m3 under per-tensor scales, which heavy-tailed weights broke); weights prepared with launch_split_weights(fp8_cross = 1).
// mode 4: quantised (ConvArgs::q_*): u8 NHWC activations, s8 OHWI weights (Cin % 128 == 0), output u8 or (out_f32) dequantised f32.
// cfg: tile configuration index (conv_igemm_num_configs), -1 = built-in heuristic.  Every
// configuration produces bit-identical results; only the speed differs.
// mode 5: three-byte tensors (f16 hi + e5m2 lo planes, ConvArgs::*_lo), hi * hi on the f16 MFMA + both cross terms on the bf8 MX MFMA,
// every operand staged by LDS-DMA (conv_hl.hip); Cin % 32 == 0; configurations 11, 0, 6, 5 (their tile shapes).
hipError_t launch_conv_igemm(const ConvArgs& a, int mode, int out_f32, int cfg, hipStream_t s);
int conv_igemm_num_configs();
int conv_igemm_config_tile_area(int cfg);  // BM * BN of a configuration (operand re-reads fall with it)
int conv_igemm_default_config(const ConvArgs& a);
bool conv_igemm_config_valid(const ConvArgs& a, int cfg, int mode, int out_f32);
const char* conv_igemm_config_name(int cfg, int mode);

// mode 5 (conv_hl.hip)
bool conv_hl_config_valid(const ConvArgs& a, int cfg, int out_f32);
hipError_t launch_conv_hl(const ConvArgs& a, int out_f32, int cfg, hipStream_t s);
// configuration 15 of mode 5 (conv_hl_areg.hip): 1x1 expansions (Cin 64 / 128 / 256, Cout a multiple of 128) with the activation
// fragment in registers, weights and residual streamed by LDS-DMA; bit-identical to the tiled forms
bool conv_hl_areg_valid(const ConvArgs& a, int out_f32);
hipError_t launch_conv_hl_areg(const ConvArgs& a, hipStream_t s);
// weights of mode 5, one-off at load: f32 [n] (kernel K order) * scale -> f16 hi [n], e5m2 lo [n] (n % 4 == 0)
hipError_t launch_hl_pack_weights(const float* w, size_t rows, size_t cols, float scale, void* hi, void* lo, hipStream_t s);  // K-block-major planes
hipError_t launch_hl_pack_weights_planes(const float* w, size_t rows, size_t cols, int planes, const float* scales, void* hi, void* lo, hipStream_t s);  // <= 64 planes back to back, one launch

// 1x1 convolution with Cin in {64, 128, 256}, f16 operands (mode 1), f16 output: the activation tile stays in registers
// while the workgroup walks all N tiles (conv1x1_areg.hip).  Reached through launch_conv_igemm as one more configuration.
bool conv1x1_areg_valid(const ConvArgs& a, int mode, int out_f32);
hipError_t launch_conv1x1_areg(const ConvArgs& a, hipStream_t s);
// quantised models (mode 4, u8 output): activation tile in registers, 16 consecutive channels per lane straight from the
// accumulators (conv1x1_q8.hip): configuration 15 in that mode
bool conv1x1_q8_valid(const ConvArgs& a, int mode, int out_f32);
hipError_t launch_conv1x1_q8(const ConvArgs& a, int nsplit, hipStream_t s);
int conv1x1_q8_nsplit(const ConvArgs& a);  // configuration 18: N tiles shared out over this many workgroups per M tile (0: not a candidate)

// stride-1 3x3 convolutions (pad = dilation 1 / 2 / 4), f16 operands and output, no residual: the input patch of a 16 x 16 output
// tile stays in LDS for all nine taps (conv3x3_halo.hip).  Configurations 19 (bn = 128), 20 (bn = 256) and 21 (the 4-wave form) of mode 1.
bool conv3x3_halo_valid(const ConvArgs& a, int mode, int out_f32, int bn);
hipError_t launch_conv3x3_halo(const ConvArgs& a, int bn, hipStream_t s);
// the same kernel on the quantised model's tensors (mode 4, u8 out): configurations 19 / 20 of that mode
hipError_t launch_conv3x3_halo_q(const ConvArgs& a, int bn, hipStream_t s);
// the 4-wave form (configuration 21): BN = 256, one wave per SIMD with a 128 x 128 wave tile, dilation 1 / 2
bool conv3x3_halo4_valid(const ConvArgs& a, int mode, int out_f32);
hipError_t launch_conv3x3_halo4(const ConvArgs& a, hipStream_t s);

// Two 1x1 convolutions back to back on the same pixels, f16 (conv1x1_b2b.hip): y = ReLU(w3 * in + b3 + res) -- a
// bottleneck's conv3 + residual -- is written once and immediately multiplied by the NEXT bottleneck's conv1 weights:
// out2 = ReLU(w1 * y + b1).  C2 = channels of `in` = channels of out2 (128 or 256); y and res have 4 * C2 channels.
// Bit-identical to the two launches it replaces.
struct B2bArgs {
    const void* in;    // [M][C2] f16
    const void* w3;    // [4*C2][C2] f16 in the step-interleaved row order of launch_b2b_pack_w3
    const float* b3;   // [4*C2]
    const void* res;   // [M][4*C2] f16
    void* y;           // [M][4*C2] f16
    const void* w1;    // [C2][4*C2] f16
    const float* b1;   // [C2]
    void* out2;        // [M][C2] f16
    int M, C2, relu1, relu2;
};
bool conv1x1_b2b_valid(const B2bArgs& a);
// one-off at model load: the conv3 weight matrix ([4*C2][C2] f16, plain row order) in the row order the kernel streams
hipError_t launch_b2b_pack_w3(const void* w3, void* w3i, int C2, hipStream_t s);
hipError_t launch_conv1x1_b2b(const B2bArgs& a, hipStream_t s);

// Winograd F(mt x mt, 3x3), mt = 2 or 4, for stride-1 3x3 convs (f32, any dilation d with pad = d);
// (mt+2)^2 transform planes; see winograd.hip
int wino_num_tiles(int H, int W, int d, int mt);
// amax (optional): atomicMax target for max |V| / max |out| (range monitor of the split mode)
hipError_t launch_wino_input(const float* in, int H, int W, int C, int d, int mt, float* V, unsigned* amax, hipStream_t s);
hipError_t launch_wino_output(const float* M, int H, int W, int Cout, int d, int mt, const float* bias, int relu,
                              float* out, unsigned* amax, hipStream_t s);
hipError_t launch_wino_weights(const float* w_oihw, int O, int I, int mt, float* U, hipStream_t s);
// the same transforms on three-byte tensors (mode 5; C, Cout % 128 == 0): input hi / lo planes -> V * v_scale as hi / lo planes
// [(mt+2)^2][T][C]; f32 M -> the conv's output as hi / lo planes
hipError_t launch_wino_input_hl(const void* in_hi, const void* in_lo, int H, int W, int C, int d, int mt, float v_scale, void* V_hi, void* V_lo,
                                hipStream_t s);
hipError_t launch_wino_output_hl(const float* M, int H, int W, int Cout, int d, int mt, const float* bias, int relu, void* out_hi, void* out_lo,
                                 hipStream_t s);

// three-byte tensors (hl_format.h): the lo plane of an [elems] tensor starts hl_lo_offset(elems) bytes after the hi plane
__host__ __device__ inline size_t hl_lo_offset(size_t elems) { return (elems * 2 + 255) & ~(size_t)255; }
__host__ __device__ inline size_t hl_tensor_bytes(size_t elems) { return hl_lo_offset(elems) + ((elems + 255) & ~(size_t)255); }
// f32 [n] -> hi / lo planes (n % 8 == 0); NHWC hi / lo planes -> planar f32 [C][H][W] (read-back)
hipError_t launch_hl_from_f32(const float* in, size_t n, void* hi, void* lo, hipStream_t s);
hipError_t launch_hl_nhwc_to_planar(const void* hi, const void* lo, int H, int W, int C, float* out, hipStream_t s);

// stem: packed BGR u8 -> (LUT normalise, BGR->RGB) -> conv 7x7/2 pad 3 (3->64) + bias + ReLU,
// NHWC out (f32, or f16 when f16 != 0; the arithmetic is f32 either way).
// wt: [7][7][3][64] (ky,kx,c,cout) f32, lut: [3][256] in RGB order.
hipError_t launch_stem_conv7x7(const uint8_t* bgr, int H, int W, const float* wt,
                               const float* bias, const float* lut, void* out, int f16, int OH, int OW,
                               hipStream_t s);

// stem + maxpool 3x3/2 pad 1 in one kernel: out = pooled [PH][PW][64]; the SH x SW stem tensor is never written.
// mode 0: exact f32 MFMA, bit-identical to launch_stem_conv7x7 -> launch_maxpool3x3s2 (f32 out).  mode 1 (f16 out): f16
// operands on the f16 MFMA.  mode 2 (f32 out): operands as f16 hi + lo pairs of value * a_scale / weight * w_scale (powers
// of two), three MFMAs per product, f32-grade.
// quantised model: QuantizeLinear + QLinearConv 7x7/2 + u8 max-pool in one launch, exact on the f16 MFMA (stem_pool.hip).
// wt: [147][64] f32 = the s8 weights, k = (ky * 7 + kx) * 3 + c; lut: [3][256] f32 = q - x_zp (RGB order); out: [PH][PW][128] u8
hipError_t launch_stem_pool_q(const uint8_t* bgr, int H, int W, const void* wimg, const float* lut, const int32_t* q_bias, const float* q_mult,
                              int y_zp, uint8_t* out, int cstride, int SH, int SW, int PH, int PW, hipStream_t s);
// the f16-rate stems' weights as their finished LDS image (f16 hi (, lo) planes, pads zero), built once per model from wt[147][64] f32
size_t stem16_image_bytes();
hipError_t launch_stem16_pack(const float* wt, float w_scale, int split, void* img, hipStream_t s);
hipError_t launch_stem_pool(const uint8_t* bgr, int H, int W, const float* wt, const void* wimg, const float* bias, const float* lut, void* out,
                            int mode, int SH, int SW, int PH, int PW, float a_scale, float w_scale, unsigned* amax, hipStream_t s);

// maxpool 3x3 stride 2 pad 1, NHWC f32 / f16 (C % 4 == 0)
hipError_t launch_maxpool3x3s2(const void* in, int H, int W, int C, void* out, int f16, int OH, int OW,
                               unsigned* amax, hipStream_t s);

// OIHW f32 -> OHWI f32 / f16 weight repack (one-off at model load)
hipError_t launch_repack_oihw_to_ohwi(const float* src, void* dst, int f16, int O, int I, int KH, int KW,
                                      hipStream_t s);
// mode 2 weight preparation (one-off at model load).  absmax: *out = max |w[i]| (out must be zeroed).
// split: in place, every 32 consecutive f32 (one 128-byte K-step row chunk) become
// [32 x f16 hi][32 x f16 lo] of w * scale, hi = rne(w*scale), lo = rne(w*scale - hi).
hipError_t launch_absmax(const float* w, size_t n, float* out, hipStream_t s);
// out[p] = max |plane p| for `planes` tensors of `per` elements back to back, one launch (out must be zeroed)
hipError_t launch_absmax_planes(const float* w, size_t per, int planes, float* out, hipStream_t s);
// fp8_cross: the [f16 hi][e5m2 lo * 2^11][e5m2 hi] row form of conv_igemm mode 3 instead of [f16 hi][f16 lo]
hipError_t launch_split_weights(float* w, size_t n, float scale, int fp8_cross, hipStream_t s);
// two-source GEMM weights (one-off at load): out[r] = a[r] ++ b[r] for `rows` rows of a_bytes / b_bytes
// (multiples of 16); sum[i] = x[i] + y[i]
hipError_t launch_concat_rows(const void* a, size_t a_bytes, const void* b, size_t b_bytes, void* out, int rows, hipStream_t s);
hipError_t launch_add_f32(const float* x, const float* y, float* sum, int n, hipStream_t s);
// OIHW (O=64,I=3,7x7) -> [ky][kx][c][o] for the stem kernel
// reverse_c: the kernels read a pixel's channels as (p[2], p[1], p[0]); a model whose channel 0 is B (Uint8 input:
// BGR kept, predict_onnx.rs:296-301) gets its stem weights stored with the input-channel axis reversed instead
hipError_t launch_repack_stem(const float* src, float* dst, int reverse_c, hipStream_t s);

// ---- quantised models (quant.hip) ----
// stem: packed BGR u8 frame -> QuantizeLinear of the normalised image through a [3][256] u8 table (RGB order; out-of-frame
// taps take x_zp) -> QLinearConv 7x7/2 pad 3 (wq: [64][7][7] dwords = (r, g, b, 0) s8; q_bias / q_mult per channel) -> u8
// NHWC [SH][SW][64]
hipError_t launch_stem_q(const uint8_t* bgr, int H, int W, const uint8_t* qlut, int x_zp, const int32_t* wq, const int32_t* q_bias,
                         const float* q_mult, int y_zp, uint8_t* out, int SH, int SW, hipStream_t s);
// max-pool 3x3/2 pad 1 on u8 NHWC [H][W][C] -> [OH][OW][CP] (CP >= C: channels C..CP-1 are written as 0; C, CP % 16 == 0)
hipError_t launch_maxpool_q(const uint8_t* in, int H, int W, int C, uint8_t* out, int OH, int OW, int CP, hipStream_t s);
// OIHW s8 -> OHWI s8 with the input-channel axis zero-padded to IP and `OP - O` zero rows appended; also sums every row
// (wsum[o] = sum_k w[o][k], i32; rows O..OP-1: 0)
hipError_t launch_repack_q(const int8_t* src, int8_t* dst, int32_t* wsum, int O, int I, int KH, int KW, int OP, int IP, hipStream_t s);
// u8 NHWC -> planar f32 [C][H][W] (debug read-back of a quantised activation: the byte values)
hipError_t launch_u8_nhwc_to_planar(const uint8_t* in, int H, int W, int C, float* out, hipStream_t s);

// Scale: packed BGR u8 resize.  mode 0 nearest, 1 bilinear (definitions: oracle/infur_oracle.c)
hipError_t launch_scale_bgr(const uint8_t* in, int W, int H, uint8_t* out, int OW, int OH,
                            int mode, hipStream_t s);

// pre-proc only (used when the stem is not fused): BGR u8 HWC -> RGB f32 CHW via LUT
hipError_t launch_pack_normalize(const uint8_t* bgr, int W, int H, const float* lut, float* chw,
                                 hipStream_t s);

// display conversion (app.rs:132-144): packed BGR -> [r,g,b,255]
hipError_t launch_bgr_to_rgba(const uint8_t* bgr, int W, int H, uint32_t* rgba, hipStream_t s);

// NHWC (f32 or f16) -> planar f32 [C][H][W] (low-res logits, debug activation read-back)
hipError_t launch_nhwc_to_planar(const void* in, int f16, int H, int W, int C, float* out,
                                 hipStream_t s);

// quantised models that Resize their u8 logits before DequantizeLinear (prepost.hip: up_post): the low-res tensor holds the
// codes, every interpolated value v becomes (trunc(v) - zp) * scale.  on == 0: plain float interpolation.
struct UpQuant {
    int on = 0;
    float zp = 0.f, scale = 1.f;
};

// bilinear up-sample (ONNX Resize linear / pytorch_half_pixel) of NHWC low-res logits to
// planar [K][OH][OW] f32
hipError_t launch_upsample_planar(const float* low, int LH, int LW, int K, float* out, int OH,
                                  int OW, hipStream_t s, const UpQuant uq = UpQuant());

// ColorCode over planar [K][H][W] f32 -> premultiplied RGBA8.  lut: [20][256] uchar4
hipError_t launch_colorcode_planar(const float* khw, int K, int H, int W, const uint32_t* lut,
                                   uint32_t* rgba, hipStream_t s);

// fused: bilinear up-sample of NHWC low-res logits + argmax + shade -> RGBA8; never
// materialises the full-resolution logits.  Bit-identical to upsample_planar -> colorcode.
hipError_t launch_upsample_argmax_shade(const float* low, int LH, int LW, int K,
                                        const uint32_t* lut, uint32_t* rgba, int OH, int OW,
                                        hipStream_t s, const UpQuant uq = UpQuant());

}  // namespace infur
