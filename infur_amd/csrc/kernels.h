// Internal launcher interface between the host runtime (infur_capi.cpp) and the
// gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace infur {

// NHWC f32 activations; weights OHWI ([Cout][KH][KW][Cin], K contiguous per output channel).
struct ConvArgs {
    const float* in;    // [H][W][Cin]
    const float* wt;    // [Cout][KH*KW*Cin]
    const float* bias;  // [Cout]
    const float* res;   // optional residual, same shape as out
    float* out;         // [OH][OW][Cout]
    int H, W, Cin;
    int OH, OW, Cout;
    int KH, KW, stride, pad, dil;
    int relu;
};

// conv as implicit GEMM on v_mfma_f32_32x32x2_f32 (Cin % 32 == 0)
hipError_t launch_conv_igemm_f32(const ConvArgs& a, hipStream_t s);
// name of the tile configuration launch_conv_igemm_f32 picks for these arguments
const char* conv_igemm_f32_config(const ConvArgs& a);

// stem: packed BGR u8 -> (LUT normalise, BGR->RGB) -> conv 7x7/2 pad 3 (3->64) + bias + ReLU,
// NHWC f32 out.  wt: [7][7][3][64] (ky,kx,c,cout), lut: [3][256] in RGB order.
hipError_t launch_stem_conv7x7(const uint8_t* bgr, int H, int W, const float* wt,
                               const float* bias, const float* lut, float* out, int OH, int OW,
                               hipStream_t s);

// maxpool 3x3 stride 2 pad 1, NHWC f32 (C % 4 == 0)
hipError_t launch_maxpool3x3s2(const float* in, int H, int W, int C, float* out, int OH, int OW,
                               hipStream_t s);

// OIHW -> OHWI weight repack (one-off at model load)
hipError_t launch_repack_oihw_to_ohwi(const float* src, float* dst, int O, int I, int KH, int KW,
                                      hipStream_t s);
// OIHW (O=64,I=3,7x7) -> [ky][kx][c][o] for the stem kernel
hipError_t launch_repack_stem(const float* src, float* dst, hipStream_t s);

// Scale: packed BGR u8 resize.  mode 0 nearest, 1 bilinear (definitions: oracle/infur_oracle.c)
hipError_t launch_scale_bgr(const uint8_t* in, int W, int H, uint8_t* out, int OW, int OH,
                            int mode, hipStream_t s);

// pre-proc only (used when the stem is not fused): BGR u8 HWC -> RGB f32 CHW via LUT
hipError_t launch_pack_normalize(const uint8_t* bgr, int W, int H, const float* lut, float* chw,
                                 hipStream_t s);

// display conversion (app.rs:132-144): packed BGR -> [r,g,b,255]
hipError_t launch_bgr_to_rgba(const uint8_t* bgr, int W, int H, uint32_t* rgba, hipStream_t s);

// low-res NHWC logits [lh][lw][K] -> planar [K][lh][lw]
hipError_t launch_nhwc_to_planar(const float* in, int H, int W, int C, float* out,
                                 hipStream_t s);

// bilinear up-sample (ONNX Resize linear / pytorch_half_pixel) of NHWC low-res logits to
// planar [K][OH][OW] f32
hipError_t launch_upsample_planar(const float* low, int LH, int LW, int K, float* out, int OH,
                                  int OW, hipStream_t s);

// ColorCode over planar [K][H][W] f32 -> premultiplied RGBA8.  lut: [20][256] uchar4
hipError_t launch_colorcode_planar(const float* khw, int K, int H, int W, const uint32_t* lut,
                                   uint32_t* rgba, hipStream_t s);

// fused: bilinear up-sample of NHWC low-res logits + argmax + shade -> RGBA8; never
// materialises the full-resolution logits.  Bit-identical to upsample_planar -> colorcode.
hipError_t launch_upsample_argmax_shade(const float* low, int LH, int LW, int K,
                                        const uint32_t* lut, uint32_t* rgba, int OH, int OW,
                                        hipStream_t s);

}  // namespace infur
