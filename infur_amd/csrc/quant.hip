// quant.hip -- the kernels of the quantised path that are not convolutions on the i8 MFMA (conv_igemm.hip, mode 4):
// the stem (QuantizeLinear of the normalised image + QLinearConv 7x7/2), the u8 max-pool, the one-off weight repack and the
// debug read-back.  Replaces the nodes ONNX Runtime executes for `fcn-resnet50-12-int8.onnx` inside `session.run`
// (infur/src/predict_onnx.rs:138; the file the reference's own tests load: infur-test-gen/build.rs:88-93).
//
// Integer arithmetic is exact; every floating-point step of a requantisation is one f32 operation, never contracted --
// the results are defined bit for bit (oracle/infur_qoracle.py).
#include "kernels.h"
#include "qepilogue.h"

namespace infur {

namespace {

// ---- stem: frame bytes -> quantised image (table) -> 7x7/2 convolution as 49 four-way dot products per output channel ----
// A workgroup owns 8 x 32 output pixels; the 21 x 69 input patch sits in LDS as one dword per pixel: (r, g, b, 0) with 128
// subtracted from every channel (signed operands for v_dot4_i32_i8; the 128 * sum w goes through q_bias), out-of-frame pixels
// hold x_zp - 128 (QLinearConv pads with the zero point).  The 64 x 49 weight dwords sit in LDS too and are read as
// broadcasts.  One thread = one output pixel x all 64 channels: its 7 x 7 window is 49 dwords in registers.
constexpr int SQ_TH = 8, SQ_TW = 32, SQ_PH = 2 * SQ_TH + 5, SQ_PW = 2 * SQ_TW + 5;

__global__ void __launch_bounds__(256)
    stem_q_kernel(const uint8_t* __restrict__ bgr, int H, int W, const uint8_t* __restrict__ qlut, int x_zp, const int32_t* __restrict__ wq,
                  const int32_t* __restrict__ q_bias, const float* __restrict__ q_mult, int y_zp, uint8_t* __restrict__ out, int SH, int SW) {
    __shared__ int patch[SQ_PH * SQ_PW];
    __shared__ __attribute__((aligned(16))) int wsm[64 * 52];  // [channel][49 taps + 3 pad]: rows of 13 x 16 bytes
    __shared__ uint8_t lut[768];
    const int tid = threadIdx.x;
    const int oy0 = blockIdx.y * SQ_TH, ox0 = blockIdx.x * SQ_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    for (int i = tid; i < 768; i += 256) lut[i] = qlut[i];
    for (int i = tid; i < 64 * 52; i += 256) {
        const int o = i / 52, t = i - o * 52;
        wsm[i] = t < 49 ? wq[o * 49 + t] : 0;
    }
    __syncthreads();
    const int pad = (((x_zp - 128) & 0xff) * 0x010101);  // (r, g, b) = x_zp - 128, 4th byte 0
    for (int i = tid; i < SQ_PH * SQ_PW; i += 256) {
        const int r = i / SQ_PW, q = i - r * SQ_PW;
        const int iy = iy0 + r, ix = ix0 + q;
        int v = pad;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const uint8_t* p = bgr + ((size_t)iy * W + ix) * 3;
            const int rr = lut[p[2]] ^ 0x80, gg = lut[256 + p[1]] ^ 0x80, bb = lut[512 + p[0]] ^ 0x80;  // RGB planes of the model
            v = rr | (gg << 8) | (bb << 16);
        }
        patch[i] = v;
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31;
    const int oy = oy0 + ty, ox = ox0 + tx;
    const float q_yzpf = (float)y_zp, q_lo = -q_yzpf, q_hi = 255.f - q_yzpf;
    int xw[49];
#pragma unroll
    for (int ky = 0; ky < 7; ky++)
#pragma unroll
        for (int kx = 0; kx < 7; kx++) xw[ky * 7 + kx] = patch[(2 * ty + ky) * SQ_PW + 2 * tx + kx];
    if (oy >= SH || ox >= SW) return;
    uint8_t* o = out + ((size_t)oy * SW + ox) * 64;
#pragma unroll 1
    for (int c0 = 0; c0 < 64; c0 += 16) {
        unsigned pk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int* wr = wsm + (c0 + c) * 52;
            int acc = q_bias[c0 + c];
#pragma unroll
            for (int t = 0; t < 49; t++) acc = __builtin_amdgcn_sdot4(xw[t], wr[t], acc, false);
            pk[c >> 2] = q_pack(q_requant_c(acc, q_mult[c0 + c], q_lo, q_hi) + q_yzpf, c & 3, pk[c >> 2]);
        }
        *reinterpret_cast<uint4*>(o + c0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}

// ---- max-pool 3x3/2 pad 1 on u8, 16 channels per thread; output channels beyond C are zero (channel padding to the K step
//      of the i8 GEMM) ----
__global__ void __launch_bounds__(256) maxpool_q_kernel(const uint8_t* __restrict__ in, int H, int W, int C, uint8_t* __restrict__ out, int OH,
                                                        int OW, int CP) {
    const int cvn = CP / 16;
    const size_t total = (size_t)OH * OW * cvn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const size_t p = i / cvn;
        const int ox = (int)(p % OW), oy = (int)(p / OW);
        uint4 m = make_uint4(0, 0, 0, 0);
        if (cv * 16 < C) {
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const int y = 2 * oy - 1 + dy;
                if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    const int x = 2 * ox - 1 + dx;
                    if ((unsigned)x >= (unsigned)W) continue;
                    const uint4 v = *reinterpret_cast<const uint4*>(in + ((size_t)y * W + x) * C + cv * 16);
                    auto mx = [](unsigned a, unsigned b) {  // byte-wise unsigned max
                        unsigned r = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const unsigned ab = (a >> (8 * k)) & 0xff, bb = (b >> (8 * k)) & 0xff;
                            r |= (ab > bb ? ab : bb) << (8 * k);
                        }
                        return r;
                    };
                    m.x = mx(m.x, v.x); m.y = mx(m.y, v.y); m.z = mx(m.z, v.z); m.w = mx(m.w, v.w);
                }
            }
        }
        *reinterpret_cast<uint4*>(out + p * CP + cv * 16) = m;
    }
}

// ---- OIHW s8 -> OHWI s8, input channels zero-padded to IP, rows O..OP-1 zero; row sums ----
__global__ void __launch_bounds__(256) repack_q_kernel(const int8_t* __restrict__ src, int8_t* __restrict__ dst, int32_t* __restrict__ wsum, int O,
                                                       int I, int KH, int KW, int OP, int IP) {
    const int o = blockIdx.x;
    const int taps = KH * KW;
    int part = 0;
    for (int i = threadIdx.x; i < taps * IP; i += 256) {
        const int t = i / IP, c = i - t * IP;
        int8_t v = 0;
        if (o < O && c < I) v = src[((size_t)o * I + c) * taps + t];
        dst[(size_t)o * taps * IP + i] = v;
        part += v;
    }
    __shared__ int red[256];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) wsum[o] = red[0];
}

__global__ void __launch_bounds__(256) u8_nhwc_to_planar_kernel(const uint8_t* __restrict__ in, int H, int W, int C, float* __restrict__ out) {
    const size_t total = (size_t)H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        out[(size_t)c * H * W + p] = (float)in[i];
    }
}

unsigned grid_q(size_t work) {
    size_t b = (work + 255) / 256;
    return (unsigned)(b > 256 * 32 ? 256 * 32 : (b ? b : 1));
}

}  // namespace

hipError_t launch_stem_q(const uint8_t* bgr, int H, int W, const uint8_t* qlut, int x_zp, const int32_t* wq, const int32_t* q_bias,
                         const float* q_mult, int y_zp, uint8_t* out, int SH, int SW, hipStream_t s) {
    const dim3 grid((unsigned)((SW + SQ_TW - 1) / SQ_TW), (unsigned)((SH + SQ_TH - 1) / SQ_TH));
    hipLaunchKernelGGL(stem_q_kernel, grid, dim3(256), 0, s, bgr, H, W, qlut, x_zp, wq, q_bias, q_mult, y_zp, out, SH, SW);
    return hipGetLastError();
}

hipError_t launch_maxpool_q(const uint8_t* in, int H, int W, int C, uint8_t* out, int OH, int OW, int CP, hipStream_t s) {
    if ((C & 15) || (CP & 15) || CP < C) return hipErrorInvalidValue;
    hipLaunchKernelGGL(maxpool_q_kernel, dim3(grid_q((size_t)OH * OW * (CP / 16))), dim3(256), 0, s, in, H, W, C, out, OH, OW, CP);
    return hipGetLastError();
}

hipError_t launch_repack_q(const int8_t* src, int8_t* dst, int32_t* wsum, int O, int I, int KH, int KW, int OP, int IP, hipStream_t s) {
    if (OP < O || IP < I) return hipErrorInvalidValue;
    hipLaunchKernelGGL(repack_q_kernel, dim3((unsigned)OP), dim3(256), 0, s, src, dst, wsum, O, I, KH, KW, OP, IP);
    return hipGetLastError();
}

hipError_t launch_u8_nhwc_to_planar(const uint8_t* in, int H, int W, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(u8_nhwc_to_planar_kernel, dim3(grid_q((size_t)H * W * C)), dim3(256), 0, s, in, H, W, C, out);
    return hipGetLastError();
}

}  // namespace infur
