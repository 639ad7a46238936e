// prepost.hip -- the HBM-bound, bit-exact stages either side of the conv stack.
// Compiled with -ffp-contract=off: every f32 expression here must round exactly like the
// CPU oracle's (oracle/infur_oracle.c), which in turn follows the reference's scalar Rust.
//
//   scale_bgr            Scale::advance resize           infur/src/processing.rs:246-278
//   pack_normalize       ImageSession::forward pre-proc  infur/src/predict_onnx.rs:103-137
//   upsample_planar      the model's final Resize(linear) node (inside session.run, :138)
//   colorcode_planar     ColorCode::advance              infur/src/decode_predict.rs:53-79
//   upsample_argmax_shade  fusion of the last two: reads the 2.7 MB output-stride-8 logits
//                        instead of writing + re-reading 174 MB of full-resolution logits.
#include "kernels.h"

namespace infur {

// ---------------------------------------------------------------------------------------
// Scale.  One thread per output pixel (3 bytes).  Source index rules are the oracle's:
//   nearest : src = trunc(0.5*s + s*dst) in f64, s = src_len/dst_len, clamp to src_len-1
//   bilinear: p = (dst+0.5)*s - 0.5 in f32 clamped to [0,src_len-1]; lerp x then y; round half up
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    scale_nearest_kernel(const uint8_t* __restrict__ in, int W, int H, uint8_t* __restrict__ out,
                         int OW, int OH) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= OW || y >= OH) return;
    const double sx = (double)W / (double)OW, sy = (double)H / (double)OH;
    const double px = sx * 0.5 + sx * (double)x;
    const double py = sy * 0.5 + sy * (double)y;
    unsigned ix = (unsigned)px, iy = (unsigned)py;
    if (ix > (unsigned)W - 1) ix = W - 1;
    if (iy > (unsigned)H - 1) iy = H - 1;
    const uint8_t* s = in + ((size_t)iy * W + ix) * 3;
    uint8_t* d = out + ((size_t)y * OW + x) * 3;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
}

__device__ __forceinline__ void bilinear_coord(int i, int src, int dst, int& a, int& b, float& f) {
    const float scale = (float)src / (float)dst;
    float p = ((float)i + 0.5f) * scale - 0.5f;
    if (p < 0.0f) p = 0.0f;
    const float lim = (float)(src - 1);
    if (p > lim) p = lim;
    a = (int)p;
    b = a + 1 < src ? a + 1 : src - 1;
    f = p - (float)a;
}

__global__ void __launch_bounds__(256)
    scale_bilinear_kernel(const uint8_t* __restrict__ in, int W, int H, uint8_t* __restrict__ out,
                          int OW, int OH) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= OW || y >= OH) return;
    int x0, x1, y0, y1;
    float fx, fy;
    bilinear_coord(x, W, OW, x0, x1, fx);
    bilinear_coord(y, H, OH, y0, y1, fy);
    const uint8_t* r0 = in + (size_t)y0 * W * 3;
    const uint8_t* r1 = in + (size_t)y1 * W * 3;
    uint8_t* d = out + ((size_t)y * OW + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float p00 = r0[3 * x0 + c], p01 = r0[3 * x1 + c];
        const float p10 = r1[3 * x0 + c], p11 = r1[3 * x1 + c];
        const float top = p00 + (p01 - p00) * fx;
        const float bot = p10 + (p11 - p10) * fx;
        const float v = top + (bot - top) * fy;
        float r = floorf(v + 0.5f);
        r = fminf(fmaxf(r, 0.0f), 255.0f);
        d[c] = (uint8_t)r;
    }
}

hipError_t launch_scale_bgr(const uint8_t* in, int W, int H, uint8_t* out, int OW, int OH,
                            int mode, hipStream_t s) {
    dim3 grid((OW + 63) / 64, (OH + 3) / 4);
    if (mode == 0)
        hipLaunchKernelGGL(scale_nearest_kernel, grid, dim3(256), 0, s, in, W, H, out, OW, OH);
    else
        hipLaunchKernelGGL(scale_bilinear_kernel, grid, dim3(256), 0, s, in, W, H, out, OW, OH);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// pack_normalize: BGR u8 HWC -> RGB f32 CHW through the 3x256 LUT.  One thread = 4 pixels
// (12 input bytes as three aligned dwords, three float4 plane stores).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    pack_normalize_kernel(const uint8_t* __restrict__ bgr, size_t npix, const float* __restrict__ lut,
                          float* __restrict__ chw) {
    __shared__ float slut[768];
    for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
    __syncthreads();
    const size_t nquad = npix >> 2;
    const bool aligned = ((reinterpret_cast<uintptr_t>(bgr) | reinterpret_cast<uintptr_t>(chw)) & 15) == 0 &&
                         (npix & 3) == 0;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nquad + (aligned ? 0 : 1);
         q += (size_t)gridDim.x * 256) {
        if (aligned) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(bgr + q * 12);
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
            // bytes: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
            float4 R, G, B;
            B.x = slut[512 + (w0 & 255)];
            G.x = slut[256 + ((w0 >> 8) & 255)];
            R.x = slut[(w0 >> 16) & 255];
            B.y = slut[512 + (w0 >> 24)];
            G.y = slut[256 + (w1 & 255)];
            R.y = slut[(w1 >> 8) & 255];
            B.z = slut[512 + ((w1 >> 16) & 255)];
            G.z = slut[256 + (w1 >> 24)];
            R.z = slut[w2 & 255];
            B.w = slut[512 + ((w2 >> 8) & 255)];
            G.w = slut[256 + ((w2 >> 16) & 255)];
            R.w = slut[w2 >> 24];
            *reinterpret_cast<float4*>(chw + q * 4) = R;
            *reinterpret_cast<float4*>(chw + npix + q * 4) = G;
            *reinterpret_cast<float4*>(chw + 2 * npix + q * 4) = B;
        } else {
            // ragged / unaligned sizes: same values, scalar accesses
            const size_t lo = q * 4, hi = lo + 4 < npix ? lo + 4 : npix;
            for (size_t i = lo; i < hi; i++) {
                chw[i] = slut[bgr[3 * i + 2]];
                chw[npix + i] = slut[256 + bgr[3 * i + 1]];
                chw[2 * npix + i] = slut[512 + bgr[3 * i + 0]];
            }
        }
    }
}

hipError_t launch_pack_normalize(const uint8_t* bgr, int W, int H, const float* lut, float* chw,
                                 hipStream_t s) {
    const size_t npix = (size_t)W * H;
    size_t blocks = (npix / 4 + 256) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bgr, npix, lut, chw);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// display conversion of the (scaled) frame, infur/src/app.rs:132-144:
// Color32::from_rgb(cs[2], cs[1], cs[0]) per pixel -> [r, g, b, 255].  4 pixels per thread.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    bgr_to_rgba_kernel(const uint8_t* __restrict__ bgr, size_t npix, uint32_t* __restrict__ rgba) {
    const size_t nquad = npix >> 2;
    const bool aligned = ((reinterpret_cast<uintptr_t>(bgr) & 3) | (reinterpret_cast<uintptr_t>(rgba) & 15)) == 0;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nquad + 1; q += (size_t)gridDim.x * 256) {
        if (aligned && q < nquad) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(bgr + q * 12);
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];  // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
            uint4 o;
            o.x = 0xff000000u | ((w0 & 0xffu) << 16) | (w0 & 0xff00u) | ((w0 >> 16) & 0xffu);
            o.y = 0xff000000u | ((w0 >> 24) << 16) | ((w1 & 0xffu) << 8) | ((w1 >> 8) & 0xffu);
            o.z = 0xff000000u | (((w1 >> 16) & 0xffu) << 16) | ((w1 >> 24) << 8) | (w2 & 0xffu);
            o.w = 0xff000000u | (((w2 >> 8) & 0xffu) << 16) | (((w2 >> 16) & 0xffu) << 8) | (w2 >> 24);
            *reinterpret_cast<uint4*>(rgba + q * 4) = o;
        } else {
            const size_t lo = q * 4, hi = lo + 4 < npix ? lo + 4 : npix;
            for (size_t i = lo; i < hi; i++)
                rgba[i] = 0xff000000u | ((uint32_t)bgr[3 * i] << 16) | ((uint32_t)bgr[3 * i + 1] << 8) | bgr[3 * i + 2];
        }
    }
}

hipError_t launch_bgr_to_rgba(const uint8_t* bgr, int W, int H, uint32_t* rgba, hipStream_t s) {
    const size_t npix = (size_t)W * H;
    size_t blocks = (npix / 4 + 256) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bgr_to_rgba_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bgr, npix, rgba);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// NHWC -> planar f32 (low-res logits: 21 x 135 x 240 floats; debug read-back of activations)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void nhwc_to_planar_kernel(const T* __restrict__ in, int HW, int C, float* __restrict__ out) {
    const size_t total = (size_t)HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / HW);
        const size_t p = i - (size_t)c * HW;
        out[i] = (float)in[p * C + c];
    }
}

hipError_t launch_nhwc_to_planar(const void* in, int f16, int H, int W, int C, float* out, hipStream_t s) {
    const size_t total = (size_t)H * W * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (f16)
        hipLaunchKernelGGL(nhwc_to_planar_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)in, H * W, C, out);
    else
        hipLaunchKernelGGL(nhwc_to_planar_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)in, H * W, C, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Bilinear up-sample.  Coordinates follow the oracle's upsample_table():
//   scale = out_len/in_len; src = out_len > 1 ? (dst+0.5)/scale - 0.5 : 0; clamp [0,in_len-1]
//   i1 = trunc(src), i2 = min(i1+1, in_len-1); d1 = |src-i1|, d2 = |src-i2| (0.5/0.5 if i1==i2)
//   v = dx2*dy2*X11 + dx1*dy2*X21 + dx2*dy1*X12 + dx1*dy1*X22  (left to right, no FMA)
// HIP's f32 division is correctly rounded by default (no -ffast-math here), so the
// coordinates match the host's bit for bit.
// ---------------------------------------------------------------------------------------
struct Lerp {
    int i1, i2;
    float d1, d2;
};

__device__ __forceinline__ Lerp lerp_coord(int i, int in_len, int out_len) {
    const float scale = (float)out_len / (float)in_len;
    float src = out_len > 1 ? ((float)i + 0.5f) / scale - 0.5f : 0.0f;
    if (src < 0.0f) src = 0.0f;
    const float lim = (float)(in_len - 1);
    if (src > lim) src = lim;
    Lerp t;
    t.i1 = (int)src;
    if (t.i1 > in_len - 1) t.i1 = in_len - 1;
    t.i2 = t.i1 + 1 < in_len ? t.i1 + 1 : in_len - 1;
    if (t.i1 == t.i2) {
        t.d1 = 0.5f;
        t.d2 = 0.5f;
    } else {
        t.d1 = fabsf(src - (float)t.i1);
        t.d2 = fabsf(src - (float)t.i2);
    }
    return t;
}

__device__ __forceinline__ float bilerp(float X11, float X21, float X12, float X22, float dx1,
                                        float dx2, float dy1, float dy2) {
    float v = dx2 * dy2 * X11;
    v = v + dx1 * dy2 * X21;
    v = v + dx2 * dy1 * X12;
    v = v + dx1 * dy1 * X22;
    return v;
}

// Quantised models whose file resizes the u8 logits BEFORE it dequantises them (QLinearConv -> Resize -> DequantizeLinear, what
// onnxruntime's QOperator quantiser writes when Resize is on its list): the low-res tensor then holds the u8 CODES as floats, the
// interpolation is ONNX Runtime's UpsampleBilinear<uint8_t> -- the float expression above, static_cast to uint8_t = truncation --
// and DequantizeLinear follows per output value: (trunc(v) - zp) * scale.  on == 0: the interpolated value itself.
__device__ __forceinline__ float up_post(const float v, const UpQuant q) { return q.on ? (truncf(v) - q.zp) * q.scale : v; }

// one thread per output pixel, loops over classes; writes planar [K][OH][OW]
__global__ void __launch_bounds__(256)
    upsample_planar_kernel(const float* __restrict__ low, int LH, int LW, int K, float* __restrict__ out,
                           int OH, int OW, const UpQuant uq) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= OW || y >= OH) return;
    const Lerp tx = lerp_coord(x, LW, OW), ty = lerp_coord(y, LH, OH);
    const float* p11 = low + ((size_t)ty.i1 * LW + tx.i1) * K;
    const float* p21 = low + ((size_t)ty.i1 * LW + tx.i2) * K;
    const float* p12 = low + ((size_t)ty.i2 * LW + tx.i1) * K;
    const float* p22 = low + ((size_t)ty.i2 * LW + tx.i2) * K;
    const size_t plane = (size_t)OH * OW;
    float* o = out + (size_t)y * OW + x;
    for (int k = 0; k < K; k++)
        o[k * plane] = up_post(bilerp(p11[k], p21[k], p12[k], p22[k], tx.d1, tx.d2, ty.d1, ty.d2), uq);
}


// ---------------------------------------------------------------------------------------
// ColorCode.  decode_predict.rs:67-78: k_max = 0, c_max = 0.0, strict '>' in class order;
// alpha = (c_max * 255.0) as u8 (saturating, truncating); colour = LUT[k_max % 20][alpha]
// where the LUT holds epaint's premultiplied bytes (built on the host, infur_capi.cpp).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t shade(int k_max, float c_max, const uint32_t* __restrict__ lut) {
    const float a = c_max * 255.0f;            // c_max >= 0 and never NaN by construction
    const int ai = a >= 255.0f ? 255 : (int)a;  // Rust `as u8`: saturate, truncate
    return lut[(k_max % 20) * 256 + ai];
}

__global__ void __launch_bounds__(256)
    colorcode_planar_kernel(const float* __restrict__ khw, int K, size_t HW, const uint32_t* __restrict__ lut,
                            uint32_t* __restrict__ rgba) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (size_t)gridDim.x * 256) {
        int k_max = 0;
        float c_max = 0.0f;
        for (int k = 0; k < K; k++) {
            const float c = khw[(size_t)k * HW + p];
            if (c > c_max) {
                k_max = k;
                c_max = c;
            }
        }
        rgba[p] = shade(k_max, c_max, lut);
    }
}

hipError_t launch_colorcode_planar(const float* khw, int K, int H, int W, const uint32_t* lut,
                                   uint32_t* rgba, hipStream_t s) {
    const size_t HW = (size_t)H * W;
    size_t blocks = (HW + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(colorcode_planar_kernel, dim3((unsigned)blocks), dim3(256), 0, s, khw, K, HW, lut, rgba);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// LDS-staged forms of the two up-sampling kernels (the ones that run: the scalar kernels above / below are the
// fallback for class counts > 24 or down-sampling ratios whose footprint does not fit).
//
// A workgroup owns a 64 x 8 tile of OUTPUT pixels.  Bilinear source indices are monotone in the output index, so
// the low-res pixels the tile can touch are the rectangle [i1(first row/col) .. i2(last row/col)]: at the network's
// stride-8 ratio 10 x 3 pixels x 21 classes = 2.5 KB.  The rectangle is staged ONCE, coalesced (a low-res row
// segment is contiguous in NHWC), into LDS as [pixel][24 floats]; every output pixel then reads its four
// neighbours as 6 ds_read_b128 each instead of 84 scalar global loads at stride 21 (the round-1 kernel:
// 331 GB/s, 5.7x the algorithmic HBM reads).  Same expression tree per class, same class order, strict '>' --
// bit-identical to upsample_planar -> colorcode_planar (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------
constexpr int UP_TW = 64, UP_TH = 16, UP_KP = 24;

struct UpTile {
    int r0, c0, nc;
    const Lerp* tx;  // [UP_TW] column coordinates of the tile (LDS)
    const Lerp* ty;  // [UP_TH] row coordinates
    float* pix;      // staged low-res pixels [rows][nc][UP_KP]
};

// The coordinate of every output column / row of the tile is computed ONCE (its two correctly rounded f32 divisions
// are ~25 instructions each) and shared through LDS; staging maps threads as (pixel, class) so it needs no integer
// division, and issues all of a thread's loads before the first LDS store.
__device__ __forceinline__ UpTile stage_lowres_tile(const float* __restrict__ low, int LH, int LW, int K, int OH, int OW,
                                                    float* __restrict__ smem) {
    Lerp* tx = reinterpret_cast<Lerp*>(smem);
    Lerp* ty = tx + UP_TW;
    float* tile = smem + (UP_TW + UP_TH) * (sizeof(Lerp) / sizeof(float));
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * UP_TW, y0 = blockIdx.y * UP_TH;
    if (tid < UP_TW)
        tx[tid] = lerp_coord(min(x0 + tid, OW - 1), LW, OW);
    else if (tid < UP_TW + UP_TH)
        ty[tid - UP_TW] = lerp_coord(min(y0 + tid - UP_TW, OH - 1), LH, OH);
    __syncthreads();
    UpTile t;
    t.tx = tx;
    t.ty = ty;
    t.pix = tile;
    t.c0 = tx[0].i1;
    t.r0 = ty[0].i1;
    t.nc = tx[UP_TW - 1].i2 - t.c0 + 1;  // source indices are monotone: the last column / row bound the rectangle
    const int np = (ty[UP_TH - 1].i2 - t.r0 + 1) * t.nc;
    const int k = tid & 31, pl = tid >> 5;  // 8 low-res pixels per pass, lanes along the classes
    for (int base = 0; base < np; base += 32) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = base + 8 * u + pl;
            const int r = p / t.nc, c = p - r * t.nc;  // (nc is small: this is the only division left, once per 8 pixels)
            v[u] = (p < np && k < K) ? low[((size_t)(t.r0 + r) * LW + t.c0 + c) * K + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = base + 8 * u + pl;
            if (p < np && k < UP_KP) tile[p * UP_KP + k] = v[u];
        }
    }
    __syncthreads();
    return t;
}

// Both kernels: a wave owns FOUR CONSECUTIVE output rows of the tile.  At the network's 8x ratio consecutive rows
// nearly always share their pair of source rows, so the 4 x NQ neighbour quads (ds_read_b128) stay in registers and are
// re-read only when the (wave-uniform) row pair changes: a quarter of the LDS traffic of the round-2 first form.  NQ =
// ceil(K / 4) is a template argument and every quad is evaluated whole -- the pad classes of a staged slot are zeros,
// their value is +0 and can never pass the strict '>' against c_max >= 0 -- so the body is straight-line code (the
// run-time `k < K` guards of the first form compiled to ds_read_b32 + a branch per class: 22 us at 1080p).  Class pairs
// go through 2-vectors so that the multiplies AND the adds are v_pk_*_f32; per class the expression tree is unchanged.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NQ>
__device__ __forceinline__ void load_quads(const float* ra, const float* rb, int ca, int cb, float4 (&v11)[NQ],
                                           float4 (&v21)[NQ], float4 (&v12)[NQ], float4 (&v22)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        v11[q] = *reinterpret_cast<const float4*>(ra + ca + 4 * q);
        v21[q] = *reinterpret_cast<const float4*>(ra + cb + 4 * q);
        v12[q] = *reinterpret_cast<const float4*>(rb + ca + 4 * q);
        v22[q] = *reinterpret_cast<const float4*>(rb + cb + 4 * q);
    }
}

// classes (k, k+1) of one pixel: ((w11*X11 + w21*X21) + w12*X12) + w22*X22 with w11 = dx2*dy2 ... exactly bilerp()
__device__ __forceinline__ f32x2 bilerp2(f32x2 X11, f32x2 X21, f32x2 X12, f32x2 X22, float w11, float w21, float w12,
                                         float w22) {
    f32x2 v = w11 * X11;
    v = v + w21 * X21;
    v = v + w12 * X12;
    v = v + w22 * X22;
    return v;
}

template <int NQ>
__global__ void __launch_bounds__(256)
    upsample_argmax_shade_lds_kernel(const float* __restrict__ low, int LH, int LW, int K,
                                     const uint32_t* __restrict__ lut, uint32_t* __restrict__ rgba, int OH, int OW, const UpQuant uq) {
    extern __shared__ __attribute__((aligned(16))) float up_smem[];
    const UpTile t = stage_lowres_tile(low, LH, LW, K, OH, OW, up_smem);
    const int xl = threadIdx.x & 63;
    const int x = blockIdx.x * UP_TW + xl;
    if (x >= OW) return;
    const Lerp tx = t.tx[xl];
    const int ca = (tx.i1 - t.c0) * UP_KP, cb = (tx.i2 - t.c0) * UP_KP;
    float4 v11[NQ], v21[NQ], v12[NQ], v22[NQ];
    int pi1 = -1, pi2 = -1;
#pragma unroll
    for (int rr = 0; rr < UP_TH / 4; rr++) {
        const int yl = 4 * (threadIdx.x >> 6) + rr;
        const int y = blockIdx.y * UP_TH + yl;
        if (y >= OH) break;
        const Lerp ty = t.ty[yl];
        const int i1 = __builtin_amdgcn_readfirstlane(ty.i1), i2 = __builtin_amdgcn_readfirstlane(ty.i2);
        if (i1 != pi1 || i2 != pi2) {
            load_quads<NQ>(t.pix + (i1 - t.r0) * t.nc * UP_KP, t.pix + (i2 - t.r0) * t.nc * UP_KP, ca, cb, v11, v21, v12, v22);
            pi1 = i1;
            pi2 = i2;
        }
        const float w11 = tx.d2 * ty.d2, w21 = tx.d1 * ty.d2, w12 = tx.d2 * ty.d1, w22 = tx.d1 * ty.d1;
        int k_max = 0;
        float c_max = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const f32x2 lo = bilerp2(f32x2{v11[q].x, v11[q].y}, f32x2{v21[q].x, v21[q].y}, f32x2{v12[q].x, v12[q].y},
                                     f32x2{v22[q].x, v22[q].y}, w11, w21, w12, w22);
            const f32x2 hi = bilerp2(f32x2{v11[q].z, v11[q].w}, f32x2{v21[q].z, v21[q].w}, f32x2{v12[q].z, v12[q].w},
                                     f32x2{v22[q].z, v22[q].w}, w11, w21, w12, w22);
            const float c[4] = {up_post(lo.x, uq), up_post(lo.y, uq), up_post(hi.x, uq), up_post(hi.y, uq)};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (c[e] > c_max) {
                    k_max = 4 * q + e;
                    c_max = c[e];
                }
            }
        }
        rgba[(size_t)y * OW + x] = shade(k_max, c_max, lut);
    }
}

template <int NQ>
__global__ void __launch_bounds__(256)
    upsample_planar_lds_kernel(const float* __restrict__ low, int LH, int LW, int K, float* __restrict__ out, int OH, int OW, const UpQuant uq) {
    extern __shared__ __attribute__((aligned(16))) float up_smem[];
    const UpTile t = stage_lowres_tile(low, LH, LW, K, OH, OW, up_smem);
    const int xl = threadIdx.x & 63;
    const int x = blockIdx.x * UP_TW + xl;
    if (x >= OW) return;
    const Lerp tx = t.tx[xl];
    const int ca = (tx.i1 - t.c0) * UP_KP, cb = (tx.i2 - t.c0) * UP_KP;
    const size_t plane = (size_t)OH * OW;
    float4 v11[NQ], v21[NQ], v12[NQ], v22[NQ];
    int pi1 = -1, pi2 = -1;
#pragma unroll
    for (int rr = 0; rr < UP_TH / 4; rr++) {
        const int yl = 4 * (threadIdx.x >> 6) + rr;
        const int y = blockIdx.y * UP_TH + yl;
        if (y >= OH) break;
        const Lerp ty = t.ty[yl];
        const int i1 = __builtin_amdgcn_readfirstlane(ty.i1), i2 = __builtin_amdgcn_readfirstlane(ty.i2);
        if (i1 != pi1 || i2 != pi2) {
            load_quads<NQ>(t.pix + (i1 - t.r0) * t.nc * UP_KP, t.pix + (i2 - t.r0) * t.nc * UP_KP, ca, cb, v11, v21, v12, v22);
            pi1 = i1;
            pi2 = i2;
        }
        const float w11 = tx.d2 * ty.d2, w21 = tx.d1 * ty.d2, w12 = tx.d2 * ty.d1, w22 = tx.d1 * ty.d1;
        float* o = out + (size_t)y * OW + x;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const f32x2 lo = bilerp2(f32x2{v11[q].x, v11[q].y}, f32x2{v21[q].x, v21[q].y}, f32x2{v12[q].x, v12[q].y},
                                     f32x2{v22[q].x, v22[q].y}, w11, w21, w12, w22);
            const f32x2 hi = bilerp2(f32x2{v11[q].z, v11[q].w}, f32x2{v21[q].z, v21[q].w}, f32x2{v12[q].z, v12[q].w},
                                     f32x2{v22[q].z, v22[q].w}, w11, w21, w12, w22);
            const float c[4] = {up_post(lo.x, uq), up_post(lo.y, uq), up_post(hi.x, uq), up_post(hi.y, uq)};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (4 * q + e < K) o[(size_t)(4 * q + e) * plane] = c[e];
        }
    }
}

// NQ = ceil(K / 4) -> the instantiation (K <= UP_KP = 24 is checked by up_tile_lds_bytes)
#define UP_DISPATCH_NQ(KERNEL, K, ...)                                              \
    switch (((K) + 3) / 4) {                                                        \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                  \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                  \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                  \
        case 4: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;                  \
        case 5: hipLaunchKernelGGL(KERNEL<5>, __VA_ARGS__); break;                  \
        default: hipLaunchKernelGGL(KERNEL<6>, __VA_ARGS__); break;                 \
    }

// LDS bytes of the staged rectangle for this geometry, or 0 when the scalar fallback must run (more classes than
// the padded pixel slot holds, or a ratio whose footprint is too big to be worth staging)
static size_t up_tile_lds_bytes(int LH, int LW, int K, int OH, int OW) {
    if (K > UP_KP || K <= 0 || OH <= 0 || OW <= 0) return 0;
    const size_t cols = (size_t)(((long long)UP_TW * LW + OW - 1) / OW) + 3;
    const size_t rows = (size_t)(((long long)UP_TH * LH + OH - 1) / OH) + 3;
    const size_t bytes = (UP_TW + UP_TH) * sizeof(Lerp) + rows * cols * UP_KP * sizeof(float);
    return bytes <= 48 * 1024 ? bytes : 0;
}

hipError_t launch_upsample_planar(const float* low, int LH, int LW, int K, float* out, int OH,
                                  int OW, hipStream_t s, const UpQuant uq) {
    const size_t lds = up_tile_lds_bytes(LH, LW, K, OH, OW);
    if (lds) {
        dim3 grid((OW + UP_TW - 1) / UP_TW, (OH + UP_TH - 1) / UP_TH);
        UP_DISPATCH_NQ(upsample_planar_lds_kernel, K, grid, dim3(256), lds, s, low, LH, LW, K, out, OH, OW, uq)
    } else {
        dim3 grid((OW + 63) / 64, (OH + 3) / 4);
        hipLaunchKernelGGL(upsample_planar_kernel, grid, dim3(256), 0, s, low, LH, LW, K, out, OH, OW, uq);
    }
    return hipGetLastError();
}

// fused up-sample + argmax + shade, scalar form: same expression tree as upsample_planar -> colorcode
__global__ void __launch_bounds__(256)
    upsample_argmax_shade_kernel(const float* __restrict__ low, int LH, int LW, int K,
                                 const uint32_t* __restrict__ lut, uint32_t* __restrict__ rgba, int OH, int OW, const UpQuant uq) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= OW || y >= OH) return;
    const Lerp tx = lerp_coord(x, LW, OW), ty = lerp_coord(y, LH, OH);
    const float* p11 = low + ((size_t)ty.i1 * LW + tx.i1) * K;
    const float* p21 = low + ((size_t)ty.i1 * LW + tx.i2) * K;
    const float* p12 = low + ((size_t)ty.i2 * LW + tx.i1) * K;
    const float* p22 = low + ((size_t)ty.i2 * LW + tx.i2) * K;
    int k_max = 0;
    float c_max = 0.0f;
    for (int k = 0; k < K; k++) {
        const float c = up_post(bilerp(p11[k], p21[k], p12[k], p22[k], tx.d1, tx.d2, ty.d1, ty.d2), uq);
        if (c > c_max) {
            k_max = k;
            c_max = c;
        }
    }
    rgba[(size_t)y * OW + x] = shade(k_max, c_max, lut);
}

hipError_t launch_upsample_argmax_shade(const float* low, int LH, int LW, int K, const uint32_t* lut,
                                        uint32_t* rgba, int OH, int OW, hipStream_t s, const UpQuant uq) {
    const size_t lds = up_tile_lds_bytes(LH, LW, K, OH, OW);
    if (lds) {
        dim3 grid((OW + UP_TW - 1) / UP_TW, (OH + UP_TH - 1) / UP_TH);
        UP_DISPATCH_NQ(upsample_argmax_shade_lds_kernel, K, grid, dim3(256), lds, s, low, LH, LW, K, lut, rgba, OH, OW, uq)
    } else {
        dim3 grid((OW + 63) / 64, (OH + 3) / 4);
        hipLaunchKernelGGL(upsample_argmax_shade_kernel, grid, dim3(256), 0, s, low, LH, LW, K, lut, rgba, OH, OW, uq);
    }
    return hipGetLastError();
}

}  // namespace infur
