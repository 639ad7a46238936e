// stem_pool.hip -- the network's first two nodes, fed straight from the packed BGR frame.
//
//  * stem_conv7x7: fuses the reference's pre-processing (infur/src/predict_onnx.rs:103-137:
//    BGR->RGB flip, HWC->CHW, /255, mean/std) into the first convolution (7x7 stride 2 pad 3,
//    3->64, + bias + ReLU).  The normalised value of a byte is looked up in a 3x256 table
//    computed on the host with the reference's exact f32 operation order, so the f32 input
//    tensor the reference materialises (24.9 MB at 1080p) never exists here.
//  * maxpool3x3s2: MaxPool 3x3 stride 2 pad 1 on NHWC f32.
//  * weight repacks (one-off at model load).
#include <cstdlib>

#include "hl_format.h"
#include "kernels.h"
#include "qepilogue.h"

namespace infur {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// stem as an implicit GEMM on the f32 matrix core (v_mfma_f32_32x32x2_f32, exact f32):
//   M = output pixels, N = 64 channels, K = 7*7*3 = 147 (+1 zero row), k = (ky*7 + kx)*3 + c.
// workgroup = 8 rows x 32 cols of output pixels, 4 waves; a wave owns 2 rows (two 32-pixel M blocks) x 64
// channels (two N blocks): 4 accumulator tiles, 74 k-pairs x 4 MFMAs.  The normalised 21 x 69 x 3 input
// patch (filled straight from the u8 frame through the LUT, zero padding of the NORMALISED tensor) and the
// 148 x 64 weight matrix live in LDS; a lane reads its A value at patch[pixel base + offset(k)] -- (kx, c) are
// contiguous in a patch row, so offset(k) = (k / 21) * row + k % 21 is a compile-time constant per k.
// The MFMA takes the weights as the row operand: a lane ends up with one pixel x 4 consecutive channels per
// register group (16-byte stores).
// ---------------------------------------------------------------------------------------
constexpr int ST_TH = 8, ST_TW = 32;
constexpr int ST_PH = 2 * ST_TH + 5;  // 21 input rows
constexpr int ST_PW = 2 * ST_TW + 5;  // 69 input cols
constexpr int ST_K = 147, ST_KP = 148;

typedef float f32x16s __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int stem_koff(int k) {  // float offset of tap k relative to the pixel's patch origin
    return (k / 21) * (ST_PW * 3) + (k % 21);
}

template <typename OutT>
__global__ void __launch_bounds__(256, 2)
    stem_conv7x7_kernel(const uint8_t* __restrict__ bgr, int H, int W,
                        const float* __restrict__ wt,    // [147][64]  (k, cout)
                        const float* __restrict__ bias,  // [64]
                        const float* __restrict__ lut,   // [3][256] RGB order
                        OutT* __restrict__ out, int OH, int OW) {
    __shared__ float patch[ST_PH * ST_PW * 3];
    __shared__ float wsm[ST_KP * 64];
    const int tid = threadIdx.x;
    const int oy0 = blockIdx.y * ST_TH, ox0 = blockIdx.x * ST_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

    for (int i = tid; i < ST_PH * ST_PW; i += 256) {
        const int r = i / ST_PW, q = i - r * ST_PW;
        const int iy = iy0 + r, ix = ix0 + q;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;  // zero padding of the NORMALISED tensor
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const uint8_t* p = bgr + ((size_t)iy * W + ix) * 3;
            v0 = lut[0 * 256 + p[2]];  // R
            v1 = lut[1 * 256 + p[1]];  // G
            v2 = lut[2 * 256 + p[0]];  // B
        }
        patch[i * 3 + 0] = v0;
        patch[i * 3 + 1] = v1;
        patch[i * 3 + 2] = v2;
    }
    for (int i = tid; i < ST_KP * 64 / 4; i += 256) {
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);  // row 147: the zero row that pads K to a multiple of 2
        if (i < ST_K * 64 / 4) w4 = reinterpret_cast<const float4*>(wt)[i];
        reinterpret_cast<float4*>(wsm)[i] = w4;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int px = lane & 31, half = lane >> 5;
    f32x16s acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const float* pa0 = patch + ((2 * (2 * wave + 0)) * ST_PW + 2 * px) * 3;
    const float* pa1 = patch + ((2 * (2 * wave + 1)) * ST_PW + 2 * px) * 3;
    const float* pw = wsm + half * 64 + px;
#pragma unroll
    for (int s = 0; s < ST_KP / 2; s++) {
        // lanes 0-31 take k = 2s, lanes 32-63 k = 2s + 1; tap 147 does not exist (its weights are zero): read tap 146
        constexpr int kLast = ST_K - 1;
        const int k0 = 2 * s, k1 = 2 * s + 1 > kLast ? kLast : 2 * s + 1;
        const int off = half ? stem_koff(k1) : stem_koff(k0);
        const float a0 = pa0[off], a1 = pa1[off];
        const float b0 = pw[(2 * s) * 64], b1 = pw[(2 * s) * 64 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, acc[1][1], 0, 0, 0);
    }

    // C/D layout: col = lane & 31 (pixel), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (channel of the N block)
    const int ox = ox0 + px;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int oy = oy0 + 2 * wave + i;
        if (oy >= OH || ox >= OW) continue;
        OutT* o = out + ((size_t)oy * OW + ox) * 64;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = j * 32 + 8 * g + 4 * half;
                const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
                const float v0 = fmaxf(acc[i][j][4 * g + 0] + b4.x, 0.f), v1 = fmaxf(acc[i][j][4 * g + 1] + b4.y, 0.f);
                const float v2 = fmaxf(acc[i][j][4 * g + 2] + b4.z, 0.f), v3 = fmaxf(acc[i][j][4 * g + 3] + b4.w, 0.f);
                if constexpr (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>(o + n) = make_float4(v0, v1, v2, v3);
                } else {
                    f16x4 hv = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
                    *reinterpret_cast<f16x4*>(o + n) = hv;
                }
            }
    }
}

hipError_t launch_stem_conv7x7(const uint8_t* bgr, int H, int W, const float* wt,
                               const float* bias, const float* lut, void* out, int f16, int OH, int OW,
                               hipStream_t s) {
    dim3 grid((OW + ST_TW - 1) / ST_TW, (OH + ST_TH - 1) / ST_TH);
    if (f16)
        hipLaunchKernelGGL(stem_conv7x7_kernel<_Float16>, grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut,
                           (_Float16*)out, OH, OW);
    else
        hipLaunchKernelGGL(stem_conv7x7_kernel<float>, grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut,
                           (float*)out, OH, OW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// stem + max-pool fused: the 7x7/2 convolution's output (132.7 MB at 1080p) is never written.
// A workgroup owns a 5 x 11 tile of POOLED pixels; their 3x3/2 windows cover 11 x 23 = 253 stem pixels (one halo
// row/column recomputed: 1.15x the MFMA work of the unfused stem), which are exactly 8 M-blocks of 32 -- two per
// wave, the same 2 x 2 accumulator tiles as stem_conv7x7_kernel, the same k order (bit-identical values).  The
// stem tile (+bias, ReLU) goes to LDS -- over the then idle input patch and weights -- and is max-pooled from there;
// stem pixels outside the image are skipped, as MaxPool's -inf padding does.
// ---------------------------------------------------------------------------------------
constexpr int SP_PR = 5, SP_PC = 11;
constexpr int SP_SR = 2 * SP_PR + 1, SP_SC = 2 * SP_PC + 1;  // 11 x 23 stem pixels
constexpr int SP_NPIX = SP_SR * SP_SC;                       // 253
constexpr int SP_IH = 2 * SP_SR + 5, SP_IW = 2 * SP_SC + 5;  // 27 x 51 input pixels
constexpr int SP_STAGE = 68;                                 // floats per staged stem pixel (64 + pad: conflict-free b128 writes)
// The normalised input patch of the f32 kernel lives in LDS as FOUR images -- input rows and columns split by parity -- of
// [14 rows][SP_PP floats]: a stem pixel (r, c) reads tap (ky, kx, ch) at image (ky & 1, kx & 1), row r + ky / 2, float
// 3 (c + kx / 2) + ch, so its lane-dependent part is r * SP_PP + 3 c.  With SP_PP = 133 (= 5 mod 64, and 3 * 23 = 69 = 5 mod 64:
// the tile is 23 stem pixels wide) that is 3 * (23 r + c) mod 64 = 3 x the lane's linear pixel index: the 32 lanes of an
// M block always hit 32 different banks.  (The plain [27][51][3] image gave a lane stride of 6 floats plus a row jump of an
// even number of floats: 6.2 M conflict cycles per 1080p launch in the MFMA loop, scripts/stem_conflicts.sh.)
constexpr int SP_PP = 133, SP_PROWS = (SP_IH + 1) / 2, SP_PIMG = SP_PROWS * SP_PP;
static_assert(SP_PP >= 3 * ((SP_IW + 1) / 2) && SP_PP % 64 == (3 * SP_SC) % 64, "patch row pitch");
constexpr int SP_PATCH = 4 * SP_PIMG;
constexpr int SP_LDS_FLOATS = (SP_PATCH + ST_KP * 64 + 768) > 256 * SP_STAGE ? (SP_PATCH + ST_KP * 64 + 768) : 256 * SP_STAGE;  // patch, weights, LUT | stem tile
static_assert(SP_NPIX <= 256 && SP_NPIX > 224, "the stem tile must fill 8 M-blocks of 32");

// float offset of tap k = ky * 21 + kx * 3 + ch relative to the stem pixel's base r * SP_PP + 3 c
__host__ __device__ constexpr int sp_koff(int k) {
    return (((k / 21) & 1) * 2 + (((k % 21) / 3) & 1)) * SP_PIMG + ((k / 21) >> 1) * SP_PP + (((k % 21) / 3) >> 1) * 3 + (k % 3);
}
// where input pixel (row, col) of the patch sits
__host__ __device__ constexpr int sp_pixoff(int row, int col) { return ((row & 1) * 2 + (col & 1)) * SP_PIMG + (row >> 1) * SP_PP + (col >> 1) * 3; }

// stem tile (+bias, ReLU, * acc_scale) -> LDS -> 3x3/2 max-pool -> global; shared by the f32 and the f16-rate stems.
// `smem` is the whole (now idle) operand LDS; every wave has passed a barrier since its last operand read.
// QUANT (the quantised model's stem, stem_pool16_kernel<unsigned char, false, true>): the accumulators are EXACT integers (see
// there); they are staged raw, pooled, and only the pooled maxima are requantised -- QLinearConv's requantisation is monotone in the
// accumulator (positive multiplier, round, clamp), so it commutes with the max: a quarter of the requantisations of stem-then-pool
// and the same bytes.  Output rows are 128 channels (the K step of the i8 GEMMs): 64 values + 64 zeros.
struct StemQuant {
    const int32_t* bias;  // the operator's own bias (the operand is q - x_zp: nothing to fold)
    const float* mult;    // (x_s * w_s[o]) / y_s
    int y_zp;
    int cstride = 128;  // bytes per output pixel: 128 (64 values + 64 zeros, the K step of the i8 GEMMs) or 64 (compact: the pixel-pair view)
};
template <typename OutT, bool QUANT = false>
__device__ __forceinline__ void stem_stage_and_pool(float* smem, const f32x16s (&acc)[2][2], const int (&pidx)[2], int half, float acc_scale,
                                                    const float* __restrict__ bias, OutT* __restrict__ out, int py0, int px0, int sy0, int sx0,
                                                    int SH, int SW, int PH, int PW, unsigned* __restrict__ amax, const StemQuant q = StemQuant{}) {
    const int tid = threadIdx.x;
    float* stage = smem;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float* o = stage + pidx[i] * SP_STAGE;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = j * 32 + 8 * g + 4 * half;
                if constexpr (QUANT) {
                    *reinterpret_cast<float4*>(o + n) = make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    continue;
                }
                const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
                float4 v;
                v.x = fmaxf(acc[i][j][4 * g + 0] * acc_scale + b4.x, 0.f);
                v.y = fmaxf(acc[i][j][4 * g + 1] * acc_scale + b4.y, 0.f);
                v.z = fmaxf(acc[i][j][4 * g + 2] * acc_scale + b4.z, 0.f);
                v.w = fmaxf(acc[i][j][4 * g + 3] * acc_scale + b4.w, 0.f);
                if constexpr (sizeof(OutT) == 2) {  // the unfused path stores the stem output as f16: round here, max after
                    v.x = (float)(_Float16)v.x; v.y = (float)(_Float16)v.y; v.z = (float)(_Float16)v.z; v.w = (float)(_Float16)v.w;
                }
                *reinterpret_cast<float4*>(o + n) = v;
            }
    }
    __syncthreads();

    float vmax = 0.f;
    for (int it = tid; it < SP_PR * SP_PC * 16; it += 256) {
        const int c4 = it & 15, pp = it >> 4;
        const int pr = pp / SP_PC, pc = pp - pr * SP_PC;
        const int py = py0 + pr, pxo = px0 + pc;
        if (py >= PH || pxo >= PW) continue;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int sr = 2 * pr + dy;  // stem row inside the tile; absolute row sy0 + sr
            if ((unsigned)(sy0 + sr) >= (unsigned)SH) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const int sc = 2 * pc + dx;
                if ((unsigned)(sx0 + sc) >= (unsigned)SW) continue;
                const float4 v = *reinterpret_cast<const float4*>(stage + (sr * SP_SC + sc) * SP_STAGE + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        if constexpr (QUANT) {
            const int4 b4 = *reinterpret_cast<const int4*>(q.bias + c4 * 4);
            const float4 m4 = *reinterpret_cast<const float4*>(q.mult + c4 * 4);
            const float yz = (float)q.y_zp, lo = -yz, hi = 255.f - yz;
            unsigned w = 0;
            w = q_pack(q_requant_c((int)m.x + b4.x, m4.x, lo, hi) + yz, 0, w);
            w = q_pack(q_requant_c((int)m.y + b4.y, m4.y, lo, hi) + yz, 1, w);
            w = q_pack(q_requant_c((int)m.z + b4.z, m4.z, lo, hi) + yz, 2, w);
            w = q_pack(q_requant_c((int)m.w + b4.w, m4.w, lo, hi) + yz, 3, w);
            unsigned char* o = reinterpret_cast<unsigned char*>(out) + ((size_t)py * PW + pxo) * (size_t)q.cstride + c4 * 4;
            *reinterpret_cast<unsigned*>(o) = w;
            if (q.cstride == 128) *reinterpret_cast<unsigned*>(o + 64) = 0u;  // channel padding
            continue;
        }
        if constexpr (sizeof(OutT) == 3) {  // HlTag: hi / lo planes of the three-byte format (hl_format.h), lo plane behind the hi plane
            const float x4[4] = {m.x, m.y, m.z, m.w};
            hl_f16x4 hv;
            unsigned lv;
            hl_split4(x4, hv, lv);
            const size_t e = ((size_t)py * PW + pxo) * 64 + c4 * 4;
            *reinterpret_cast<hl_f16x4*>(reinterpret_cast<_Float16*>(out) + e) = hv;
            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(out) + hl_lo_offset((size_t)PH * PW * 64) + e) = lv;
            vmax = fmaxf(vmax, fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w)));
            continue;
        }
        OutT* o = out + ((size_t)py * PW + pxo) * 64 + c4 * 4;
        if constexpr (sizeof(OutT) == 4) {
            *reinterpret_cast<float4*>(o) = m;
        } else {
            f16x4 hv = {(_Float16)m.x, (_Float16)m.y, (_Float16)m.z, (_Float16)m.w};
            *reinterpret_cast<f16x4*>(o) = hv;
        }
        vmax = fmaxf(vmax, fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w)));
    }
    if (amax) {  // range monitor of the split mode (pooled values are >= 0)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
    }
}

// ABL: timing ablations (INFUR_STEM_ABL, results wrong): 1 = no MFMA loop, 2 = no prologue loads (LDS left as is),
// 3 = no stage/pool phase
template <typename OutT, int ABL = 0>
__global__ void __launch_bounds__(256, 2)
    stem_pool_kernel(const uint8_t* __restrict__ bgr, int H, int W, const float* __restrict__ wt, const float* __restrict__ bias,
                     const float* __restrict__ lut, OutT* __restrict__ out, int SH, int SW, int PH, int PW,
                     unsigned* __restrict__ amax) {
    __shared__ __attribute__((aligned(16))) float smem[SP_LDS_FLOATS];
    float* patch = smem;
    float* wsm = smem + SP_PATCH;
    const int tid = threadIdx.x;
    const int py0 = blockIdx.y * SP_PR, px0 = blockIdx.x * SP_PC;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;  // first stem pixel of the tile (may be -1: outside)
    const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;

    // Prologue built for latency: every global load of the workgroup's inputs is issued before the first one is
    // used (frame bytes, the 3 KB look-up table, the 38 KB of weights); the table goes to LDS so that the
    // byte -> normalised value look-ups are LDS gathers instead of a second dependent trip to L2.  (With plain loops
    // the prologue was ~20 us of serial L2 round trips per workgroup against 8 us of MFMA work.)
    float* slut = smem + SP_LDS_FLOATS - 768;  // top of the allocation: overwritten only by the stem tile, after the barrier
    constexpr int NPX = (SP_IH * SP_IW + 255) / 256;  // 6 frame pixels per thread
    constexpr int NW4 = (ST_KP * 64 / 4 + 255) / 256;  // 10 float4 of weights per thread
    if constexpr (ABL != 2) {
    uint8_t pb[NPX][3];
    bool pin[NPX];
#pragma unroll
    for (int j = 0; j < NPX; j++) {
        const int i = tid + 256 * j;
        const int r = i / SP_IW, q = i - r * SP_IW;
        const int iy = iy0 + r, ix = ix0 + q;
        pin[j] = i < SP_IH * SP_IW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const uint8_t* p = bgr + ((size_t)(pin[j] ? iy : 0) * W + (pin[j] ? ix : 0)) * 3;
        pb[j][0] = p[0];
        pb[j][1] = p[1];
        pb[j][2] = p[2];
    }
    float4 w4[NW4];
#pragma unroll
    for (int j = 0; j < NW4; j++) {
        const int i = tid + 256 * j;
        w4[j] = make_float4(0.f, 0.f, 0.f, 0.f);  // row 147: the zero row that pads K to a multiple of 2
        if (i < ST_K * 64 / 4) w4[j] = reinterpret_cast<const float4*>(wt)[i];
    }
    const float l0 = lut[tid], l1 = lut[tid + 256], l2 = lut[tid + 512];
    slut[tid] = l0;
    slut[tid + 256] = l1;
    slut[tid + 512] = l2;
#pragma unroll
    for (int j = 0; j < NW4; j++) {
        const int i = tid + 256 * j;
        if (i < ST_KP * 64 / 4) reinterpret_cast<float4*>(wsm)[i] = w4[j];
    }
    __syncthreads();  // the table is in LDS
#pragma unroll
    for (int j = 0; j < NPX; j++) {
        const int i = tid + 256 * j;
        if (i >= SP_IH * SP_IW) continue;
        // zero padding of the NORMALISED tensor outside the frame
        const int pr_ = i / SP_IW, pq_ = i - pr_ * SP_IW;
        float* pp = patch + sp_pixoff(pr_, pq_);
        pp[0] = pin[j] ? slut[0 * 256 + pb[j][2]] : 0.f;  // R
        pp[1] = pin[j] ? slut[1 * 256 + pb[j][1]] : 0.f;  // G
        pp[2] = pin[j] ? slut[2 * 256 + pb[j][0]] : 0.f;  // B
    }
    __syncthreads();

    }
    const int wave = tid >> 6, lane = tid & 63;
    const int px = lane & 31, half = lane >> 5;
    f32x16s acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // a lane's two stem pixels: linear index over the 11 x 23 tile; slots 253..255 recompute pixel 252 (never used)
    int pidx[2];
    const float* pa[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        pidx[i] = (2 * wave + i) * 32 + px;
        const int p = pidx[i] < SP_NPIX ? pidx[i] : SP_NPIX - 1;
        const int r = p / SP_SC, c = p - r * SP_SC;
        pa[i] = patch + r * SP_PP + 3 * c;  // (input pixel (2r, 2c): both even -> image 0, row r, column c)
    }
    const float* pw = wsm + half * 64 + px;
#pragma unroll
    for (int s = 0; s < (ABL == 1 ? 1 : ST_KP / 2); s++) {
        constexpr int kLast = ST_K - 1;
        const int k0 = 2 * s, k1 = 2 * s + 1 > kLast ? kLast : 2 * s + 1;
        const int off = half ? sp_koff(k1) : sp_koff(k0);
        const float a0 = pa[0][off], a1 = pa[1][off];
        const float b0 = pw[(2 * s) * 64], b1 = pw[(2 * s) * 64 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();  // every wave is done with the patch and the weights: their LDS becomes the stem tile
    if constexpr (ABL == 3) {
        if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 123.456f) out[0] = (OutT)1;  // keep the MFMAs alive
        return;
    }

    stem_stage_and_pool<OutT>(smem, acc, pidx, half, 1.0f, bias, out, py0, px0, sy0, sx0, SH, SW, PH, PW, amax);
}

// ---------------------------------------------------------------------------------------
// The same fused stem + max-pool on the f16 matrix cores, for the two f16-rate modes of the conv stack
// (INFUR_DTYPE_F16: f16 operands; INFUR_DTYPE_F32_SPLIT: every operand value as an f16 hi + lo pair, three MFMAs per
// product, f32-grade result -- conv_igemm.hip).  The exact-f32 stem costs 18.9k MFMA cycles per wave and was the largest
// single item below its roofline in those modes (0.16 of 5.4 ms in the split mode); v_mfma_f32_32x32x16_f16 does the
// same contraction in 1.4k (f16) / 4.2k (split) cycles.
//
// K is laid out so that a lane's 8 consecutive k are 8 consecutive halfs of one patch row: k = ky * 24 + j with
// j = kx * 3 + c for j < 21 and three zero-weight pads per row; 7 rows = 168, padded to 176 = 11 slices of 16.  Group
// g = k / 8 is (ky, jg) = (g / 3, 8 * (g % 3)); lanes 0-31 of slice s take group 2s, lanes 32-63 group 2s + 1.  A pixel's
// fragment starts at an arbitrary 4-byte boundary (pixel stride 12 bytes), hence four ds_read_b32 (bank stride 3:
// conflict-free); the weights sit in LDS as [n][184] halfs, one ds_read_b128 per fragment.  Pad positions read real
// neighbouring patch values (finite) against zero weights.
// ---------------------------------------------------------------------------------------
constexpr int S16_PSTR = 160;  // halfs per patch row (51 px * 3 = 153, + room for the last pixel's 24-wide window)
constexpr int S16_K = 176, S16_WSTR = 184, S16_GROUPS = 22;
constexpr int S16_PATCH_H = SP_IH * S16_PSTR;  // halfs per patch plane
constexpr int S16_W_H = 64 * S16_WSTR;         // halfs per weight plane
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// QUANT: the stem of a QUANTISED model (QuantizeLinear of the normalised image + QLinearConv 7x7/2, then the u8 max-pool) on the
// same f16 MFMA, exactly: the table holds q - x_zp (integers in -255..255), the weights are the s8 values, both exact in f16; a
// product is below 2^15 and the 147-term sum below 2^23, so the f32 accumulation is exact integer arithmetic.  Out-of-frame taps are
// zero = "the zero point", QLinearConv's padding.  (quant.hip's stem_q + maxpool_q are the unfused form: 0.09 + 0.017 ms.)
template <typename OutT, bool SPLIT, bool QUANT = false>
__global__ void __launch_bounds__(256, 2)
    stem_pool16_kernel(const uint8_t* __restrict__ bgr, int H, int W, const float* __restrict__ wt, const float* __restrict__ bias,
                       const float* __restrict__ lut, OutT* __restrict__ out, int SH, int SW, int PH, int PW,
                       float a_scale, float w_scale, float acc_scale, unsigned* __restrict__ amax, const u32x4s* __restrict__ wimg,
                       const StemQuant sq = StemQuant{}) {
    constexpr int NP = SPLIT ? 2 : 1;  // operand planes: hi (, lo)
    // three-byte output (INFUR_DTYPE_F16_HL): hl_split4 requires MODE.FP16_OVFL = 1 -- a pooled value beyond 65504 clamps instead of
    // becoming hi = inf, rem = -inf
    if constexpr (sizeof(OutT) == 3) hl_set_fp16_ovfl();
    __shared__ __attribute__((aligned(16))) float smem[SP_LDS_FLOATS];
    _Float16* patch = reinterpret_cast<_Float16*>(smem);  // [NP][SP_IH][S16_PSTR]
    _Float16* wsm = patch + NP * S16_PATCH_H;              // [NP][64][S16_WSTR]
    float* slut = smem + SP_LDS_FLOATS - 768;
    static_assert((NP * (S16_PATCH_H + S16_W_H)) * 2 <= (SP_LDS_FLOATS - 768) * 4, "operands + look-up table exceed the LDS allocation");
    const int tid = threadIdx.x;
    const int py0 = blockIdx.y * SP_PR, px0 = blockIdx.x * SP_PC;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;
    if constexpr (SPLIT) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // MODE.FP16_OVFL: saturate, never inf

    // all global loads first (frame bytes, table, weights), then the LDS fills
    constexpr int NPX = (SP_IH * SP_IW + 255) / 256;
    // The weights arrive as the finished LDS image -- [NP][64][S16_WSTR] halfs, K pads zero, already f16 (hi, lo) -- built once per
    // model by stem16_pack_kernel with the very conversions this kernel used to run per workgroup: 37 scattered f32 loads, 37
    // conversions and 37 two-byte LDS writes per thread were 30 of the 79 us of the quantised 1080p stem (ablation, same box);
    // now 6 (12: split) 16-byte loads and as many ds_write_b128.
    constexpr int W_CHUNKS = NP * S16_W_H * 2 / 16;  // 16-byte chunks of the image
    constexpr int NWC = (W_CHUNKS + 255) / 256;
    static_assert((NP * S16_PATCH_H * 2) % 16 == 0 && (S16_W_H * 2) % 16 == 0, "the weight planes start on 16-byte boundaries");
    u32x4s wv[NWC];
#pragma unroll
    for (int j = 0; j < NWC; j++) {
        const int i = tid + 256 * j;
        wv[j] = i < W_CHUNKS ? wimg[i] : u32x4s{0u, 0u, 0u, 0u};
    }
    uint8_t pb[NPX][3];
    bool pin[NPX];
#pragma unroll
    for (int j = 0; j < NPX; j++) {
        const int i = tid + 256 * j;
        const int r = i / SP_IW, q = i - r * SP_IW;
        const int iy = iy0 + r, ix = ix0 + q;
        pin[j] = i < SP_IH * SP_IW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const uint8_t* p = bgr + ((size_t)(pin[j] ? iy : 0) * W + (pin[j] ? ix : 0)) * 3;
        pb[j][0] = p[0];
        pb[j][1] = p[1];
        pb[j][2] = p[2];
    }
    const float l0 = lut[tid], l1 = lut[tid + 256], l2 = lut[tid + 512];
    // zero the patch planes (row pads must be finite zeros), copy the weight image, publish the table
    for (int i = tid; i < NP * S16_PATCH_H / 2; i += 256) reinterpret_cast<unsigned*>(patch)[i] = 0u;
#pragma unroll
    for (int j = 0; j < NWC; j++) {
        const int i = tid + 256 * j;
        if (i < W_CHUNKS) reinterpret_cast<u32x4s*>(wsm)[i] = wv[j];
    }
    slut[tid] = l0;
    slut[tid + 256] = l1;
    slut[tid + 512] = l2;
    __syncthreads();
    auto put = [&](_Float16* plane0, int plane_halfs, int idx, float x) {  // x -> f16 (, hi + lo)
        const _Float16 hi = (_Float16)x;
        plane0[idx] = hi;
        if constexpr (SPLIT) plane0[plane_halfs + idx] = (_Float16)(x - (float)hi);
    };
#pragma unroll
    for (int j = 0; j < NPX; j++) {
        const int i = tid + 256 * j;
        if (i >= SP_IH * SP_IW || !pin[j]) continue;  // outside the frame: the zero padding of the NORMALISED tensor
        const int r = i / SP_IW, q = i - r * SP_IW;
        const int o = r * S16_PSTR + q * 3;
        put(patch, S16_PATCH_H, o + 0, slut[0 * 256 + pb[j][2]] * a_scale);  // R
        put(patch, S16_PATCH_H, o + 1, slut[1 * 256 + pb[j][1]] * a_scale);  // G
        put(patch, S16_PATCH_H, o + 2, slut[2 * 256 + pb[j][0]] * a_scale);  // B
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int px = lane & 31, half = lane >> 5;
    f32x16s acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    int pidx[2];
    const char* pa[2];  // byte address of the pixel's patch origin (plane 0)
#pragma unroll
    for (int i = 0; i < 2; i++) {
        pidx[i] = (2 * wave + i) * 32 + px;
        const int p = pidx[i] < SP_NPIX ? pidx[i] : SP_NPIX - 1;
        const int r = p / SP_SC, c = p - r * SP_SC;
        pa[i] = reinterpret_cast<const char*>(patch) + ((2 * r) * S16_PSTR + 6 * c) * 2;
    }
    const char* pw = reinterpret_cast<const char*>(wsm) + px * S16_WSTR * 2;
#pragma unroll
    for (int s = 0; s < S16_K / 16; s++) {
        // this lane's group: 2s (lanes 0-31) or 2s + 1 (lanes 32-63); both are compile-time per half
        const int g0 = 2 * s, g1 = 2 * s + 1 < S16_GROUPS - 1 ? 2 * s + 1 : S16_GROUPS - 1;
        const int ga = (g0 < 21 ? g0 : 20), gb = (g1 < 21 ? g1 : 20);  // groups 21+ are K padding: weights are zero there
        const int aoff = half ? ((gb / 3) * S16_PSTR * 2 + (gb % 3) * 16) : ((ga / 3) * S16_PSTR * 2 + (ga % 3) * 16);
        const int boff = (half ? g1 : g0) * 16;
        f16x8 a[2][NP], b[2][NP];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {
                const unsigned* q = reinterpret_cast<const unsigned*>(pa[i] + pl * S16_PATCH_H * 2 + aoff);
                const u32x4s v = {q[0], q[1], q[2], q[3]};
                a[i][pl] = __builtin_bit_cast(f16x8, v);
            }
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
#pragma unroll
            for (int pl = 0; pl < NP; pl++)
                b[jn][pl] = *reinterpret_cast<const f16x8*>(pw + pl * S16_W_H * 2 + jn * 32 * S16_WSTR * 2 + boff);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int jn = 0; jn < 2; jn++) {
                if constexpr (SPLIT) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[jn][0], a[i][1], acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[jn][1], a[i][0], acc[i][jn], 0, 0, 0);
                }
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[jn][0], a[i][0], acc[i][jn], 0, 0, 0);
            }
    }
    __syncthreads();  // every wave is done with the operands: their LDS becomes the stem tile
    stem_stage_and_pool<OutT, QUANT>(smem, acc, pidx, half, acc_scale, bias, out, py0, px0, sy0, sx0, SH, SW, PH, PW, amax, sq);
}

// The weight image of stem_pool16_kernel, built once per model: wt[k][n] f32 (k = ky * 21 + kx * 3 + c) -> img[plane][n][ky * 24 +
// kx * 3 + c] halfs, hi = f16(w * w_scale) (, lo = f16(w * w_scale - hi)), pads zero.  Same operations, same rounding (FP16_OVFL
// set as in the consumer) as the per-workgroup conversion it replaces: the consumer's results do not change by a bit.
template <bool SPLIT>
__global__ void __launch_bounds__(256) stem16_pack_kernel(const float* __restrict__ wt, float w_scale, _Float16* __restrict__ img) {
    constexpr int NP = SPLIT ? 2 : 1;
    if constexpr (SPLIT) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < NP * S16_W_H; i += gridDim.x * 256) img[i] = (_Float16)0.f;
    __syncthreads();  // (one block: launch_stem16_pack)
    for (int i = threadIdx.x; i < ST_K * 64; i += 256) {
        const int k = i >> 6, n = i & 63;
        const int ky = k / 21, jj = k - ky * 21;
        const float x = wt[i] * w_scale;
        const _Float16 hi = (_Float16)x;
        img[n * S16_WSTR + ky * 24 + jj] = hi;
        if constexpr (SPLIT) img[S16_W_H + n * S16_WSTR + ky * 24 + jj] = (_Float16)(x - (float)hi);
    }
}

size_t stem16_image_bytes() { return (size_t)2 * S16_W_H * 2; }

hipError_t launch_stem16_pack(const float* wt, float w_scale, int split, void* img, hipStream_t s) {
    if (split)
        hipLaunchKernelGGL(stem16_pack_kernel<true>, dim3(1), dim3(256), 0, s, wt, w_scale, (_Float16*)img);
    else
        hipLaunchKernelGGL(stem16_pack_kernel<false>, dim3(1), dim3(256), 0, s, wt, w_scale, (_Float16*)img);
    return hipGetLastError();
}

hipError_t launch_stem_pool_q(const uint8_t* bgr, int H, int W, const void* wimg, const float* lut, const int32_t* q_bias, const float* q_mult,
                              int y_zp, uint8_t* out, int cstride, int SH, int SW, int PH, int PW, hipStream_t s) {
    if (cstride != 64 && cstride != 128) return hipErrorInvalidValue;
    dim3 grid((PW + SP_PC - 1) / SP_PC, (PH + SP_PR - 1) / SP_PR);
    hipLaunchKernelGGL((stem_pool16_kernel<unsigned char, false, true>), grid, dim3(256), 0, s, bgr, H, W, (const float*)nullptr, (const float*)nullptr, lut,
                       out, SH, SW, PH, PW, 1.0f, 1.0f, 1.0f, (unsigned*)nullptr, (const u32x4s*)wimg, StemQuant{q_bias, q_mult, y_zp, cstride});
    return hipGetLastError();
}

// (wimg: the f16-rate modes' weight image, launch_stem16_pack -- nullptr only where the f32 MFMA stem runs)
hipError_t launch_stem_pool(const uint8_t* bgr, int H, int W, const float* wt, const void* wimg, const float* bias, const float* lut, void* out,
                            int mode, int SH, int SW, int PH, int PW, float a_scale, float w_scale, unsigned* amax, hipStream_t s) {
    dim3 grid((PW + SP_PC - 1) / SP_PC, (PH + SP_PR - 1) / SP_PR);
    static const int abl = getenv("INFUR_STEM_ABL") ? atoi(getenv("INFUR_STEM_ABL")) : 0;
    static const int exact = getenv("INFUR_STEM_F32") ? atoi(getenv("INFUR_STEM_F32")) : 0;  // measurement hook: f32 MFMA stem in every mode
    if ((mode == 1 || mode == 2 || mode == 5) && !(exact && mode != 5) && !wimg) return hipErrorInvalidValue;
    if (mode == 1 && !exact)
        hipLaunchKernelGGL((stem_pool16_kernel<_Float16, false>), grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (_Float16*)out, SH, SW, PH,
                           PW, 1.0f, 1.0f, 1.0f, amax, (const u32x4s*)wimg);
    else if (mode == 2 && !exact)
        hipLaunchKernelGGL((stem_pool16_kernel<float, true>), grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (float*)out, SH, SW, PH, PW,
                           a_scale, w_scale, 1.0f / (a_scale * w_scale), amax, (const u32x4s*)wimg);
    else if (mode == 5)  // the split arithmetic (three f16 MFMAs per product: 10 GFLOP of the frame), three-byte output
        hipLaunchKernelGGL((stem_pool16_kernel<HlTag, true>), grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (HlTag*)out, SH, SW, PH, PW,
                           a_scale, w_scale, 1.0f / (a_scale * w_scale), amax, (const u32x4s*)wimg);
    else if (mode == 1)
        hipLaunchKernelGGL(stem_pool_kernel<_Float16>, grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (_Float16*)out, SH, SW, PH, PW, amax);
    else if (abl == 1)
        hipLaunchKernelGGL((stem_pool_kernel<float, 1>), grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (float*)out, SH, SW, PH, PW, amax);
    else if (abl == 2)
        hipLaunchKernelGGL((stem_pool_kernel<float, 2>), grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (float*)out, SH, SW, PH, PW, amax);
    else if (abl == 3)
        hipLaunchKernelGGL((stem_pool_kernel<float, 3>), grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (float*)out, SH, SW, PH, PW, amax);
    else
        hipLaunchKernelGGL(stem_pool_kernel<float>, grid, dim3(256), 0, s, bgr, H, W, wt, bias, lut, (float*)out, SH, SW, PH, PW, amax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// maxpool 3x3 / 2, pad 1, NHWC: one thread = one output pixel x 4 channels
// ---------------------------------------------------------------------------------------
template <typename T, typename V4>
__global__ void __launch_bounds__(256)
    maxpool3x3s2_kernel(const T* __restrict__ in, int H, int W, int C, T* __restrict__ out, int OH, int OW, unsigned* __restrict__ amax) {
    float vmax = 0.f;  // the stem's outputs are >= 0 (ReLU): max of the pooled values = max |activation|
    const int c4n = C >> 2;
    const size_t total = (size_t)OH * OW * c4n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        const size_t p = i / c4n;
        const int ox = (int)(p % OW), oy = (int)(p / OW);
        float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const int iy = 2 * oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const V4 v = *reinterpret_cast<const V4*>(in + ((size_t)iy * W + ix) * C + c4 * 4);
                m0 = fmaxf(m0, (float)v[0]);
                m1 = fmaxf(m1, (float)v[1]);
                m2 = fmaxf(m2, (float)v[2]);
                m3 = fmaxf(m3, (float)v[3]);
            }
        }
        V4 r = {(T)m0, (T)m1, (T)m2, (T)m3};
        *reinterpret_cast<V4*>(out + p * C + c4 * 4) = r;
        vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(m0), fabsf(m1)), fmaxf(fabsf(m2), fabsf(m3))));
    }
    if (amax) {  // range monitor of the split mode
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
    }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

hipError_t launch_maxpool3x3s2(const void* in, int H, int W, int C, void* out, int f16, int OH, int OW,
                               unsigned* amax, hipStream_t s) {
    const size_t total = (size_t)OH * OW * (C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (f16)
        hipLaunchKernelGGL((maxpool3x3s2_kernel<_Float16, f16x4>), dim3(blocks), dim3(256), 0, s, (const _Float16*)in, H, W, C,
                           (_Float16*)out, OH, OW, amax);
    else
        hipLaunchKernelGGL((maxpool3x3s2_kernel<float, f32x4v>), dim3(blocks), dim3(256), 0, s, (const float*)in, H, W, C,
                           (float*)out, OH, OW, amax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// weight repacks
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void repack_oihw_to_ohwi_kernel(const float* __restrict__ src, T* __restrict__ dst, int O, int I, int KH,
                                           int KW) {
    const size_t total = (size_t)O * I * KH * KW;
    for (size_t d = (size_t)blockIdx.x * 256 + threadIdx.x; d < total; d += (size_t)gridDim.x * 256) {
        // d indexes dst [o][ky][kx][i]
        const int i = (int)(d % I);
        size_t r = d / I;
        const int kx = (int)(r % KW);
        r /= KW;
        const int ky = (int)(r % KH);
        const int o = (int)(r / KH);
        dst[d] = (T)src[(((size_t)o * I + i) * KH + ky) * KW + kx];  // f16: round to nearest even
    }
}

hipError_t launch_repack_oihw_to_ohwi(const float* src, void* dst, int f16, int O, int I, int KH, int KW,
                                      hipStream_t s) {
    const size_t total = (size_t)O * I * KH * KW;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (f16)
        hipLaunchKernelGGL(repack_oihw_to_ohwi_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, src, (_Float16*)dst, O, I, KH, KW);
    else
        hipLaunchKernelGGL(repack_oihw_to_ohwi_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, O, I, KH, KW);
    return hipGetLastError();
}

__global__ void repack_stem_kernel(const float* __restrict__ src, float* __restrict__ dst, int reverse_c) {
    // dst [ky][kx][c][o]  <-  src [o][c][ky][kx]   (O=64, C=3, 7x7)
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= 7 * 7 * 3 * 64) return;
    const int o = d & 63;
    int r = d >> 6;
    const int c = r % 3;
    r /= 3;
    const int kx = r % 7, ky = r / 7;
    dst[d] = src[((o * 3 + (reverse_c ? 2 - c : c)) * 7 + ky) * 7 + kx];
}

hipError_t launch_repack_stem(const float* src, float* dst, int reverse_c, hipStream_t s) {
    hipLaunchKernelGGL(repack_stem_kernel, dim3((7 * 7 * 3 * 64 + 255) / 256), dim3(256), 0, s, src, dst, reverse_c);
    return hipGetLastError();
}

// ---- weight preparation for the f32-split GEMM mode (conv_igemm.hip, SPLIT) ----
__global__ void absmax_kernel(const float* __restrict__ w, size_t n, float* __restrict__ out) {
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

hipError_t launch_absmax(const float* w, size_t n, float* out, hipStream_t s) {
    const int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, w, n, out);
    return hipGetLastError();
}

// `planes` tensors of `per` elements each, back to back: out[p] = max |w[p * per + i]| in ONE launch (blockIdx.y = plane) -- the 16 / 36 /
// 64 Winograd planes of a layer (round 6: model load in the split / HL modes issued ~950 of the single-tensor launches above)
__global__ void absmax_planes_kernel(const float* __restrict__ w, size_t per, float* __restrict__ out) {
    const float* wp = w + (size_t)blockIdx.y * per;
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(wp[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out + blockIdx.y), __float_as_uint(m));
}

hipError_t launch_absmax_planes(const float* w, size_t per, int planes, float* out, hipStream_t s) {
    if (planes < 1 || planes > 65535) return hipErrorInvalidValue;
    const int blocks = (int)((per + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(absmax_planes_kernel, dim3(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks), planes), dim3(256), 0, s, w, per, out);
    return hipGetLastError();
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// one thread per 128-byte group: 32 f32 in, 32 f16 hi + 32 f16 lo out (in place)
__global__ void split_weights_kernel(float* __restrict__ w, size_t groups, float scale) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    f32x8* p = reinterpret_cast<f32x8*>(w + g * 32);
    h16x8 hi[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x8 x = p[q] * scale;
        hi[q] = __builtin_convertvector(x, h16x8);
        lo[q] = __builtin_convertvector(x - __builtin_convertvector(hi[q], f32x8), h16x8);
    }
    h16x8* o = reinterpret_cast<h16x8*>(w + g * 32);
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] = hi[q];
#pragma unroll
    for (int q = 0; q < 4; q++) o[4 + q] = lo[q];
}

// the bf8 cross-term form (conv_igemm mode 3): 32 f32 in, in place [32 x f16 hi][32 x e5m2 of lo * 2^11][32 x e5m2 of hi].
// With max |w * scale| in [2^13, 2^14) both planes stay below 2^14 (e5m2 max 57344); clamped anyway (a weight tensor holding
// inf / NaN must not turn into a different kind of garbage here).
__global__ void split_weights_fp8_kernel(float* __restrict__ w, size_t groups, float scale) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    f32x8* p = reinterpret_cast<f32x8*>(w + g * 32);
    h16x8 hi[4];
    int lo8[8], hi8[8];
    auto cl = [](float v) { return __builtin_amdgcn_fmed3f(v, -57344.0f, 57344.0f); };
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x8 x = p[q] * scale;
        hi[q] = __builtin_convertvector(x, h16x8);
        const f32x8 hf = __builtin_convertvector(hi[q], f32x8);
        const f32x8 l = (x - hf) * 2048.0f, hs = hf;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            int a = 0, b = 0;
            a = __builtin_amdgcn_cvt_pk_bf8_f32(cl(l[4 * t]), cl(l[4 * t + 1]), a, false);
            a = __builtin_amdgcn_cvt_pk_bf8_f32(cl(l[4 * t + 2]), cl(l[4 * t + 3]), a, true);
            b = __builtin_amdgcn_cvt_pk_bf8_f32(cl(hs[4 * t]), cl(hs[4 * t + 1]), b, false);
            b = __builtin_amdgcn_cvt_pk_bf8_f32(cl(hs[4 * t + 2]), cl(hs[4 * t + 3]), b, true);
            lo8[2 * q + t] = a;
            hi8[2 * q + t] = b;
        }
    }
    h16x8* o = reinterpret_cast<h16x8*>(w + g * 32);
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] = hi[q];
    int* o8 = reinterpret_cast<int*>(w + g * 32) + 16;
#pragma unroll
    for (int q = 0; q < 8; q++) o8[q] = lo8[q];
#pragma unroll
    for (int q = 0; q < 8; q++) o8[8 + q] = hi8[q];
}

hipError_t launch_split_weights(float* w, size_t n, float scale, int fp8_cross, hipStream_t s) {
    if (n % 32 != 0) return hipErrorInvalidValue;
    const size_t groups = n / 32;
    if (fp8_cross)
        hipLaunchKernelGGL(split_weights_fp8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, w, groups, scale);
    else
        hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, w, groups, scale);
    return hipGetLastError();
}

// ---- weights of the two-source 1x1 GEMM (conv3 ++ downsample, conv_igemm.hip DUAL) ----
__global__ void concat_rows_kernel(const uint4* __restrict__ a, int a16, const uint4* __restrict__ b, int b16,
                                   uint4* __restrict__ out, size_t total16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total16) return;
    const int w = a16 + b16;
    const size_t r = i / w;
    const int c = (int)(i - r * w);
    out[i] = c < a16 ? a[r * a16 + c] : b[r * b16 + (c - a16)];
}

hipError_t launch_concat_rows(const void* a, size_t a_bytes, const void* b, size_t b_bytes, void* out, int rows, hipStream_t s) {
    if (a_bytes % 16 || b_bytes % 16) return hipErrorInvalidValue;
    const size_t total16 = (a_bytes + b_bytes) / 16 * (size_t)rows;
    hipLaunchKernelGGL(concat_rows_kernel, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, s, (const uint4*)a,
                       (int)(a_bytes / 16), (const uint4*)b, (int)(b_bytes / 16), (uint4*)out, total16);
    return hipGetLastError();
}

__global__ void add_f32_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ sum, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sum[i] = x[i] + y[i];
}

hipError_t launch_add_f32(const float* x, const float* y, float* sum, int n, hipStream_t s) {
    hipLaunchKernelGGL(add_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, sum, n);
    return hipGetLastError();
}

}  // namespace infur
