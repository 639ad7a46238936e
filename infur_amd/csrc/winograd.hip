// winograd.hip -- Winograd F(m x m, 3x3) transforms (m = 2, 4 or 6) around the batched implicit-GEMM
// kernel.
//
// A stride-1 3x3 convolution (any dilation d, pad = d) costs 9 multiplies per output in the direct
// form, 4 with F(2x2,3x3), 2.25 with F(4x4,3x3) and 1.78 with F(6x6,3x3): 2.25x / 4x / 5.06x fewer MFMA FLOPs, still plain f32
// arithmetic.  The stride-1 3x3 conv nodes of FCN-ResNet that run inside `session.run`
// (infur/src/predict_onnx.rs:138) and are MFMA-bound in the direct form take this route:
//
//   V[xi][tile][c]  = (B^T d B)[xi]                  input transform   (this file, HBM-bound)
//   M[xi][tile][o]  = sum_c V[xi][tile][c] * U[xi][o][c]    (m+2)^2 batched GEMMs (conv_igemm.hip)
//   Y[tile m x m][o] = A^T M A, + bias, ReLU         output transform  (this file, HBM-bound)
//   U[xi][o][c]     = (G g G^T)[xi]                  weight transform, once at model load
//
// (matrices: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks", F(2x2,3x3) and
// F(4x4,3x3) with interpolation points 0, +-1, +-2, inf.)
//
// Dilation: the output grid splits into d x d interleaved sub-grids (oy = d*y' + ry); on each the
// dilated conv is an ordinary 3x3 / pad-1 conv over the equally sub-sampled input, so a tile is
// (ry, rx, ty, tx) and its (m+2)^2 input patch sits at rows d*(m*ty - 1 + i) + ry.
#include <cstdlib>

#include "hl_format.h"
#include "kernels.h"

namespace infur {

// tile t = ((ry*d + rx) * TY + ty) * TX + tx;  TY/TX = tiles per sub-grid (largest sub-grid)
struct WinoGeom {
    int H, W, d, TY, TX;
};

__device__ __forceinline__ void tile_coords(const WinoGeom& g, int t, int& ry, int& rx, int& ty, int& tx) {
    tx = t % g.TX;
    int r = t / g.TX;
    ty = r % g.TY;
    r /= g.TY;
    rx = r % g.d;
    ry = r / g.d;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef float f32x1 __attribute__((ext_vector_type(1)));

// 1-D transforms on strided arrays of vectors (all loops unrolled; S = element stride)
template <int MT, int S, int SO = S, typename V>
__device__ __forceinline__ void bt_1d(const V* d, V* t) {  // B^T d : (MT+2) -> (MT+2); S / SO = input / output stride
    if constexpr (MT == 2) {
        const V d0 = d[0 * S], d1 = d[1 * S], d2 = d[2 * S], d3 = d[3 * S];
        t[0 * SO] = d0 - d2;
        t[1 * SO] = d1 + d2;
        t[2 * SO] = d2 - d1;
        t[3 * SO] = d1 - d3;
    } else if constexpr (MT == 6) {  // points 0, +-1, +-2, +-1/2, inf
        const V d0 = d[0 * S], d1 = d[1 * S], d2 = d[2 * S], d3 = d[3 * S], d4 = d[4 * S], d5 = d[5 * S], d6 = d[6 * S], d7 = d[7 * S];
        const V a12 = d2 + d6 - 4.25f * d4, b12 = d1 + d5 - 4.25f * d3;
        const V a34 = d6 + 0.25f * d2 - 1.25f * d4, b34 = 0.5f * d1 - 2.5f * d3 + 2.0f * d5;
        const V a56 = d6 + 4.0f * d2 - 5.0f * d4, b56 = 2.0f * d1 - 2.5f * d3 + 0.5f * d5;
        t[0 * SO] = d0 - d6 + 5.25f * (d4 - d2);
        t[1 * SO] = a12 + b12;
        t[2 * SO] = a12 - b12;
        t[3 * SO] = a34 + b34;
        t[4 * SO] = a34 - b34;
        t[5 * SO] = a56 + b56;
        t[6 * SO] = a56 - b56;
        t[7 * SO] = d7 - d1 + 5.25f * (d3 - d5);
    } else {
        const V d0 = d[0 * S], d1 = d[1 * S], d2 = d[2 * S], d3 = d[3 * S], d4 = d[4 * S], d5 = d[5 * S];
        t[0 * SO] = 4.0f * d0 - 5.0f * d2 + d4;
        t[1 * SO] = -4.0f * d1 - 4.0f * d2 + d3 + d4;
        t[2 * SO] = 4.0f * d1 - 4.0f * d2 - d3 + d4;
        t[3 * SO] = -2.0f * d1 - d2 + 2.0f * d3 + d4;
        t[4 * SO] = 2.0f * d1 - d2 - 2.0f * d3 + d4;
        t[5 * SO] = 4.0f * d1 - 5.0f * d3 + d5;
    }
}

template <int MT, int S, int SO = S, typename V>
__device__ __forceinline__ void at_1d(const V* m, V* y) {  // A^T m : (MT+2) -> MT; S / SO = input / output stride
    if constexpr (MT == 2) {
        const V m0 = m[0 * S], m1 = m[1 * S], m2 = m[2 * S], m3 = m[3 * S];
        y[0 * SO] = m0 + m1 + m2;
        y[1 * SO] = m1 - m2 - m3;
    } else if constexpr (MT == 6) {
        const V m0 = m[0 * S], m1 = m[1 * S], m2 = m[2 * S], m3 = m[3 * S], m4 = m[4 * S], m5 = m[5 * S], m6 = m[6 * S], m7 = m[7 * S];
        const V s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4, s56 = m5 + m6, d56 = m5 - m6;
        y[0 * SO] = m0 + s12 + s34 + 32.0f * s56;
        y[1 * SO] = d12 + 2.0f * d34 + 16.0f * d56;
        y[2 * SO] = s12 + 4.0f * s34 + 8.0f * s56;
        y[3 * SO] = d12 + 8.0f * d34 + 4.0f * d56;
        y[4 * SO] = s12 + 16.0f * s34 + 2.0f * s56;
        y[5 * SO] = m7 + d12 + 32.0f * d34 + d56;
    } else {
        const V m0 = m[0 * S], m1 = m[1 * S], m2 = m[2 * S], m3 = m[3 * S], m4 = m[4 * S], m5 = m[5 * S];
        y[0 * SO] = m0 + m1 + m2 + m3 + m4;
        y[1 * SO] = m1 - m2 + 2.0f * m3 - 2.0f * m4;
        y[2 * SO] = m1 + m2 + 4.0f * m3 + 4.0f * m4;
        y[3 * SO] = m1 - m2 + 8.0f * m3 - 8.0f * m4 + m5;
    }
}

template <int MT>
__device__ __forceinline__ void g_1d(float g0, float g1, float g2, float* u) {  // G g : 3 -> (MT+2)
    if constexpr (MT == 2) {
        u[0] = g0;
        u[1] = 0.5f * (g0 + g1 + g2);
        u[2] = 0.5f * (g0 - g1 + g2);
        u[3] = g2;
    } else if constexpr (MT == 6) {
        u[0] = g0;
        u[1] = -(g0 + g1 + g2) * (2.0f / 9.0f);
        u[2] = -(g0 - g1 + g2) * (2.0f / 9.0f);
        u[3] = g0 * (1.0f / 90.0f) + g1 * (1.0f / 45.0f) + g2 * (2.0f / 45.0f);
        u[4] = g0 * (1.0f / 90.0f) - g1 * (1.0f / 45.0f) + g2 * (2.0f / 45.0f);
        u[5] = g0 * (1.0f / 45.0f) + g1 * (1.0f / 90.0f) + g2 * (1.0f / 180.0f);
        u[6] = g0 * (1.0f / 45.0f) - g1 * (1.0f / 90.0f) + g2 * (1.0f / 180.0f);
        u[7] = g2;
    } else {
        u[0] = 0.25f * g0;
        u[1] = -(g0 + g1 + g2) * (1.0f / 6.0f);
        u[2] = -(g0 - g1 + g2) * (1.0f / 6.0f);
        u[3] = g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
        u[4] = g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
        u[5] = g2;
    }
}

// ---- input transform: one thread = one tile x VW channels (VW = 4: 16-byte accesses; VW = 2 for C % 4 != 0) ----
// All (m+2)^2 patch loads are issued first (they are independent: 36 x 16 bytes in flight per lane), the column
// pass runs in place, the row pass stores each transformed row as soon as it is complete.
template <int MT, typename VT>
__global__ void __launch_bounds__(256)
    wino_input_kernel(const float* __restrict__ in, WinoGeom g, int C, int T, float* __restrict__ V, unsigned* __restrict__ amax) {
    constexpr int VW = sizeof(VT) / 4;
    float vmax = 0.f;
    constexpr int AL = MT + 2;
    const int cvn = C / VW;
    const size_t total = (size_t)T * cvn;
    const size_t plane = (size_t)T * C;  // floats per xi plane
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const int t = (int)(i / cvn);
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        VT d[AL * AL];
#pragma unroll
        for (int a = 0; a < AL; a++) {
            const int y = g.d * (MT * ty - 1 + a) + ry;
#pragma unroll
            for (int b = 0; b < AL; b++) {
                const int x = g.d * (MT * tx - 1 + b) + rx;
                VT v = {};
                if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                    v = *reinterpret_cast<const VT*>(in + ((size_t)y * g.W + x) * C + cv * VW);
                d[a * AL + b] = v;
            }
        }
#pragma unroll
        for (int b = 0; b < AL; b++) {  // columns: B^T d, in place
            VT col[AL];
            bt_1d<MT, AL, 1>(d + b, col);
#pragma unroll
            for (int a = 0; a < AL; a++) d[a * AL + b] = col[a];
        }
        float* o = V + (size_t)t * C + cv * VW;
#pragma unroll
        for (int a = 0; a < AL; a++) {  // rows: (.) B, stored as they complete
            VT row[AL];
            bt_1d<MT, 1>(d + a * AL, row);
#pragma unroll
            for (int b = 0; b < AL; b++) {
                *reinterpret_cast<VT*>(o + (size_t)(a * AL + b) * plane) = row[b];
#pragma unroll
                for (int e = 0; e < VW; e++) vmax = fmaxf(vmax, fabsf(row[b][e]));
            }
        }
    }
    if (amax) {  // range monitor of the split mode: max |V|, one atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
    }
}

// ---- output transform: one thread = one tile x VW output channels; + bias, ReLU ----
// The (m+2)^2 plane loads are issued first; the column pass A^T m reduces them to m x (m+2) in place, the row pass
// emits one output row at a time.
template <int MT, typename VT>
__global__ void __launch_bounds__(256)
    wino_output_kernel(const float* __restrict__ M, WinoGeom g, int Cout, int T, const float* __restrict__ bias,
                       int relu, float* __restrict__ out, unsigned* __restrict__ amax) {
    constexpr int VW = sizeof(VT) / 4;
    float vmax = 0.f;
    constexpr int AL = MT + 2;
    const int nvn = Cout / VW;
    const size_t total = (size_t)T * nvn;
    const size_t plane = (size_t)T * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int nv = (int)(i % nvn);
        const int t = (int)(i / nvn);
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        const float* mp = M + (size_t)t * Cout + nv * VW;
        // column by column: the 6 plane vectors of column b are reduced to 4 at once.  (hipcc still hoists all 36 loads
        // to the top -- 207 VGPRs, 2 waves per SIMD; forcing <= 128 registers spills and runs 2x slower, and the kernel
        // moves 4.7-4.9 TB/s either way.)
        VT s[MT * AL];
#pragma unroll
        for (int b = 0; b < AL; b++) {
            VT col[AL];
#pragma unroll
            for (int a = 0; a < AL; a++) col[a] = *reinterpret_cast<const VT*>(mp + (size_t)(a * AL + b) * plane);
            at_1d<MT, 1, AL>(col, s + b);  // A^T m  -> [MT][AL]
        }
        const VT bv = *reinterpret_cast<const VT*>(bias + nv * VW);
#pragma unroll
        for (int a = 0; a < MT; a++) {
            VT yv[MT];
            at_1d<MT, 1>(s + a * AL, yv);  // rows: (.) A -> [MT]
            const int y = g.d * (MT * ty + a) + ry;
            if (y >= g.H) continue;
#pragma unroll
            for (int b = 0; b < MT; b++) {
                const int x = g.d * (MT * tx + b) + rx;
                if (x >= g.W) continue;
                VT v = yv[b] + bv;
#pragma unroll
                for (int e = 0; e < VW; e++) {
                    if (relu) v[e] = fmaxf(v[e], 0.f);
                    vmax = fmaxf(vmax, fabsf(v[e]));
                }
                *reinterpret_cast<VT*>(out + ((size_t)y * g.W + x) * Cout + nv * VW) = v;
            }
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
    }
}

// ---- the same two transforms with 16-BYTE accesses for F(6x6) (round 3) ----
// The per-thread forms above hold a whole (m+2)^2 patch per lane, which for F(6x6) fits the register file only with 8-byte
// vectors (64 vectors) -- and 8-byte global accesses run at 0.54-0.70 of the 16-byte rate on this part
// (MI355X_MICROARCH.md).  Here a workgroup owns one tile x 128 channels and splits the two 1-D passes between its
// threads through LDS: thread (b, cv) loads column b of the patch (8 x 16 bytes, 32 lanes = the 512 contiguous bytes of
// 128 channels), runs the column pass and parks its 8 results in LDS; after one barrier thread (a, cv) reads row a back,
// runs the row pass and stores 8 x 16 bytes.  8 vectors per lane instead of 64 (~50 VGPRs: 5 workgroups per CU), every
// global access 16 bytes and 512-byte contiguous per 32 lanes.  Same 1-D functions on the same values in the same order:
// bit-identical to the per-thread forms (tests/test_gpu_parity.py runs both).
constexpr int WL_CV = 32;  // float4 channel vectors per unit = 128 channels

template <int MT>
__global__ void __launch_bounds__(256)
    wino_input_lds_kernel(const float* __restrict__ in, WinoGeom g, int C, int T, float* __restrict__ V, unsigned* __restrict__ amax) {
    constexpr int AL = MT + 2;
    static_assert(AL <= 8, "one row / column per 32-lane group of a 256-thread workgroup");
    __shared__ f32x4w lds[AL * AL * WL_CV];
    const int tid = threadIdx.x, q = tid >> 5, cvl = tid & 31;  // q: column in pass 1, row in pass 2
    const int cgroups = C / (4 * WL_CV);
    const size_t units = (size_t)T * cgroups;
    const size_t plane = (size_t)T * C;
    float vmax = 0.f;
    for (size_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int cg = (int)(u % cgroups), t = (int)(u / cgroups);
        const int c0 = cg * 4 * WL_CV + cvl * 4;
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        if (q < AL) {
            const int x = g.d * (MT * tx - 1 + q) + rx;
            f32x4w d[AL], col[AL];
#pragma unroll
            for (int a = 0; a < AL; a++) {
                const int y = g.d * (MT * ty - 1 + a) + ry;
                f32x4w v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                    v = *reinterpret_cast<const f32x4w*>(in + ((size_t)y * g.W + x) * C + c0);
                d[a] = v;
            }
            bt_1d<MT, 1, 1>(d, col);  // columns: B^T d
#pragma unroll
            for (int a = 0; a < AL; a++) lds[(a * AL + q) * WL_CV + cvl] = col[a];
        }
        __syncthreads();
        if (q < AL) {
            f32x4w rin[AL], row[AL];
#pragma unroll
            for (int b = 0; b < AL; b++) rin[b] = lds[(q * AL + b) * WL_CV + cvl];
            bt_1d<MT, 1>(rin, row);  // rows: (.) B
            float* o = V + (size_t)t * C + c0;
#pragma unroll
            for (int b = 0; b < AL; b++) {
                *reinterpret_cast<f32x4w*>(o + (size_t)(q * AL + b) * plane) = row[b];
#pragma unroll
                for (int e = 0; e < 4; e++) vmax = fmaxf(vmax, fabsf(row[b][e]));
            }
        }
        __syncthreads();  // the next unit overwrites the LDS image
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
    }
}

template <int MT>
__global__ void __launch_bounds__(256)
    wino_output_lds_kernel(const float* __restrict__ M, WinoGeom g, int Cout, int T, const float* __restrict__ bias, int relu,
                           float* __restrict__ out, unsigned* __restrict__ amax) {
    constexpr int AL = MT + 2;
    static_assert(AL <= 8, "one row / column per 32-lane group of a 256-thread workgroup");
    __shared__ f32x4w lds[MT * AL * WL_CV];
    const int tid = threadIdx.x, q = tid >> 5, cvl = tid & 31;
    const int cgroups = Cout / (4 * WL_CV);
    const size_t units = (size_t)T * cgroups;
    const size_t plane = (size_t)T * Cout;
    float vmax = 0.f;
    for (size_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int cg = (int)(u % cgroups), t = (int)(u / cgroups);
        const int n0 = cg * 4 * WL_CV + cvl * 4;
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        if (q < AL) {
            const float* mp = M + (size_t)t * Cout + n0;
            f32x4w col[AL], s[MT];
#pragma unroll
            for (int a = 0; a < AL; a++) col[a] = *reinterpret_cast<const f32x4w*>(mp + (size_t)(a * AL + q) * plane);
            at_1d<MT, 1, 1>(col, s);  // A^T m over the rows of column q
#pragma unroll
            for (int a = 0; a < MT; a++) lds[(a * AL + q) * WL_CV + cvl] = s[a];
        }
        __syncthreads();
        if (q < MT) {
            f32x4w rin[AL], yv[MT];
#pragma unroll
            for (int b = 0; b < AL; b++) rin[b] = lds[(q * AL + b) * WL_CV + cvl];
            at_1d<MT, 1>(rin, yv);  // rows: (.) A
            const f32x4w bv = *reinterpret_cast<const f32x4w*>(bias + n0);
            const int y = g.d * (MT * ty + q) + ry;
            if (y < g.H) {
#pragma unroll
                for (int b = 0; b < MT; b++) {
                    const int x = g.d * (MT * tx + b) + rx;
                    if (x >= g.W) continue;
                    f32x4w v = yv[b] + bv;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                        vmax = fmaxf(vmax, fabsf(v[e]));
                    }
                    *reinterpret_cast<f32x4w*>(out + ((size_t)y * g.W + x) * Cout + n0) = v;
                }
            }
        }
        __syncthreads();
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
    }
}

// ---- the two transforms on THREE-BYTE tensors (INFUR_DTYPE_F16_HL, hl_format.h) ----
// Same two-pass LDS scheme as above with 8 channels per lane: a lane's accesses are 16 bytes of the f16 hi plane + 8 bytes of the
// e5m2 lo plane (16 lanes = 256 + 128 contiguous bytes of 128 channels).  A workgroup = one tile x 128 channels, 8 groups of 16
// lanes.  The input transform reads x = hi + lo / kHlLoScale, transforms in f32 and writes V * v_scale as hi / lo planes
// [(mt+2)^2][T][C] for the batched HL GEMM (conv_hl.hip); the output transform reads that GEMM's f32 result and writes the conv's
// output as hi / lo planes.  Same 1-D functions in the same order as the f32 forms.
typedef float f32x8w __attribute__((ext_vector_type(8)));
constexpr int WH_L = 16;  // lanes per group = 128 channels

__device__ __forceinline__ f32x8w hl_load8(const _Float16* hi, const unsigned char* lo, size_t e) {
    float x[8];
    hl_join8(*reinterpret_cast<const hl_f16x8*>(hi + e), *reinterpret_cast<const hl_u32x2*>(lo + e), x);
    f32x8w v;
#pragma unroll
    for (int t = 0; t < 8; t++) v[t] = x[t];
    return v;
}
__device__ __forceinline__ void hl_store8(_Float16* hi, unsigned char* lo, size_t e, const f32x8w v, const float lb = -kHlHiMax) {
    float x[8];
#pragma unroll
    for (int t = 0; t < 8; t++) x[t] = v[t];
    hl_f16x8 hv;
    hl_u32x2 lv;
    hl_split8(x, hv, lv, lb);
    *reinterpret_cast<hl_f16x8*>(hi + e) = hv;
    *reinterpret_cast<hl_u32x2*>(lo + e) = lv;
}

template <int MT>
__global__ void __launch_bounds__(128)
    wino_input_hl_kernel(const _Float16* __restrict__ in_hi, const unsigned char* __restrict__ in_lo, WinoGeom g, int C, int T, float v_scale,
                         _Float16* __restrict__ V_hi, unsigned char* __restrict__ V_lo) {
    constexpr int AL = MT + 2;
    static_assert(AL <= 8, "one row / column per 16-lane group of a 128-thread workgroup");
    __shared__ f32x8w lds[AL * AL * WH_L];
    hl_set_fp16_ovfl();
    const int tid = threadIdx.x, q = tid >> 4, cvl = tid & 15;
    const int cgroups = C / (8 * WH_L);
    const size_t units = (size_t)T * cgroups;
    const size_t plane = (size_t)T * C;
    for (size_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int cg = (int)(u % cgroups), t = (int)(u / cgroups);
        const int c0 = cg * 8 * WH_L + cvl * 8;
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        if (q < AL) {
            const int x = g.d * (MT * tx - 1 + q) + rx;
            f32x8w d[AL], col[AL];
            // the loads are UNCONDITIONAL (an out-of-frame tap reads pixel (0, 0) and is zeroed afterwards): behind a branch hipcc
            // waits for each load before it issues the next -- 16 serial round trips per unit, 2.5 TB/s instead of 4+
            hl_f16x8 rh[AL];
            hl_u32x2 rl[AL];
            bool ok[AL];
#pragma unroll
            for (int a = 0; a < AL; a++) {
                const int y = g.d * (MT * ty - 1 + a) + ry;
                ok[a] = (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
                const size_t e = ok[a] ? ((size_t)y * g.W + x) * C + c0 : (size_t)c0;
                rh[a] = *reinterpret_cast<const hl_f16x8*>(in_hi + e);
                rl[a] = *reinterpret_cast<const hl_u32x2*>(in_lo + e);
            }
#pragma unroll
            for (int a = 0; a < AL; a++) {
                float xv[8];
                hl_join8(rh[a], rl[a], xv);
#pragma unroll
                for (int t = 0; t < 8; t++) d[a][t] = ok[a] ? xv[t] : 0.f;
            }
            bt_1d<MT, 1, 1>(d, col);
#pragma unroll
            for (int a = 0; a < AL; a++) lds[(a * AL + q) * WH_L + cvl] = col[a];
        }
        __syncthreads();
        if (q < AL) {
            f32x8w rin[AL], row[AL];
#pragma unroll
            for (int b = 0; b < AL; b++) rin[b] = lds[(q * AL + b) * WH_L + cvl];
            bt_1d<MT, 1>(rin, row);
            const size_t o = (size_t)t * C + c0;
#pragma unroll
            for (int b = 0; b < AL; b++) hl_store8(V_hi, V_lo, o + (size_t)(q * AL + b) * plane, row[b] * v_scale);
        }
        __syncthreads();
    }
}

template <int MT>
__global__ void __launch_bounds__(128)
    wino_output_hl_kernel(const float* __restrict__ M, WinoGeom g, int Cout, int T, const float* __restrict__ bias, int relu,
                          _Float16* __restrict__ out_hi, unsigned char* __restrict__ out_lo) {
    constexpr int AL = MT + 2;
    static_assert(AL <= 8, "one row / column per 16-lane group of a 128-thread workgroup");
    __shared__ f32x8w lds[MT * AL * WH_L];
    hl_set_fp16_ovfl();
    const int tid = threadIdx.x, q = tid >> 4, cvl = tid & 15;
    const int cgroups = Cout / (8 * WH_L);
    const size_t units = (size_t)T * cgroups;
    const size_t plane = (size_t)T * Cout;
    for (size_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int cg = (int)(u % cgroups), t = (int)(u / cgroups);
        const int n0 = cg * 8 * WH_L + cvl * 8;
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        if (q < AL) {
            const float* mp = M + (size_t)t * Cout + n0;
            f32x8w col[AL], s[MT];
#pragma unroll
            for (int a = 0; a < AL; a++) {
                const f32x4w v0 = *reinterpret_cast<const f32x4w*>(mp + (size_t)(a * AL + q) * plane);
                const f32x4w v1 = *reinterpret_cast<const f32x4w*>(mp + (size_t)(a * AL + q) * plane + 4);
                col[a] = f32x8w{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            }
            at_1d<MT, 1, 1>(col, s);
#pragma unroll
            for (int a = 0; a < MT; a++) lds[(a * AL + q) * WH_L + cvl] = s[a];
        }
        __syncthreads();
        if (q < MT) {
            f32x8w rin[AL], yv[MT];
#pragma unroll
            for (int b = 0; b < AL; b++) rin[b] = lds[(q * AL + b) * WH_L + cvl];
            at_1d<MT, 1>(rin, yv);
            const f32x4w b0 = *reinterpret_cast<const f32x4w*>(bias + n0), b1 = *reinterpret_cast<const f32x4w*>(bias + n0 + 4);
            const f32x8w bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            const int y = g.d * (MT * ty + q) + ry;
            if (y < g.H) {
#pragma unroll
                for (int b = 0; b < MT; b++) {
                    const int x = g.d * (MT * tx + b) + rx;
                    if (x >= g.W) continue;
                    const f32x8w v = yv[b] + bv;
                    hl_store8(out_hi, out_lo, ((size_t)y * g.W + x) * Cout + n0, v, relu ? 0.f : -kHlHiMax);  // (the ReLU is the split's lower clamp)
                }
            }
        }
        __syncthreads();
    }
}

// ---- weight transform: U[xi][o][c] = (G g G^T)[xi],  g = w[o][c][3][3] (OIHW) ----
template <int MT>
__global__ void wino_weight_kernel(const float* __restrict__ w, int O, int I, float* __restrict__ U) {
    constexpr int AL = MT + 2;
    const size_t total = (size_t)O * I;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float* gk = w + i * 9;  // [o][c][ky][kx], i = o*I + c
        float t[AL][3], u[AL];
#pragma unroll
        for (int b = 0; b < 3; b++) {  // G g (over ky)
            g_1d<MT>(gk[0 * 3 + b], gk[1 * 3 + b], gk[2 * 3 + b], u);
#pragma unroll
            for (int a = 0; a < AL; a++) t[a][b] = u[a];
        }
#pragma unroll
        for (int a = 0; a < AL; a++) {  // (.) G^T (over kx)
            g_1d<MT>(t[a][0], t[a][1], t[a][2], u);
#pragma unroll
            for (int b = 0; b < AL; b++) U[(size_t)(AL * a + b) * total + i] = u[b];
        }
    }
}

static WinoGeom geom(int H, int W, int d, int mt) {
    WinoGeom g;
    g.H = H;
    g.W = W;
    g.d = d;
    g.TY = ((H + d - 1) / d + mt - 1) / mt;
    g.TX = ((W + d - 1) / d + mt - 1) / mt;
    return g;
}

int wino_num_tiles(int H, int W, int d, int mt) {
    const WinoGeom g = geom(H, W, d, mt);
    return d * d * g.TY * g.TX;
}

static unsigned grid_for(size_t work) {
    size_t blocks = (work + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    return (unsigned)blocks;
}

// test/measurement hook: INFUR_WINO_VEC=2 forces the 8-byte form, INFUR_WINO_VEC=1 the 4-byte form of F(6x6).
// The 4-byte form needs 122 / 118 VGPRs instead of 162 / 210, so a transform wave fits on a SIMD next to the waves of
// an f32 GEMM workgroup of the other frame in flight -- measured (scripts/ab_wino_vec.sh, same box, back to back): the
// transforms alone run at the same speed (0.49 + 0.35 ms vs 0.51 + 0.36 ms per 1080p frame) and the two-frames-in-flight
// rate does not move (f32 102.6 vs 102.9, f32s 205.0 vs 204.8 frames/s): co-residency is not what limits the overlap.
static int wino_forced() {
    static const int forced = getenv("INFUR_WINO_VEC") ? atoi(getenv("INFUR_WINO_VEC")) : 0;
    return forced;
}
static bool wino_vec4(int C) { return (C & 3) == 0 && wino_forced() != 2; }
// the LDS two-pass forms (16-byte accesses for F(6x6)): INFUR_WINO_LDS=0 switches back to the per-thread forms (A/B, tests)
static bool wino_lds(int C, int mt) {
    static const int on = getenv("INFUR_WINO_LDS") ? atoi(getenv("INFUR_WINO_LDS")) : 1;
    return on && mt == 6 && C % (4 * WL_CV) == 0 && wino_forced() == 0;
}
static unsigned grid_units(size_t units) { return (unsigned)(units < 256 * 40 ? units : 256 * 40); }

hipError_t launch_wino_input(const float* in, int H, int W, int C, int d, int mt, float* V, unsigned* amax, hipStream_t s) {
    const WinoGeom g = geom(H, W, d, mt);
    const int T = d * d * g.TY * g.TX;
    const bool v4 = wino_vec4(C) && mt != 6;  // F(6x6): 64 patch vectors per thread -- 8-byte vectors keep them in registers
    const unsigned blocks = grid_for((size_t)T * (C / (v4 ? 4 : 2)));
    if (wino_lds(C, mt))
        hipLaunchKernelGGL(wino_input_lds_kernel<6>, dim3(grid_units((size_t)T * (C / (4 * WL_CV)))), dim3(256), 0, s, in, g, C, T, V, amax);
    else if (mt == 6 && wino_forced() == 1)
        hipLaunchKernelGGL((wino_input_kernel<6, f32x1>), dim3(grid_for((size_t)T * C)), dim3(256), 0, s, in, g, C, T, V, amax);
    else if (mt == 6)
        hipLaunchKernelGGL((wino_input_kernel<6, f32x2>), dim3(blocks), dim3(256), 0, s, in, g, C, T, V, amax);
    else if (mt == 2 && v4)
        hipLaunchKernelGGL((wino_input_kernel<2, f32x4w>), dim3(blocks), dim3(256), 0, s, in, g, C, T, V, amax);
    else if (mt == 2)
        hipLaunchKernelGGL((wino_input_kernel<2, f32x2>), dim3(blocks), dim3(256), 0, s, in, g, C, T, V, amax);
    else if (v4)
        hipLaunchKernelGGL((wino_input_kernel<4, f32x4w>), dim3(blocks), dim3(256), 0, s, in, g, C, T, V, amax);
    else
        hipLaunchKernelGGL((wino_input_kernel<4, f32x2>), dim3(blocks), dim3(256), 0, s, in, g, C, T, V, amax);
    return hipGetLastError();
}

hipError_t launch_wino_output(const float* M, int H, int W, int Cout, int d, int mt, const float* bias, int relu,
                              float* out, unsigned* amax, hipStream_t s) {
    const WinoGeom g = geom(H, W, d, mt);
    const int T = d * d * g.TY * g.TX;
    const bool v4 = wino_vec4(Cout) && mt != 6;
    const unsigned blocks = grid_for((size_t)T * (Cout / (v4 ? 4 : 2)));
    if (wino_lds(Cout, mt))
        hipLaunchKernelGGL(wino_output_lds_kernel<6>, dim3(grid_units((size_t)T * (Cout / (4 * WL_CV)))), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    else if (mt == 6 && wino_forced() == 1)
        hipLaunchKernelGGL((wino_output_kernel<6, f32x1>), dim3(grid_for((size_t)T * Cout)), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    else if (mt == 6)
        hipLaunchKernelGGL((wino_output_kernel<6, f32x2>), dim3(blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    else if (mt == 2 && v4)
        hipLaunchKernelGGL((wino_output_kernel<2, f32x4w>), dim3(blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    else if (mt == 2)
        hipLaunchKernelGGL((wino_output_kernel<2, f32x2>), dim3(blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    else if (v4)
        hipLaunchKernelGGL((wino_output_kernel<4, f32x4w>), dim3(blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    else
        hipLaunchKernelGGL((wino_output_kernel<4, f32x2>), dim3(blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    return hipGetLastError();
}

hipError_t launch_wino_input_hl(const void* in_hi, const void* in_lo, int H, int W, int C, int d, int mt, float v_scale, void* V_hi, void* V_lo,
                                hipStream_t s) {
    if (C % (8 * WH_L) != 0) return hipErrorInvalidValue;
    const WinoGeom g = geom(H, W, d, mt);
    const int T = d * d * g.TY * g.TX;
    const dim3 grid(grid_units((size_t)T * (C / (8 * WH_L))));
    const _Float16* ih = static_cast<const _Float16*>(in_hi);
    const unsigned char* il = static_cast<const unsigned char*>(in_lo);
    _Float16* vh = static_cast<_Float16*>(V_hi);
    unsigned char* vl = static_cast<unsigned char*>(V_lo);
    if (mt == 6)
        hipLaunchKernelGGL(wino_input_hl_kernel<6>, grid, dim3(128), 0, s, ih, il, g, C, T, v_scale, vh, vl);
    else if (mt == 4)
        hipLaunchKernelGGL(wino_input_hl_kernel<4>, grid, dim3(128), 0, s, ih, il, g, C, T, v_scale, vh, vl);
    else
        hipLaunchKernelGGL(wino_input_hl_kernel<2>, grid, dim3(128), 0, s, ih, il, g, C, T, v_scale, vh, vl);
    return hipGetLastError();
}

hipError_t launch_wino_output_hl(const float* M, int H, int W, int Cout, int d, int mt, const float* bias, int relu, void* out_hi, void* out_lo,
                                 hipStream_t s) {
    if (Cout % (8 * WH_L) != 0) return hipErrorInvalidValue;
    const WinoGeom g = geom(H, W, d, mt);
    const int T = d * d * g.TY * g.TX;
    const dim3 grid(grid_units((size_t)T * (Cout / (8 * WH_L))));
    _Float16* oh = static_cast<_Float16*>(out_hi);
    unsigned char* ol = static_cast<unsigned char*>(out_lo);
    if (mt == 6)
        hipLaunchKernelGGL(wino_output_hl_kernel<6>, grid, dim3(128), 0, s, M, g, Cout, T, bias, relu, oh, ol);
    else if (mt == 4)
        hipLaunchKernelGGL(wino_output_hl_kernel<4>, grid, dim3(128), 0, s, M, g, Cout, T, bias, relu, oh, ol);
    else
        hipLaunchKernelGGL(wino_output_hl_kernel<2>, grid, dim3(128), 0, s, M, g, Cout, T, bias, relu, oh, ol);
    return hipGetLastError();
}

hipError_t launch_wino_weights(const float* w_oihw, int O, int I, int mt, float* U, hipStream_t s) {
    size_t blocks = ((size_t)O * I + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (mt == 2)
        hipLaunchKernelGGL(wino_weight_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, O, I, U);
    else if (mt == 6)
        hipLaunchKernelGGL(wino_weight_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, O, I, U);
    else
        hipLaunchKernelGGL(wino_weight_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, O, I, U);
    return hipGetLastError();
}

}  // namespace infur
