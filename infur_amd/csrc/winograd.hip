// winograd.hip -- Winograd F(m x m, 3x3) transforms (m = 2 or 4) around the batched implicit-GEMM
// kernel.
//
// A stride-1 3x3 convolution (any dilation d, pad = d) costs 9 multiplies per output in the direct
// form, 4 with F(2x2,3x3) and 2.25 with F(4x4,3x3): 2.25x / 4x fewer MFMA FLOPs, still plain f32
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

// 1-D transforms on strided arrays of vectors (all loops unrolled; S = element stride)
template <int MT, int S, typename V>
__device__ __forceinline__ void bt_1d(const V* d, V* t) {  // B^T d : (MT+2) -> (MT+2)
    if constexpr (MT == 2) {
        t[0 * S] = d[0 * S] - d[2 * S];
        t[1 * S] = d[1 * S] + d[2 * S];
        t[2 * S] = d[2 * S] - d[1 * S];
        t[3 * S] = d[1 * S] - d[3 * S];
    } else {
        const V d0 = d[0 * S], d1 = d[1 * S], d2 = d[2 * S], d3 = d[3 * S], d4 = d[4 * S], d5 = d[5 * S];
        t[0 * S] = 4.0f * d0 - 5.0f * d2 + d4;
        t[1 * S] = -4.0f * d1 - 4.0f * d2 + d3 + d4;
        t[2 * S] = 4.0f * d1 - 4.0f * d2 - d3 + d4;
        t[3 * S] = -2.0f * d1 - d2 + 2.0f * d3 + d4;
        t[4 * S] = 2.0f * d1 - d2 - 2.0f * d3 + d4;
        t[5 * S] = 4.0f * d1 - 5.0f * d3 + d5;
    }
}

template <int MT, int S, typename V>
__device__ __forceinline__ void at_1d(const V* m, V* y) {  // A^T m : (MT+2) -> MT
    if constexpr (MT == 2) {
        y[0 * S] = m[0 * S] + m[1 * S] + m[2 * S];
        y[1 * S] = m[1 * S] - m[2 * S] - m[3 * S];
    } else {
        const V m0 = m[0 * S], m1 = m[1 * S], m2 = m[2 * S], m3 = m[3 * S], m4 = m[4 * S], m5 = m[5 * S];
        y[0 * S] = m0 + m1 + m2 + m3 + m4;
        y[1 * S] = m1 - m2 + 2.0f * m3 - 2.0f * m4;
        y[2 * S] = m1 + m2 + 4.0f * m3 + 4.0f * m4;
        y[3 * S] = m1 - m2 + 8.0f * m3 - 8.0f * m4 + m5;
    }
}

template <int MT>
__device__ __forceinline__ void g_1d(float g0, float g1, float g2, float* u) {  // G g : 3 -> (MT+2)
    if constexpr (MT == 2) {
        u[0] = g0;
        u[1] = 0.5f * (g0 + g1 + g2);
        u[2] = 0.5f * (g0 - g1 + g2);
        u[3] = g2;
    } else {
        u[0] = 0.25f * g0;
        u[1] = -(g0 + g1 + g2) * (1.0f / 6.0f);
        u[2] = -(g0 - g1 + g2) * (1.0f / 6.0f);
        u[3] = g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
        u[4] = g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
        u[5] = g2;
    }
}

// ---- input transform: one thread = one tile x 2 channels ----
template <int MT>
__global__ void __launch_bounds__(256)
    wino_input_kernel(const float* __restrict__ in, WinoGeom g, int C, int T, float* __restrict__ V, unsigned* __restrict__ amax) {
    float vmax = 0.f;
    constexpr int AL = MT + 2;
    const int cvn = C >> 1;
    const size_t total = (size_t)T * cvn;
    const size_t plane = (size_t)T * C;  // floats per xi plane
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const int t = (int)(i / cvn);
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        f32x2 d[AL * AL], r[AL * AL];
#pragma unroll
        for (int a = 0; a < AL; a++) {
            const int y = g.d * (MT * ty - 1 + a) + ry;
#pragma unroll
            for (int b = 0; b < AL; b++) {
                const int x = g.d * (MT * tx - 1 + b) + rx;
                f32x2 v = {0.f, 0.f};
                if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                    v = *reinterpret_cast<const f32x2*>(in + ((size_t)y * g.W + x) * C + cv * 2);
                d[a * AL + b] = v;
            }
        }
#pragma unroll
        for (int b = 0; b < AL; b++) bt_1d<MT, AL>(d + b, r + b);      // columns: B^T d
#pragma unroll
        for (int a = 0; a < AL; a++) bt_1d<MT, 1>(r + a * AL, d + a * AL);  // rows: (.) B
        float* o = V + (size_t)t * C + cv * 2;
#pragma unroll
        for (int xi = 0; xi < AL * AL; xi++) {
            *reinterpret_cast<f32x2*>(o + (size_t)xi * plane) = d[xi];
            vmax = fmaxf(vmax, fmaxf(fabsf(d[xi].x), fabsf(d[xi].y)));
        }
    }
    if (amax) {  // range monitor of the split mode: max |V|, one atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
    }
}

// ---- output transform: one thread = one tile x 2 output channels; + bias, ReLU ----
template <int MT>
__global__ void __launch_bounds__(256)
    wino_output_kernel(const float* __restrict__ M, WinoGeom g, int Cout, int T, const float* __restrict__ bias,
                       int relu, float* __restrict__ out, unsigned* __restrict__ amax) {
    float vmax = 0.f;
    constexpr int AL = MT + 2;
    const int nvn = Cout >> 1;
    const size_t total = (size_t)T * nvn;
    const size_t plane = (size_t)T * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int nv = (int)(i % nvn);
        const int t = (int)(i / nvn);
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        const float* mp = M + (size_t)t * Cout + nv * 2;
        f32x2 m[AL * AL], s[MT * AL], yv[MT * MT];
#pragma unroll
        for (int xi = 0; xi < AL * AL; xi++) m[xi] = *reinterpret_cast<const f32x2*>(mp + (size_t)xi * plane);
#pragma unroll
        for (int b = 0; b < AL; b++) at_1d<MT, AL>(m + b, s + b);        // columns: A^T m  -> [MT][AL]
#pragma unroll
        for (int a = 0; a < MT; a++) at_1d<MT, 1>(s + a * AL, yv + a * MT);  // rows: (.) A -> [MT][MT]
        const f32x2 bv = *reinterpret_cast<const f32x2*>(bias + nv * 2);
#pragma unroll
        for (int a = 0; a < MT; a++) {
            const int y = g.d * (MT * ty + a) + ry;
            if (y >= g.H) continue;
#pragma unroll
            for (int b = 0; b < MT; b++) {
                const int x = g.d * (MT * tx + b) + rx;
                if (x >= g.W) continue;
                f32x2 v = yv[a * MT + b] + bv;
                if (relu) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                }
                *reinterpret_cast<f32x2*>(out + ((size_t)y * g.W + x) * Cout + nv * 2) = v;
                vmax = fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
            }
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(vmax));
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

hipError_t launch_wino_input(const float* in, int H, int W, int C, int d, int mt, float* V, unsigned* amax, hipStream_t s) {
    const WinoGeom g = geom(H, W, d, mt);
    const int T = d * d * g.TY * g.TX;
    const unsigned blocks = grid_for((size_t)T * (C / 2));
    if (mt == 2)
        hipLaunchKernelGGL(wino_input_kernel<2>, dim3(blocks), dim3(256), 0, s, in, g, C, T, V, amax);
    else
        hipLaunchKernelGGL(wino_input_kernel<4>, dim3(blocks), dim3(256), 0, s, in, g, C, T, V, amax);
    return hipGetLastError();
}

hipError_t launch_wino_output(const float* M, int H, int W, int Cout, int d, int mt, const float* bias, int relu,
                              float* out, unsigned* amax, hipStream_t s) {
    const WinoGeom g = geom(H, W, d, mt);
    const int T = d * d * g.TY * g.TX;
    const unsigned blocks = grid_for((size_t)T * (Cout / 2));
    if (mt == 2)
        hipLaunchKernelGGL(wino_output_kernel<2>, dim3(blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    else
        hipLaunchKernelGGL(wino_output_kernel<4>, dim3(blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out, amax);
    return hipGetLastError();
}

hipError_t launch_wino_weights(const float* w_oihw, int O, int I, int mt, float* U, hipStream_t s) {
    size_t blocks = ((size_t)O * I + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (mt == 2)
        hipLaunchKernelGGL(wino_weight_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, O, I, U);
    else
        hipLaunchKernelGGL(wino_weight_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, O, I, U);
    return hipGetLastError();
}

}  // namespace infur
