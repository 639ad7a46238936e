// winograd.hip -- Winograd F(2x2, 3x3) transforms around the batched implicit-GEMM kernel.
//
// A stride-1 3x3 convolution (any dilation d, pad = d) costs 9 multiplies per output in the direct
// form and 4 (16 per 2x2 output tile) in the Winograd domain: 2.25x fewer MFMA FLOPs, still plain
// f32 arithmetic.  The conv nodes of FCN-ResNet that run inside `session.run`
// (infur/src/predict_onnx.rs:138) with >= 512 input channels are MFMA-bound and take this route:
//
//   V[xi][tile][c]  = (B^T d B)[xi]            input transform   (this file, HBM-bound)
//   M[xi][tile][o]  = sum_c V[xi][tile][c] * U[xi][o][c]         16 batched GEMMs (conv_igemm.hip)
//   Y[tile 2x2][o]  = A^T M A, + bias, ReLU     output transform  (this file, HBM-bound)
//   U[xi][o][c]     = (G g G^T)[xi]             weight transform, once at model load
//
// Dilation: the output grid splits into d x d interleaved sub-grids (oy = d*y' + ry); on each the
// dilated conv is an ordinary 3x3 / pad-1 conv over the equally sub-sampled input, so a tile is
// (ry, rx, ty, tx) and its 4x4 input patch sits at rows d*(2*ty - 1 + i) + ry.
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

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// ---- input transform: one thread = one tile x 4 channels ----
__global__ void __launch_bounds__(256)
    wino_input_kernel(const float* __restrict__ in, WinoGeom g, int C, int T, float* __restrict__ V) {
    const int c4n = C >> 2;
    const size_t total = (size_t)T * c4n;
    const size_t plane = (size_t)T * C;  // floats per xi plane
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        const int t = (int)(i / c4n);
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int y = g.d * (2 * ty - 1 + a) + ry;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = g.d * (2 * tx - 1 + b) + rx;
                d[a][b] = ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                              ? *reinterpret_cast<const float4*>(in + ((size_t)y * g.W + x) * C + c4 * 4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // B^T d: rows (d0-d2, d1+d2, d2-d1, d1-d3)
        float4 r[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            r[0][b] = f4sub(d[0][b], d[2][b]);
            r[1][b] = f4add(d[1][b], d[2][b]);
            r[2][b] = f4sub(d[2][b], d[1][b]);
            r[3][b] = f4sub(d[1][b], d[3][b]);
        }
        float* o = V + (size_t)t * C + c4 * 4;
#pragma unroll
        for (int a = 0; a < 4; a++) {  // (.) B: columns (c0-c2, c1+c2, c2-c1, c1-c3)
            *reinterpret_cast<float4*>(o + (size_t)(4 * a + 0) * plane) = f4sub(r[a][0], r[a][2]);
            *reinterpret_cast<float4*>(o + (size_t)(4 * a + 1) * plane) = f4add(r[a][1], r[a][2]);
            *reinterpret_cast<float4*>(o + (size_t)(4 * a + 2) * plane) = f4sub(r[a][2], r[a][1]);
            *reinterpret_cast<float4*>(o + (size_t)(4 * a + 3) * plane) = f4sub(r[a][1], r[a][3]);
        }
    }
}

// ---- output transform: one thread = one tile x 4 output channels; + bias, ReLU ----
__global__ void __launch_bounds__(256)
    wino_output_kernel(const float* __restrict__ M, WinoGeom g, int Cout, int T, const float* __restrict__ bias,
                       int relu, float* __restrict__ out) {
    const int n4n = Cout >> 2;
    const size_t total = (size_t)T * n4n;
    const size_t plane = (size_t)T * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n4 = (int)(i % n4n);
        const int t = (int)(i / n4n);
        int ry, rx, ty, tx;
        tile_coords(g, t, ry, rx, ty, tx);
        const float* mp = M + (size_t)t * Cout + n4 * 4;
        float4 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) m[a][b] = *reinterpret_cast<const float4*>(mp + (size_t)(4 * a + b) * plane);
        // A^T m: rows (m0+m1+m2, m1-m2-m3)
        float4 s[2][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            s[0][b] = f4add(f4add(m[0][b], m[1][b]), m[2][b]);
            s[1][b] = f4sub(f4sub(m[1][b], m[2][b]), m[3][b]);
        }
        const float4 bv = *reinterpret_cast<const float4*>(bias + n4 * 4);
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int y = g.d * (2 * ty + a) + ry;
            if (y >= g.H) continue;
            float4 yv[2];
            yv[0] = f4add(f4add(s[a][0], s[a][1]), s[a][2]);
            yv[1] = f4sub(f4sub(s[a][1], s[a][2]), s[a][3]);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int x = g.d * (2 * tx + b) + rx;
                if (x >= g.W) continue;
                float4 v = f4add(yv[b], bv);
                if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *reinterpret_cast<float4*>(out + ((size_t)y * g.W + x) * Cout + n4 * 4) = v;
            }
        }
    }
}

// ---- weight transform: U[xi][o][c] = (G g G^T)[xi],  g = w[o][c][3][3] (OIHW) ----
__global__ void wino_weight_kernel(const float* __restrict__ w, int O, int I, float* __restrict__ U) {
    const size_t total = (size_t)O * I;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float* gk = w + i * 9;  // [o][c][ky][kx], i = o*I + c
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; b++) {  // G g : rows (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
            const float g0 = gk[0 * 3 + b], g1 = gk[1 * 3 + b], g2 = gk[2 * 3 + b];
            t[0][b] = g0;
            t[1][b] = 0.5f * (g0 + g1 + g2);
            t[2][b] = 0.5f * (g0 - g1 + g2);
            t[3][b] = g2;
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {  // (.) G^T
            const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                        u3 = t[a][2];
            U[(size_t)(4 * a + 0) * total + i] = u0;
            U[(size_t)(4 * a + 1) * total + i] = u1;
            U[(size_t)(4 * a + 2) * total + i] = u2;
            U[(size_t)(4 * a + 3) * total + i] = u3;
        }
    }
}

static WinoGeom geom(int H, int W, int d) {
    WinoGeom g;
    g.H = H;
    g.W = W;
    g.d = d;
    g.TY = ((H + d - 1) / d + 1) / 2;
    g.TX = ((W + d - 1) / d + 1) / 2;
    return g;
}

int wino_num_tiles(int H, int W, int d) {
    const WinoGeom g = geom(H, W, d);
    return d * d * g.TY * g.TX;
}

hipError_t launch_wino_input(const float* in, int H, int W, int C, int d, float* V, hipStream_t s) {
    const WinoGeom g = geom(H, W, d);
    const int T = d * d * g.TY * g.TX;
    size_t blocks = ((size_t)T * (C / 4) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, g, C, T, V);
    return hipGetLastError();
}

hipError_t launch_wino_output(const float* M, int H, int W, int Cout, int d, const float* bias, int relu, float* out,
                              hipStream_t s) {
    const WinoGeom g = geom(H, W, d);
    const int T = d * d * g.TY * g.TX;
    size_t blocks = ((size_t)T * (Cout / 4) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)blocks), dim3(256), 0, s, M, g, Cout, T, bias, relu, out);
    return hipGetLastError();
}

hipError_t launch_wino_weights(const float* w_oihw, int O, int I, float* U, hipStream_t s) {
    size_t blocks = ((size_t)O * I + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, O, I, U);
    return hipGetLastError();
}

}  // namespace infur
