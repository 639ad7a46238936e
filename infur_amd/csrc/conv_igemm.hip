// conv_igemm.hip -- convolution as an im2col-free implicit GEMM on the gfx950 f32 matrix
// core (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain).
//
// Replaces the Conv nodes ONNX Runtime executes inside `session.run`
// (infur/src/predict_onnx.rs:138) for every 1x1 and 3x3 convolution of FCN-ResNet
// (stride 1/2, dilation 1/2/4), with bias, residual add and ReLU fused into the epilogue.
//
//   GEMM view:  M = OH*OW output pixels, N = Cout, K = KH*KW*Cin  (tap-major, Cin inner)
//   A[m][k]  = in[(oy*s - p + ky*d), (ox*s - p + kx*d), c]   NHWC, gathered, zero padded
//   B[n][k]  = wt[n][ky][kx][c]                               OHWI, k contiguous
//
// Tiling: BM x BN x 32 per workgroup, one wave per SIMD, each wave TM x TN tiles of 32x32.
// Operands are staged global -> VGPR -> LDS (row stride 36 floats: ds_write_b128 and
// ds_read_b128 both conflict-free) with two LDS buffers and one barrier per K step; the
// global loads of step k+1 are issued before the MFMAs of step k.
// A lane reads 4 consecutive k of its row with one ds_read_b128 (lanes 0-31: k 0-3,
// lanes 32-63: k 4-7 of an 8-wide slice) and feeds them to 4 MFMAs; A and B use the same
// permutation of k, so the sum is complete.
#include "kernels.h"

namespace infur {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;  // floats

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64, 2)
    conv_igemm_f32_kernel(const ConvArgs a, const int mtiles, const int ntiles) {
    constexpr int T = WM * WN * 64;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_IT = BM * 8 / T;  // float4 per thread per K step
    constexpr int B_IT = BN * 8 / T;
    static_assert(BM * 8 % T == 0 && BN * 8 % T == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                        // [2][BM][LDS_STRIDE]
    float* Bs = smem + 2 * BM * LDS_STRIDE;  // [2][BN][LDS_STRIDE]

    // XCD-aware tile order: block b runs on XCD b % 8; give every XCD a contiguous run of
    // tiles (n fastest) so the N-tiles that share an activation tile share one L2.
    const int nblk = mtiles * ntiles;
    int tile;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, loc = b >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int M = a.OH * a.OW;
    const int Ktot = a.KH * a.KW * a.Cin;

    // per-thread gather coordinates of the A rows it stages
    int a_iy0[A_IT], a_ix0[A_IT];
    bool a_ok[A_IT];
    const int c4 = tid & 7;  // which float4 of the 32-channel slice
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        const int row = (tid >> 3) + i * (T / 8);
        const int m = m0 + row;
        a_ok[i] = m < M;
        const int oy = m / a.OW, ox = m - oy * a.OW;
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
    }
    const float* b_ptr[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        const int row = (tid >> 3) + i * (T / 8);
        const int n = n0 + row;
        b_ok[i] = n < a.Cout;
        b_ptr[i] = a.wt + (size_t)(b_ok[i] ? n : 0) * Ktot + c4 * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    float4 ra[A_IT], rb[B_IT];
    const int cchunks = a.Cin / BK;
    const int ksteps = a.KH * a.KW * cchunks;
    int ky = 0, kx = 0, cc = 0;  // coordinates of the K step being LOADED

    auto load_step = [&](int ks) {
        const int dy = ky * a.dil, dx = kx * a.dil;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const size_t off = ((size_t)(ok ? iy : 0) * a.W + (ok ? ix : 0)) * a.Cin + cc * BK + c4 * 4;
            ra[i] = ok ? *reinterpret_cast<const float4*>(a.in + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_IT; i++)
            rb[i] = b_ok[i] ? *reinterpret_cast<const float4*>(b_ptr[i] + (size_t)ks * BK)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        // advance (ky,kx,cc) to the next K step
        if (++cc == cchunks) {
            cc = 0;
            if (++kx == a.KW) {
                kx = 0;
                ++ky;
            }
        }
    };
    auto store_step = [&](int buf) {
        float* Ab = As + buf * BM * LDS_STRIDE;
        float* Bb = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int row = (tid >> 3) + i * (T / 8);
            *reinterpret_cast<float4*>(Ab + row * LDS_STRIDE + c4 * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            const int row = (tid >> 3) + i * (T / 8);
            *reinterpret_cast<float4*>(Bb + row * LDS_STRIDE + c4 * 4) = rb[i];
        }
    };

    load_step(0);
    store_step(0);
    __syncthreads();

    const int a_lds = (wm * TM * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    const int b_lds = (wn * TN * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;

    for (int ks = 0; ks < ksteps; ks++) {
        const int buf = ks & 1;
        if (ks + 1 < ksteps) load_step(ks + 1);

        const float* Ab = As + buf * BM * LDS_STRIDE + a_lds;
        const float* Bb = Bs + buf * BN * LDS_STRIDE + b_lds;
#pragma unroll
        for (int kk = 0; kk < BK / 8; kk++) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; i++)
                fa[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; j++)
                fb[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (ks + 1 < ksteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // epilogue: + bias, + residual, ReLU.  C/D layout of 32x32: col = lane & 31,
    // row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        if (n >= a.Cout) continue;
        const float bv = a.bias[n];
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int m = m0 + row;
                if (m < M) {
                    const size_t o = (size_t)m * a.Cout + n;
                    float v = acc[i][j][e] + bv;
                    if (a.res) v += a.res[o];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    a.out[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + BM - 1) / BM;
    const int ntiles = (a.Cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    auto k = conv_igemm_f32_kernel<BM, BN, WM, WN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(mtiles * ntiles), dim3(WM * WN * 64), lds, s, a, mtiles, ntiles);
    return hipGetLastError();
}

hipError_t launch_conv_igemm_f32(const ConvArgs& a, hipStream_t s) {
    if (a.Cin % BK != 0) return hipErrorInvalidValue;
    if (a.Cout >= 128) return launch_cfg<128, 128, 2, 2>(a, s);
    if (a.Cout > 32) return launch_cfg<128, 64, 2, 2>(a, s);
    return launch_cfg<256, 32, 4, 1>(a, s);
}

}  // namespace infur
