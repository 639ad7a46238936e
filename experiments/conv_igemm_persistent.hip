// conv_igemm.hip -- convolution as an im2col-free implicit GEMM on the gfx950 matrix cores:
// f32 operands on v_mfma_f32_32x32x2_f32 (exact f32, bitwise an fmaf chain) or f16 operands
// on v_mfma_f32_32x32x16_f16 (f32 accumulation).  One kernel source: both instructions take a
// lane's k-slice as one 16-byte register group (4 f32 / 8 f16), so tiles, staging and LDS
// image are described in BYTES of k.
//
// Replaces the Conv nodes ONNX Runtime executes inside `session.run`
// (infur/src/predict_onnx.rs:138) for every 1x1 and 3x3 convolution of FCN-ResNet
// (stride 1/2, dilation 1/2/4), with bias, residual add and ReLU fused into the epilogue.
//
//   GEMM view:  M = OH*OW output pixels, N = Cout, K = KH*KW*Cin  (tap-major, Cin inner)
//   A[m][k]  = in[(oy*s - p + ky*d), (ox*s - p + kx*d), c]   NHWC, gathered, zero padded
//   B[n][k]  = wt[n][ky][kx][c]                               OHWI, k contiguous
//
// Tiling: BM x BN x 128 bytes of k per workgroup, one wave per SIMD, each wave TM x TN tiles of
// 32x32.  Operands are staged global -> VGPR -> LDS (row stride 144 bytes: ds_write_b128 and
// ds_read_b128 both conflict-free) with two LDS buffers and one barrier per K step; global
// loads run two K steps ahead and LDS fragment reads one slice ahead of the MFMAs.
// A lane reads 4 consecutive k of its row with one ds_read_b128 (lanes 0-31: k 0-3,
// lanes 32-63: k 4-7 of an 8-wide slice) and feeds them to 4 MFMAs; A and B use the same
// permutation of k, so the sum is complete.
#include <type_traits>

#include "kernels.h"

namespace infur {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One K step covers ROW_BYTES of every operand row: 32 f32 or 64 f16 channels.  LDS rows are
// padded to 144 B: ds_write_b128 (8-lane groups) and ds_read_b128 (16-lane groups) are then
// both conflict-free.
constexpr int ROW_BYTES = 128;
constexpr int LDS_ROW = ROW_BYTES + 16;

// voffset that is out of range for every tensor this kernel accepts (< 2 GiB): the buffer
// load then returns zeros -- branch-free zero padding / tail predication.
constexpr unsigned OOB = 0x80000000u;

// T = operand type (float: v_mfma_f32_32x32x2_f32, exact f32; _Float16: v_mfma_f32_32x32x16_f16
// with f32 accumulation), OutT = type of the stored activation (f32 for the classifier logits).
template <typename T, typename OutT, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64, 2)
    conv_igemm_kernel(const ConvArgs a, const int mtiles, const int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)  // gfx950 builtins/types below mean nothing in the host pass (it only needs the stub)
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int ES = sizeof(T);              // operand element size
    constexpr int BK = ROW_BYTES / ES;         // channels per K step
    constexpr int NSL = 4;                     // slices per K step (32 bytes of k each)
    constexpr int NT = WM * WN * 64;           // threads
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_IT = BM * 8 / NT;  // 16-byte chunks per thread per K step
    constexpr int B_IT = BN * 8 / NT;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                     // [2][BM][LDS_ROW]
    char* Bs = smem + 2 * BM * LDS_ROW;  // [2][BN][LDS_ROW]

    // Persistent workgroups: the grid holds as many workgroups as stay resident and each walks a
    // list of output tiles, treating ALL its K steps as one stream -- while the epilogue of tile i
    // runs, the first two K steps of tile i+1 are already in flight / in LDS, so the per-tile
    // prologue (cold loads, first barrier) disappears from every tile but the first.
    // XCD-aware tile order: workgroup b runs on XCD b % 8; every XCD owns a contiguous chunk of
    // the tile list (n fastest) so the N-tiles that share an activation tile share one L2.
    const int per_batch = mtiles * ntiles;
    const int nblk = per_batch * (a.batch > 1 ? a.batch : 1);
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int wg_per_xcd = (gridDim.x + 7 - xcd) >> 3;          // workgroups sharing this XCD's chunk
    const int chunk = (nblk + 7) >> 3;                          // tiles per XCD chunk
    const int chunk_lo = xcd * chunk, chunk_hi = chunk_lo + chunk < nblk ? chunk_lo + chunk : nblk;
    const int first_tile = chunk_lo + loc;
    const int my_tiles = first_tile < chunk_hi ? (chunk_hi - first_tile + wg_per_xcd - 1) / wg_per_xcd : 0;
    if (my_tiles == 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int M = a.OH * a.OW;
    const int Ktot = a.KH * a.KW * a.Cin;
    const int cchunks = a.Cin / BK;  // K steps per filter tap
    const int ksteps = a.KH * a.KW * cchunks;
    const int c4 = tid & 7;  // which 16-byte chunk of the 128-byte channel slice

    // ---------------- load stream: (tile, K step) of the NEXT operand slice to fetch ----------------
    int ld_tile = first_tile, ld_ks = 0, ky = 0, kx = 0, cc = 0;
    int a_iy0[A_IT], a_ix0[A_IT];
    unsigned b_off[B_IT];
    // Buffer descriptors: hardware bounds checking turns an out-of-range offset into a zero
    // result, so padding taps and ragged tiles need no branches in the K loop.
    __amdgpu_buffer_rsrc_t in_rsrc, wt_rsrc;
    auto setup_load_tile = [&](int tile) {
        const int bidx = tile / per_batch;  // 0 for a plain convolution
        const int tl = tile - bidx * per_batch;
        const int mt = tl / ntiles, nt = tl - mt * ntiles;
        const int m0 = mt * BM, n0 = nt * BN;
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(a.in) + (size_t)bidx * a.in_bs), 0,
                                                    (unsigned)((size_t)a.H * a.W * a.Cin * ES), 0x00020000);
        wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(a.wt) + (size_t)bidx * a.wt_bs), 0,
                                                    (unsigned)((size_t)a.Cout * Ktot * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            const int m = m0 + row;
            const int oy = m / a.OW, ox = m - oy * a.OW;
            // rows past M get coordinates that fail the bounds test for every tap
            a_iy0[i] = m < M ? oy * a.stride - a.pad : -0x100000;
            a_ix0[i] = ox * a.stride - a.pad;
        }
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            const int n = n0 + row;
            b_off[i] = n < a.Cout ? (unsigned)n * (unsigned)(Ktot * ES) + c4 * 16u : OOB;
        }
        ld_ks = 0;
        ky = kx = cc = 0;
    };
    setup_load_tile(first_tile);

    u32x4 ra[A_IT], rb[B_IT];
    auto load_a = [&]() {
        const int dy = ky * a.dil, dx = kx * a.dil;
        const unsigned coff = (unsigned)(cc * ROW_BYTES + c4 * 16);
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = (unsigned)(iy * a.W + ix) * (unsigned)(a.Cin * ES) + coff;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? off : OOB, 0, 0);
        }
        // advance (ky,kx,cc) to the next K step, branch-free (all wave-uniform scalars)
        cc += 1;
        const int w1 = cc == cchunks;
        cc = w1 ? 0 : cc;
        kx += w1;
        const int w2 = kx == a.KW;
        kx = w2 ? 0 : kx;
        ky += w2;
    };
    auto load_b = [&]() {
        const unsigned koff = (unsigned)ld_ks * (unsigned)ROW_BYTES;
#pragma unroll
        for (int i = 0; i < B_IT; i++)
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc, b_off[i] == OOB ? OOB : b_off[i] + koff, 0, 0);
    };
    // after both operands of a stream step were requested: at the end of a tile's K range point the
    // load stream at the next tile of this workgroup (rare, wave-uniform branch)
    auto advance_load = [&]() {
        if (++ld_ks == ksteps) {
            ld_tile += wg_per_xcd;
            if (ld_tile < chunk_hi) setup_load_tile(ld_tile);
        }
    };
    auto store_a = [&](int buf) {
        char* Ab = As + buf * BM * LDS_ROW;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            *reinterpret_cast<u32x4*>(Ab + row * LDS_ROW + c4 * 16) = ra[i];
        }
    };
    auto store_b = [&](int buf) {
        char* Bb = Bs + buf * BN * LDS_ROW;
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            *reinterpret_cast<u32x4*>(Bb + row * LDS_ROW + c4 * 16) = rb[i];
        }
    };

    // LDS -> register fragments for one 32-byte k slice of buffer `buf`: lanes 0-31 take the
    // first 16 bytes (4 f32 / 8 f16 consecutive k), lanes 32-63 the second
    const int a_lds = (wm * TM * 32 + (lane & 31)) * LDS_ROW + (lane >> 5) * 16;
    const int b_lds = (wn * TN * 32 + (lane & 31)) * LDS_ROW + (lane >> 5) * 16;
    auto read_frags = [&](int buf, int kk, float4 (&fa)[TM], float4 (&fb)[TN]) {
        const char* Ab = As + buf * BM * LDS_ROW + a_lds + kk * 32;
        const char* Bb = Bs + buf * BN * LDS_ROW + b_lds + kk * 32;
#pragma unroll
        for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_ROW);
#pragma unroll
        for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_ROW);
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    };
    zero_acc();

    float4 fa[TM], fb[TN], fa_n[TM], fb_n[TN];

    // One K step = 4 slices of 32 bytes of k.  Software pipeline with ONE barrier per K step, placed
    // mid-step, and no control flow inside the MFMA part of a step, so the scheduler can hide the
    // staging (buffer loads, LDS writes, address arithmetic) in the shadow of the MFMAs:
    //   every slice : the fragments of the next slice (slice 0 of the OTHER buffer after
    //                 slice 3) are read while the MFMAs of this slice (16 f32 / 4 f16) issue;
    //   slice 0 / 1 : registers holding stream step s+1 (activations / weights) -> other LDS
    //                 buffer; then the global loads of stream step s+2 go into the same registers;
    //   slice 2     : s_barrier.  The other buffer is complete before slice 3 reads it, and
    //                 every read of the current buffer has completed (lgkmcnt(0)) before it
    //                 is overwritten one step later.
    // STORE / LOAD / NEXT are compile-time so the steady-state body is straight-line code.
    auto k_step = [&](int buf, auto STORE, auto LOAD, auto NEXT) {
#pragma unroll
        for (int kk = 0; kk < NSL; kk++) {
            if (kk < NSL - 1)
                read_frags(buf, kk + 1, fa_n, fb_n);
            else if (NEXT)
                read_frags(buf ^ 1, 0, fa_n, fb_n);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    // D rows = output channels, D cols = pixels (operands swapped on purpose)
                    if constexpr (F32) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].z, fa[i].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].w, fa[i].w, acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j]),
                                                                         __builtin_bit_cast(f16x8, fa[i]), acc[i][j], 0, 0, 0);
                    }
                }
            // staging spread over two slices: activations at slice 0, weights at slice 1
            if (kk == 0 && STORE) {
                store_a(buf ^ 1);
                if (LOAD) load_a();
            }
            if (kk == 1 && STORE) {
                store_b(buf ^ 1);
                if (LOAD) load_b();
            }
            // Ask the scheduler for an even interleave instead of clusters of LDS/VMEM/VALU
            // work between two MFMAs (a cluster longer than the 64-cycle MFMA shadow is a
            // bubble in this wave's MFMA stream).  Measured +2-3 % on the 3x3 convs.
            // masks: VALU 0x2, MFMA 0x8, VMEM read 0x20, DS read 0x100, DS write 0x200
            if (!F32) {
                // f16: 4 MFMAs of 32 cycles per slice -- the staging cannot hide in their shadow;
                // leave the order to the compiler
            } else if (kk <= 1 && STORE) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
                }
            }
            // the (rare) switch of the load stream to the next tile closes slice 1: the branch sits
            // after the slice's MFMAs and staging
            if (kk == 1 && STORE && LOAD) advance_load();
            if (kk == 2 && STORE) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = fa_n[i];
#pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = fb_n[j];
        }
    };
    constexpr auto Y = std::true_type{};
    constexpr auto N = std::false_type{};

    // epilogue of one tile: + bias, + residual, ReLU.  The MFMA was issued with the weight fragment
    // as the row operand, so in the 32x32 C/D layout (col = lane & 31, row = (e & 3) + 8 * (e >> 2) +
    // 4 * (lane >> 5)) a lane owns ONE pixel (col) and, per group g = e >> 2, FOUR consecutive
    // output channels: NHWC stores, residual loads and bias loads are 16 (f32) / 8 (f16) bytes wide.
    const T* res = static_cast<const T*>(a.res);
    const bool has_bias = a.bias != nullptr;
    const bool vec_ok = (a.Cout & 3) == 0;
    auto epilogue = [&](int tile) {
        const int bidx = tile / per_batch;
        const int tl = tile - bidx * per_batch;
        const int mt = tl / ntiles, nt = tl - mt * ntiles;
        const int m0 = mt * BM, n0 = nt * BN;
        OutT* out = reinterpret_cast<OutT*>(static_cast<char*>(a.out) + (size_t)bidx * a.out_bs);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = m0 + wm * TM * 32 + i * 32 + (lane & 31);
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; j++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int n = n0 + wn * TN * 32 + j * 32 + 8 * g + 4 * (lane >> 5);
                    if (n >= a.Cout) continue;
                    const size_t o = (size_t)m * a.Cout + n;
                    float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (vec_ok) {
                        if (has_bias) {
                            const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
                            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                        }
                        if (res) {
                            if constexpr (F32) {
                                const float4 rv = *reinterpret_cast<const float4*>(res + o);
                                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                            } else {
                                const f16x4 rv = *reinterpret_cast<const f16x4*>(res + o);
                                v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
                            }
                        }
                        if (a.relu) {
                            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                        }
                        if constexpr (std::is_same<OutT, float>::value) {
                            *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                            *reinterpret_cast<f16x4*>(out + o) = hv;
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            if (n + t >= a.Cout) break;
                            float x = v[t] + (has_bias ? a.bias[n + t] : 0.f);
                            if (res) x += (float)res[o + t];
                            if (a.relu) x = fmaxf(x, 0.f);
                            out[o + t] = (OutT)x;
                        }
                    }
                }
            }
        }
    };

    // ---------------- the stream of K steps over all tiles of this workgroup ----------------
    const int S = my_tiles * ksteps;
    load_a();
    load_b();                      // stream step 0 -> registers
    advance_load();
    store_a(0);
    store_b(0);
    if (S > 1) {                   // stream step 1 -> registers
        load_a();
        load_b();
        advance_load();
    }
    __syncthreads();
    read_frags(0, 0, fa, fb);

    int cur_tile = first_tile, buf = 0, s = 0;
    for (int t = 0; t < my_tiles; t++) {
        int ks = 0;
        // steady state: stream steps s+1, s+2 belong to this tile -- tight loop, one instantiation
        for (; ks + 2 < ksteps; ks++, s++) {
            k_step(buf, Y, Y, Y);
            buf ^= 1;
        }
        // the last (up to two) K steps of the tile stage / load into the NEXT tile, if there is one
        for (; ks < ksteps; ks++, s++) {
            if (s + 2 < S)
                k_step(buf, Y, Y, Y);
            else if (s + 1 < S)
                k_step(buf, Y, N, Y);  // last but one step of the stream: nothing left to load
            else
                k_step(buf, N, N, N);  // last step of the stream: nothing left to stage
            buf ^= 1;
        }
        epilogue(cur_tile);
        zero_acc();
        cur_tile += wg_per_xcd;
    }
#endif  // __HIP_DEVICE_COMPILE__
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + BM - 1) / BM;
    const int ntiles = (a.Cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_ROW;
    auto k = conv_igemm_kernel<T, OutT, BM, BN, WM, WN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    // persistent grid: as many workgroups as stay resident (2 per CU while two LDS images fit)
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int resident = cus * (2 * lds <= 160 * 1024 ? 2 : 1);
    const int nblk = mtiles * ntiles * (a.batch > 1 ? a.batch : 1);
    hipLaunchKernelGGL(k, dim3(nblk < resident ? nblk : resident), dim3(WM * WN * 64), lds, s, a, mtiles, ntiles);
    return hipGetLastError();
}

template <typename T, typename OutT>
static hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr size_t ES = sizeof(T);
    if (a.Cin % (int)(ROW_BYTES / ES) != 0) return hipErrorInvalidValue;
    // 32-bit buffer offsets with 0x80000000 as the out-of-range marker
    if ((size_t)a.H * a.W * a.Cin * ES >= 0x80000000ull || (size_t)a.Cout * a.KH * a.KW * a.Cin * ES >= 0x80000000ull)
        return hipErrorInvalidValue;
    if (a.Cout >= 128) return launch_cfg<T, OutT, 128, 128, 2, 2>(a, s);
    if (a.Cout > 32) return launch_cfg<T, OutT, 128, 64, 2, 2>(a, s);
    return launch_cfg<T, OutT, 256, 32, 4, 1>(a, s);
}

hipError_t launch_conv_igemm(const ConvArgs& a, int f16, int out_f32, hipStream_t s) {
    if (!f16) return launch_t<float, float>(a, s);
    return out_f32 ? launch_t<_Float16, float>(a, s) : launch_t<_Float16, _Float16>(a, s);
}

const char* conv_igemm_config(const ConvArgs& a, int f16) {
    if (a.Cout >= 128) return f16 ? "conv_igemm_f16<128,128>" : "conv_igemm_f32<128,128>";
    if (a.Cout > 32) return f16 ? "conv_igemm_f16<128,64>" : "conv_igemm_f32<128,64>";
    return f16 ? "conv_igemm_f16<256,32>" : "conv_igemm_f32<256,32>";
}

}  // namespace infur
