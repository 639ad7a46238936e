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

typedef float f32x4 __attribute__((ext_vector_type(4)));

// four f32 (one staged 16-byte chunk) * scale -> 4 x f16 hi at dst, 4 x f16 lo at dst + 64
// (round to nearest even twice: |x - hi| <= 2^-11 |x| is exact in f32, so hi + lo = x to 2^-22)
__device__ __forceinline__ void store_split(char* dst, const u32x4 r, const float scale) {
    const f32x4 x = __builtin_bit_cast(f32x4, r) * scale;
    const f16x4 hi = __builtin_convertvector(x, f16x4);
    const f16x4 lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), f16x4);
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + 64) = lo;
}

// T = operand type (float: v_mfma_f32_32x32x2_f32, exact f32; _Float16: v_mfma_f32_32x32x16_f16
// with f32 accumulation), OutT = type of the stored activation (f32 for the classifier logits).
// NBUF = 2: double-buffered LDS, one barrier per K step (the latency-optimised form).
// NBUF = 1: one LDS image, two barriers per K step -- half the LDS footprint, so twice the
// workgroups per CU cover each other's stalls (the occupancy-optimised form).
//
// SPLIT (T = float only): f32 tensors in HBM, f16 matrix cores.  Every f32 value x is split into
// hi = f16(x) and lo = f16(x - hi) (both round-to-nearest; the subtraction is exact), so hi + lo carries
// 22 significand bits of x: activations while their tile is staged into LDS (after a power-of-two
// a_scale that keeps lo out of the f16 subnormals), weights once at model load (launch_split_weights).  The product is accumulated in f32 as
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_32x32x16_f16 (the a_lo*b_lo term, 2^-22 relative,
// is dropped): f32-grade results (measured against the f32 oracle in tests/) at three f16 MFMAs per
// f32 MFMA-equivalent, i.e. a ceiling of 2.5 PFLOP/s / 3 = 833 TFLOP/s instead of 157.  An LDS row is
// [hi: 32 x f16][lo: 32 x f16] (128 bytes, same as the f32 row) and a K step is two 16-wide MFMA slices.
template <typename T, typename OutT, int BM, int BN, int WM, int WN, int NBUF, bool SPLIT = false>
__global__ void __launch_bounds__(WM* WN * 64, 2)
    conv_igemm_kernel(const ConvArgs a, const int mtiles, const int ntiles) {
    constexpr bool F32 = std::is_same<T, float>::value && !SPLIT;
    static_assert(!SPLIT || std::is_same<T, float>::value, "SPLIT stages f32 tensors");
    constexpr int ES = sizeof(T);              // operand element size
    constexpr int BK = ROW_BYTES / ES;         // channels per K step
    constexpr int NSL = SPLIT ? 2 : 4;         // slices per K step (32 bytes of k each; SPLIT: 16 k as hi + lo)
    constexpr int NF = SPLIT ? 2 : 1;          // fragment planes per row block (SPLIT: hi, lo)
    constexpr int NT = WM * WN * 64;           // threads
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_IT = BM * 8 / NT;  // 16-byte chunks per thread per K step
    constexpr int B_IT = BN * 8 / NT;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                        // [NBUF][BM][LDS_ROW]
    char* Bs = smem + NBUF * BM * LDS_ROW;  // [NBUF][BN][LDS_ROW]

    // XCD-aware tile order: block b runs on XCD b % 8; give every XCD a contiguous run of
    // tiles (n fastest) so the N-tiles that share an activation tile share one L2.
    const int nblk = mtiles * ntiles * (a.batch > 1 ? a.batch : 1);
    int tile;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, loc = b >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int per_batch = mtiles * ntiles;
    const int bidx = tile / per_batch;  // 0 for a plain convolution
    tile -= bidx * per_batch;
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const char* in_base = static_cast<const char*>(a.in) + (size_t)bidx * a.in_bs;
    const char* wt_base = static_cast<const char*>(a.wt) + (size_t)bidx * a.wt_bs;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int M = a.OH * a.OW;
    const int Ktot = a.KH * a.KW * a.Cin;

    // Buffer descriptors: hardware bounds checking turns an out-of-range offset into a
    // zero result, so padding taps and ragged tiles need no branches in the K loop.
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(in_base), 0, (unsigned)((size_t)a.H * a.W * a.Cin * ES), 0x00020000);
    const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wt_base), 0, (unsigned)((size_t)a.Cout * Ktot * ES), 0x00020000);

    // per-thread gather coordinates of the A rows it stages
    int a_iy0[A_IT], a_ix0[A_IT];
    const int c4 = tid & 7;  // which 16-byte chunk of the 128-byte channel slice
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        const int row = (tid >> 3) + i * (NT / 8);
        const int m = m0 + row;
        const int oy = m / a.OW, ox = m - oy * a.OW;
        // rows past M get coordinates that fail the bounds test for every tap
        a_iy0[i] = m < M ? oy * a.stride - a.pad : -0x100000;
        a_ix0[i] = ox * a.stride - a.pad;
    }
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        const int row = (tid >> 3) + i * (NT / 8);
        const int n = n0 + row;
        b_off[i] = n < a.Cout ? (unsigned)n * (unsigned)(Ktot * ES) + c4 * 16u : OOB;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    u32x4 ra[A_IT], rb[B_IT];
    const int cchunks = a.Cin / BK;  // K steps per filter tap
    const int ksteps = a.KH * a.KW * cchunks;
    int ky = 0, kx = 0, cc = 0;  // coordinates of the K step being LOADED

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
    auto load_b = [&](int ks) {
        const unsigned koff = (unsigned)ks * (unsigned)ROW_BYTES;
#pragma unroll
        for (int i = 0; i < B_IT; i++)
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc, b_off[i] == OOB ? OOB : b_off[i] + koff, 0, 0);
    };
    auto load_step = [&](int ks) {
        load_a();
        load_b(ks);
    };
    auto store_a = [&](int buf) {
        char* Ab = As + buf * BM * LDS_ROW;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            if constexpr (SPLIT)
                store_split(Ab + row * LDS_ROW + c4 * 8, ra[i], a.a_scale);
            else
                *reinterpret_cast<u32x4*>(Ab + row * LDS_ROW + c4 * 16) = ra[i];
        }
    };
    auto store_b = [&](int buf) {
        char* Bb = Bs + buf * BN * LDS_ROW;
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            *reinterpret_cast<u32x4*>(Bb + row * LDS_ROW + c4 * 16) = rb[i];  // SPLIT: split at load time
        }
    };
    auto store_step = [&](int buf) {
        store_a(buf);
        store_b(buf);
    };

    // LDS -> register fragments for one 32-byte k slice of buffer `buf`: lanes 0-31 take the
    // first 16 bytes (4 f32 / 8 f16 consecutive k), lanes 32-63 the second
    const int a_lds = (wm * TM * 32 + (lane & 31)) * LDS_ROW + (lane >> 5) * 16;
    const int b_lds = (wn * TN * 32 + (lane & 31)) * LDS_ROW + (lane >> 5) * 16;
    auto read_frags = [&](int buf, int kk, float4 (&fa)[TM * NF], float4 (&fb)[TN * NF]) {
        const char* Ab = As + buf * BM * LDS_ROW + a_lds + kk * 32;
        const char* Bb = Bs + buf * BN * LDS_ROW + b_lds + kk * 32;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int p = 0; p < NF; p++) fa[i * NF + p] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_ROW + p * 64);
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int p = 0; p < NF; p++) fb[j * NF + p] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_ROW + p * 64);
    };

    float4 fa[TM * NF], fb[TN * NF], fa_n[TM * NF], fb_n[TN * NF];

    // the MFMAs of one slice.  D rows = output channels, D cols = pixels (operands swapped on purpose)
    auto mma_slice = [&]() {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                if constexpr (F32) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].z, fa[i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].w, fa[i].w, acc[i][j], 0, 0, 0);
                } else if constexpr (SPLIT) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, fa[2 * i]), al = __builtin_bit_cast(f16x8, fa[2 * i + 1]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, fb[2 * j]), bl = __builtin_bit_cast(f16x8, fb[2 * j + 1]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j]),
                                                                     __builtin_bit_cast(f16x8, fa[i]), acc[i][j], 0, 0, 0);
                }
            }
    };

    // One K step = 4 slices of 32 bytes of k.  Software pipeline with ONE barrier per K step, placed
    // mid-step, and no control flow inside a step, so the scheduler can hide the staging
    // (buffer loads, LDS writes, address arithmetic) in the shadow of the 64-cycle MFMAs:
    //   every slice : the fragments of the next slice (slice 0 of the OTHER buffer after
    //                 slice 3) are read while the MFMAs of this slice (16 f32 / 4 f16) issue;
    //   slice 0 / 1 : registers holding K step ks+1 (activations / weights) -> other LDS
    //                 buffer; then the global loads of K step ks+2 go into the same registers;
    //   slice 2     : s_barrier.  The other buffer is complete before slice 3 reads it, and
    //                 every read of the current buffer has completed (lgkmcnt(0)) before it
    //                 is overwritten one step later.
    // STORE / LOAD / NEXT are compile-time so the steady-state body is straight-line code.
    auto k_step = [&](int ks, auto STORE, auto LOAD, auto NEXT) {
        const int buf = ks & 1;
#pragma unroll
        for (int kk = 0; kk < NSL; kk++) {
            if (kk < NSL - 1)
                read_frags(buf, kk + 1, fa_n, fb_n);
            else if (NEXT)
                read_frags(buf ^ 1, 0, fa_n, fb_n);
            mma_slice();
            // staging spread over two slices: activations at slice 0, weights at slice 1
            if (kk == 0 && STORE) {
                store_a(buf ^ 1);
                if (LOAD) load_a();
            }
            if (kk == (SPLIT ? 0 : 1) && STORE) {
                store_b(buf ^ 1);
                if (LOAD) load_b(ks + 2);
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
            if (kk == NSL - 2 && STORE) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int i = 0; i < TM * NF; i++) fa[i] = fa_n[i];
#pragma unroll
            for (int j = 0; j < TN * NF; j++) fb[j] = fb_n[j];
        }
    };
    constexpr auto Y = std::true_type{};
    constexpr auto N = std::false_type{};

    load_step(0);
    store_step(0);
    if (ksteps > 1) load_step(1);
    __syncthreads();

    if constexpr (NBUF == 2) {
        read_frags(0, 0, fa, fb);
        int ks = 0;
        for (; ks + 2 < ksteps; ks++) k_step(ks, Y, Y, Y);  // steady state
        if (ks + 1 < ksteps) k_step(ks++, Y, N, Y);          // last but one: nothing left to load
        k_step(ks, N, N, N);                                 // last: nothing left to stage
    } else {
        // single LDS image: compute a K step, barrier, overwrite the image with the registers
        // (K step ks+1), refill the registers (ks+2), barrier.  Fragment prefetch only within a step.
        for (int ks = 0; ks < ksteps; ks++) {
            read_frags(0, 0, fa, fb);
#pragma unroll
            for (int kk = 0; kk < NSL; kk++) {
                if (kk < NSL - 1) read_frags(0, kk + 1, fa_n, fb_n);
                mma_slice();
#pragma unroll
                for (int i = 0; i < TM * NF; i++) fa[i] = fa_n[i];
#pragma unroll
                for (int j = 0; j < TN * NF; j++) fb[j] = fb_n[j];
            }
            if (ks + 1 < ksteps) {
                __syncthreads();  // every wave has read the image
                store_step(0);
                if (ks + 2 < ksteps) load_step(ks + 2);
                __syncthreads();  // the new image is complete
            }
        }
    }

    // epilogue: + bias, + residual, ReLU.  The MFMA was issued with the weight fragment as the
    // row operand, so in the 32x32 C/D layout (col = lane & 31, row = (e & 3) + 8 * (e >> 2) +
    // 4 * (lane >> 5)) a lane owns ONE pixel (col) and, per group g = e >> 2, FOUR consecutive
    // output channels: NHWC stores, residual loads and bias loads are 16 (f32) / 8 (f16) bytes wide.
    const T* res = static_cast<const T*>(a.res);
    OutT* out = reinterpret_cast<OutT*>(static_cast<char*>(a.out) + (size_t)bidx * a.out_bs);
    const bool has_bias = a.bias != nullptr;
    const bool vec_ok = (a.Cout & 3) == 0;
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
                if constexpr (SPLIT) {
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] *= a.acc_scale;
                }
                if (vec_ok) {
                    if (has_bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
                        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                    }
                    if (res) {
                        if constexpr (std::is_same<T, float>::value) {
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
}

template <typename T, typename OutT, bool SPLIT, int BM, int BN, int WM, int WN, int NBUF = 2>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + BM - 1) / BM;
    const int ntiles = (a.Cout + BN - 1) / BN;
    const size_t lds = (size_t)NBUF * (BM + BN) * LDS_ROW;
    auto k = conv_igemm_kernel<T, OutT, BM, BN, WM, WN, NBUF, SPLIT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(mtiles * ntiles * (a.batch > 1 ? a.batch : 1)), dim3(WM * WN * 64), lds, s, a, mtiles, ntiles);
    return hipGetLastError();
}

// ---- tile configurations ----
// All configurations accumulate k in the same order for every output element, so the choice only
// changes speed, never a single bit of the result.
struct CfgInfo {
    int bm, bn;
    const char* name[3];  // per mode: f32, f16, f32 split into f16 pairs
};
static const CfgInfo kCfgs[] = {
    {128, 128, {"conv_igemm_f32<128,128>", "conv_igemm_f16<128,128>", "conv_igemm_f32s<128,128>"}},
    {64, 128, {"conv_igemm_f32<64,128>", "conv_igemm_f16<64,128>", "conv_igemm_f32s<64,128>"}},
    {128, 64, {"conv_igemm_f32<128,64>", "conv_igemm_f16<128,64>", "conv_igemm_f32s<128,64>"}},
    {64, 64, {"conv_igemm_f32<64,64>", "conv_igemm_f16<64,64>", "conv_igemm_f32s<64,64>"}},
    {256, 32, {"conv_igemm_f32<256,32>", "conv_igemm_f16<256,32>", "conv_igemm_f32s<256,32>"}},
    {128, 256, {"conv_igemm_f32<128,256>", "conv_igemm_f16<128,256>", "conv_igemm_f32s<128,256>"}},
    {256, 128, {"conv_igemm_f32<256,128>", "conv_igemm_f16<256,128>", "conv_igemm_f32s<256,128>"}},
    {128, 128, {"conv_igemm_f32<128,128,1buf>", "conv_igemm_f16<128,128,1buf>", "conv_igemm_f32s<128,128,1buf>"}},
    {128, 64, {"conv_igemm_f32<128,64,1buf>", "conv_igemm_f16<128,64,1buf>", "conv_igemm_f32s<128,64,1buf>"}},
    {64, 128, {"conv_igemm_f32<64,128,1buf>", "conv_igemm_f16<64,128,1buf>", "conv_igemm_f32s<64,128,1buf>"}},
    {64, 64, {"conv_igemm_f32<64,64,1buf>", "conv_igemm_f16<64,64,1buf>", "conv_igemm_f32s<64,64,1buf>"}},
    {256, 256, {"conv_igemm_f32<256,256>", "conv_igemm_f16<256,256>", "conv_igemm_f32s<256,256>"}},
};
constexpr int kNumCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));

int conv_igemm_num_configs() { return kNumCfgs; }

const char* conv_igemm_config_name(int cfg, int mode) {
    if (cfg < 0 || cfg >= kNumCfgs || mode < 0 || mode > 2) return "conv_igemm<?>";
    return kCfgs[cfg].name[mode];
}

// the heuristic used when no measurement is available
int conv_igemm_default_config(const ConvArgs& a) {
    if (a.Cout >= 128) return 0;
    if (a.Cout > 32) return 2;
    return 4;
}

// a configuration is a candidate when its N tile is not mostly padding
bool conv_igemm_config_valid(const ConvArgs& a, int cfg) {
    if (cfg < 0 || cfg >= kNumCfgs) return false;
    const int bn = kCfgs[cfg].bn;
    if (a.Cout <= 32) return bn == 32;
    if (bn == 32) return false;
    return bn <= a.Cout || bn == 64;  // Cout = 64 -> BN 64 only; Cout >= 128 -> 64 and 128 (and 256 when Cout >= 256)
}

template <typename T, typename OutT, bool SPLIT = false>
static hipError_t launch_t(const ConvArgs& a, int cfg, hipStream_t s) {
    constexpr size_t ES = sizeof(T);
    if (a.Cin % (int)(ROW_BYTES / ES) != 0) return hipErrorInvalidValue;
    // 32-bit buffer offsets with 0x80000000 as the out-of-range marker
    if ((size_t)a.H * a.W * a.Cin * ES >= 0x80000000ull || (size_t)a.Cout * a.KH * a.KW * a.Cin * ES >= 0x80000000ull)
        return hipErrorInvalidValue;
    if (cfg < 0) cfg = conv_igemm_default_config(a);
    switch (cfg) {
        case 0: return launch_cfg<T, OutT, SPLIT, 128, 128, 2, 2>(a, s);
        case 1: return launch_cfg<T, OutT, SPLIT, 64, 128, 2, 2>(a, s);
        case 2: return launch_cfg<T, OutT, SPLIT, 128, 64, 2, 2>(a, s);
        case 3: return launch_cfg<T, OutT, SPLIT, 64, 64, 2, 2>(a, s);
        case 4: return launch_cfg<T, OutT, SPLIT, 256, 32, 4, 1>(a, s);
        case 5: return launch_cfg<T, OutT, SPLIT, 128, 256, 2, 4>(a, s);
        case 6: return launch_cfg<T, OutT, SPLIT, 256, 128, 4, 2>(a, s);
        case 7: return launch_cfg<T, OutT, SPLIT, 128, 128, 2, 2, 1>(a, s);
        case 8: return launch_cfg<T, OutT, SPLIT, 128, 64, 2, 2, 1>(a, s);
        case 9: return launch_cfg<T, OutT, SPLIT, 64, 128, 2, 2, 1>(a, s);
        case 10: return launch_cfg<T, OutT, SPLIT, 64, 64, 2, 2, 1>(a, s);
        case 11: return launch_cfg<T, OutT, SPLIT, 256, 256, 2, 4>(a, s);  // 8 waves of 128x64: least staging per MFMA
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_conv_igemm(const ConvArgs& a, int mode, int out_f32, int cfg, hipStream_t s) {
    if (mode == 0) return launch_t<float, float>(a, cfg, s);
    if (mode == 2) return launch_t<float, float, true>(a, cfg, s);
    return out_f32 ? launch_t<_Float16, float>(a, cfg, s) : launch_t<_Float16, _Float16>(a, cfg, s);
}

}  // namespace infur
