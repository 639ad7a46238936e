// conv_igemm.hip -- convolution as an im2col-free implicit GEMM on the gfx950 matrix cores, three arithmetic modes
// from one kernel source:
//   f32   operands on v_mfma_f32_32x32x2_f32 (exact f32, bitwise an fmaf chain)
//   f16   operands on v_mfma_f32_32x32x16_f16, f32 accumulation
//   split f32 tensors, each value as an f16 hi + lo pair, three f16 MFMAs per product, f32 accumulation (SPLIT below)
// Both instructions take a lane's k-slice as one 16-byte register group (4 f32 / 8 f16), so tiles, staging and
// the LDS image are described in BYTES of k.
//
// Replaces the Conv nodes ONNX Runtime executes inside `session.run` (infur/src/predict_onnx.rs:138) for every
// 1x1 and 3x3 convolution of FCN-ResNet (stride 1/2, dilation 1/2/4), with bias, residual add and ReLU fused
// into the epilogue, and runs the batched Winograd-domain GEMMs of winograd.hip.
//
//   GEMM view:  M = OH*OW output pixels, N = Cout, K = KH*KW*Cin  (tap-major, Cin inner)
//   A[m][k]  = in[(oy*s - p + ky*d), (ox*s - p + kx*d), c]   NHWC, gathered, zero padded
//   B[n][k]  = wt[n][ky][kx][c]                               OHWI, k contiguous
//
// Tiling: BM x BN x 128 bytes of k per workgroup, 4 or 8 waves, each wave TM x TN tiles of 32x32.  Operands are
// staged global -> VGPR -> LDS (row stride 144 bytes: ds_write_b128 and ds_read_b128 both conflict-free); three
// loop forms (NBUF): two LDS images + fragment prefetch + one mid-step barrier; one LDS image, two barriers (half
// the LDS, more workgroups per CU); two images with one fragment set (big tiles).  A lane reads 4 consecutive k
// of its row with one ds_read_b128 (lanes 0-31: k 0-3, lanes 32-63: k 4-7 of an 8-wide slice) and feeds them to
// 4 MFMAs; A and B use the same permutation of k, so the sum is complete.  Template flags select the addressing
// form (G1: 1x1 GEMM), the residual prefetch (RESPF) and the two-source form (DUAL: conv3 + downsample branch).
// Every tile configuration accumulates k in the same order: they are bit-identical, the choice is a speed knob
// (pick_cfg in infur_capi.cpp measures it per layer shape).
//
// This header holds the kernel template and its launchers; it is compiled once per arithmetic mode (conv_igemm_f32.hip,
// conv_igemm_f16.hip, conv_igemm_split.hip, conv_igemm_i8.hip: the instantiations of one mode each, built in parallel), and
// conv_igemm.hip holds the configuration table and the dispatch.
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "qepilogue.h"

namespace infur {

// Instrumentation build (make EXTRA="-DKTRACE -DKT_CIN=512 -DKT_COUT=2048", scripts/ktrace.py): the first 8
// workgroups of every launch with Cin == KT_CIN and Cout == KT_COUT record, per wave, the shader cycles
// (s_memtime) of prologue, K loop and epilogue.  This is how the epilogue of the residual 1x1 convs was
// found to outlast their K loop.
#ifdef KTRACE
#ifndef KT_CIN
#define KT_CIN 1024
#define KT_COUT 2048
#endif
// (one buffer per translation unit -- the kernel is instantiated per mode in conv_igemm_<mode>.hip; ktrace_read in conv_igemm.hip
//  returns the one that was written last)
static __device__ unsigned long long g_ktrace[8 * 8 * 4];
static inline hipError_t ktrace_read_tu(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ktrace), sizeof(g_ktrace)); }
#endif

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
typedef int i32x4q __attribute__((ext_vector_type(4)));
typedef int i32x16q __attribute__((ext_vector_type(16)));

// (quantised epilogue, mode 4: qepilogue.h)

// four f32 (one staged 16-byte chunk) * scale -> 4 x f16 hi at dst, 4 x f16 lo at dst + 64
// (round to nearest even twice: |x - hi| <= 2^-11 |x| is exact in f32, so hi + lo = x to 2^-22)
__device__ __forceinline__ void store_split(char* dst, const u32x4 r, const float scale) {
    const f32x4 x = __builtin_bit_cast(f32x4, r) * scale;
    const f16x4 hi = __builtin_convertvector(x, f16x4);
    const f16x4 lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), f16x4);
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + 64) = lo;
}

// FP8X (f16 + bf8 cross terms): four f32 * scale -> 4 x f16 hi at dst, 4 x e5m2 of hi / 2 at row + 64 + 4 * chunk,
// 4 x e5m2 of (x - hi) * 2^10 at row + 96 + 4 * chunk.
// Round 4: the cross terms' operands are bf8 (OCP e5m2), not e4m3.  e5m2 has f16's own exponent range, so NO tensor-level scale is
// involved -- e4m3 under a per-tensor power of two lost the lo correction of every element more than ~2^15 below the tensor's
// maximum and CLAMPED the elements above 448 / a_scale (the always-on channels of a BN-folded network: exactly the products that
// dominate a sum): 4.9e-4 max-abs / 3.2e-2 per-element on the hostile parameter set against 7.7e-5 / 6.0e-3 for e5m2
// (scripts/sim_hi_lo8.py; on the friendly synthetic weights e4m3's extra mantissa bit wins, 3.8e-5 against 7.6e-5 -- robustness
// was chosen).  The hi copy is halved so that the largest f16 (65504 > e5m2's 57344) converts without any reliance on the
// conversion's overflow behaviour; |lo| <= 2^-11 |hi|, so lo * 2^10 <= |hi| / 2 as well.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int pack_fp8x4(const f32x4 v) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w, true);
    return w;
}
// activations: [hi / 2][lo * 2^10]; weights (launch_split_weights): [lo * 2^11][hi] -- both halves of the 64-deep dot product
// a_hi8 * w_lo8 + a_lo8 * w_hi8 then carry 2^10, undone by the instruction's E8M0 block scale
constexpr float kFp8HiScale = 0.5f, kFp8LoScale = 1024.0f;
constexpr int kFp8CrossScaleA = 127 - 10;
constexpr int kFp8Fmt = 1;  // cbsz / blgp of v_mfma_scale_f32_32x32x64_f8f6f4: 0 = e4m3, 1 = e5m2
__device__ __forceinline__ void store_split_fp8(char* row, const int chunk, const u32x4 r, const float scale) {
    const f32x4 x = __builtin_bit_cast(f32x4, r) * scale;
    const f16x4 hi = __builtin_convertvector(x, f16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    *reinterpret_cast<f16x4*>(row + chunk * 8) = hi;
    *reinterpret_cast<int*>(row + 64 + chunk * 4) = pack_fp8x4(hf * kFp8HiScale);
    *reinterpret_cast<int*>(row + 96 + chunk * 4) = pack_fp8x4((x - hf) * kFp8LoScale);
}

// dynamic LDS of a workgroup: the operand images, or the epilogue's per-wave staging slices
// (32 pixel rows of BN / WN f32 channels + 16 bytes) if those need more
// NBUF == 4 (the LDS-DMA form): rows are the bare 128 bytes -- a DMA piece lands lane-linear, so there is no room for
// padding; bank conflicts are avoided by an XOR swizzle of the 16-byte chunk index instead (lds_swz)
constexpr int lds_row_bytes(int nbuf) { return nbuf >= 4 ? ROW_BYTES : LDS_ROW; }
__host__ __device__ constexpr int lds_swz(int row) { return (row >> 1) & 7; }

constexpr int lds_bytes(int bm, int bn, int wm, int wn, int nbuf) {
    const int operands = (nbuf == 1 ? 1 : 2) * (bm + bn) * lds_row_bytes(nbuf);
    const int staging = wm * wn * 32 * (bn / wn * 4 + 16);
    return operands > staging ? operands : staging;
}

// waves per SIMD the register budget is planned for: what the LDS footprint lets a CU hold, at most 2
constexpr int min_waves_per_simd(int bm, int bn, int wm, int wn, int nbuf) {
    const int waves = wm * wn;
    const int blocks = 160 * 1024 / lds_bytes(bm, bn, wm, wn, nbuf);
    const int w = blocks * waves / 4;
    return w < 1 ? 1 : (w > 2 ? 2 : w);
}

// LDS-DMA: 16 bytes per lane straight from HBM/L2 into LDS at (wave-uniform byte address in M0) + lane*16, no staging
// VGPRs and no ds_write pass; a byte offset beyond the descriptor's num_records lands zeros (the same branch-free
// padding as the register path).  Inline asm on purpose: hipcc (ROCm 7.2) drains every LDS-DMA it knows about with
// vmcnt(0) before the next ds_read; this form is ordered by our own vmcnt + barrier instead.  M0 is compiler-reserved:
// saved and restored inside the statement.
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const u32x4r rsrc, const unsigned lds, const unsigned voff, const unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds), "s"(rsrc), "s"(soff)
        : "memory");
}
typedef __attribute__((address_space(3))) void lds_void_t;

// T = operand type (float: v_mfma_f32_32x32x2_f32, exact f32; _Float16: v_mfma_f32_32x32x16_f16
// with f32 accumulation), OutT = type of the stored activation (f32 for the classifier logits).
// NBUF = 2: double-buffered LDS, one barrier per K step (the latency-optimised form).
// NBUF = 3: two LDS images, one barrier per K step, fragments NOT double-buffered (big tiles: 8 waves keep 128
// accumulators each and the register file has no room for a second fragment set).
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
// G1: the convolution is 1x1 without padding (a plain GEMM over pixels, any stride): every staged row
// has ONE per-lane byte offset for the whole K loop and the K step advances through the scalar offset of
// the buffer load -- no vector address arithmetic in the loop (it competes with the MFMAs for the
// SIMD's issue port, which is what bounds the f16-rate modes).
// RESPF: the layer has a residual input and the tile keeps <= 64 accumulators per lane: the residual
// tile is fetched into registers right after the prologue, so its latency hides behind the whole K
// loop and the epilogue only adds and stores (stores need no waiting: the wave retires at once and
// its CU slot starts the next tile).  Without it the epilogue of a 1x1 conv with K = 512 takes
// longer than its K loop (measured with s_memtime: 36.5k vs 30.6k cycles).
// DUAL (with G1): the K loop runs over two activation tensors in turn (ConvArgs.in, then ConvArgs.in2 sampled
// with stride2) against one weight matrix whose rows are the two 1x1 kernels side by side: conv3 and the
// downsample branch of a bottleneck's first block in one launch, without writing and re-reading the branch.
template <typename T, typename OutT, int BM, int BN, int WM, int WN, int NBUF, bool SPLIT = false, bool G1 = false, bool RESPF = false,
          bool DUAL = false, bool FP8X = false>
__global__ void __launch_bounds__(WM* WN * 64, min_waves_per_simd(BM, BN, WM, WN, NBUF))
    conv_igemm_kernel(const ConvArgs a, const int mtiles, const int ntiles) {
    constexpr bool F32 = std::is_same<T, float>::value && !SPLIT;
    constexpr bool I8 = std::is_same<T, signed char>::value;  // u8 activations x s8 weights, i32 accumulation (ConvArgs::q_*)
    static_assert(!SPLIT || std::is_same<T, float>::value, "SPLIT stages f32 tensors");
    constexpr int ES = sizeof(T);              // operand element size
    constexpr int BK = ROW_BYTES / ES;         // channels per K step
    constexpr int NSL = SPLIT ? 2 : 4;         // slices per K step (32 bytes of k each; SPLIT: 16 k as hi + lo)
    // FP8X (with SPLIT): the cross terms ah*bl + al*bh run on the bf8 (e5m2) MX MFMA (one 32x32x64 per K step)
    static_assert(!FP8X || SPLIT, "FP8X is a form of the split mode");
    constexpr int NF = (SPLIT && !FP8X) ? 2 : 1;  // f16 fragment planes per row block (SPLIT: hi, lo)
    constexpr int NT = WM * WN * 64;           // threads
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_IT = BM * 8 / NT;  // 16-byte chunks per thread per K step
    constexpr int B_IT = BN * 8 / NT;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");

#ifdef KTRACE
    const unsigned long long kt_start = __builtin_amdgcn_s_memtime();
#endif
    // SPLIT: MODE.FP16_OVFL = 1 -- an f32 -> f16 conversion that overflows clamps to +-65504 instead of
    // producing inf, so an activation beyond the f16 pair's range (|x| * a_scale > 131008) saturates
    // instead of poisoning the accumulators with inf - inf
    if constexpr (SPLIT) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // NBUF == 4: operand tiles travel HBM/L2 -> LDS by DMA (dma16), two images, one barrier per K step; rows are the
    // bare 128 bytes with the 16-byte chunk index XOR-swizzled by lds_swz(row) -- applied to the SOURCE address of the
    // DMA (a piece lands lane-linear) and to the fragment reads.  No staging registers, no ds_write pass.
    constexpr bool GLDS = NBUF >= 4;  // 4: DMA pieces of the next K step issued up front, 5: between the slices
    static_assert(!GLDS || (!F32 && !SPLIT), "the LDS-DMA form is built for the f16 operands");
    constexpr int LR = lds_row_bytes(NBUF);  // LDS row stride
    char* As = smem;                        // [NIMG][BM][LR]
    constexpr int NIMG = NBUF == 1 ? 1 : 2;     // LDS images (NBUF 3 = two images, fragments single-buffered)
    char* Bs = smem + NIMG * BM * LR;  // [NIMG][BN][LR]

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

    static_assert(!DUAL || (G1 && !RESPF), "DUAL is a form of the 1x1 GEMM addressing");
    // f16 KxK convolutions walk K with the TAPS INSIDE each 128-byte channel chunk (chunk-major): the order in which
    // conv3x3_halo.hip -- whose input patch stays in LDS for all nine taps of a chunk -- has to sum, so that it remains one more
    // bit-identical configuration.  For this kernel the order is cost-neutral (round 3 measured it: 154.7 vs 155.6 us).
    constexpr bool CMAJ = std::is_same<T, _Float16>::value && !G1 && !SPLIT;
    const int M = a.OH * a.OW;
    const int Ktot = DUAL ? a.Cin + a.Cin2 : a.KH * a.KW * a.Cin;

    // Buffer descriptors: hardware bounds checking turns an out-of-range offset into a
    // zero result, so padding taps and ragged tiles need no branches in the K loop.
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(in_base), 0, (unsigned)((size_t)a.H * a.W * a.Cin * ES), 0x00020000);
    const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wt_base), 0, (unsigned)((size_t)a.Cout * Ktot * ES), 0x00020000);

    // per-thread gather coordinates of the A rows it stages.  Register path: thread (tid >> 3) + i * NT/8 is the tile
    // row, tid & 7 the 16-byte chunk.  DMA path: a wave instruction fills 8 whole rows (1 KB); wave w owns the row
    // groups w * IT + i, lane l row l >> 3 of the group at LDS chunk position l & 7, i.e. data chunk (l & 7) ^ swz(row).
    int a_iy0[A_IT], a_ix0[A_IT];
    auto st_row = [&](int i, int it) { return GLDS ? 8 * (wave * it + i) + (lane >> 3) : (tid >> 3) + i * (NT / 8); };
    auto st_chunk = [&](int row) { return GLDS ? ((lane & 7) ^ lds_swz(row)) : (tid & 7); };
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        const int row = st_row(i, A_IT);
        const int c4 = st_chunk(row);
        const int m = m0 + row;
        const int oy = m / a.OW, ox = m - oy * a.OW;
        // rows past M get coordinates that fail the bounds test for every tap
        a_iy0[i] = m < M ? oy * a.stride - a.pad : -0x100000;
        a_ix0[i] = ox * a.stride - a.pad;
        if constexpr (G1)  // reuse a_iy0 as the fixed byte offset of the row's pixel
            a_iy0[i] = m < M ? (int)((unsigned)(oy * a.stride * a.W + ox * a.stride) * (unsigned)(a.Cin * ES) + c4 * 16u) : (int)OOB;
        if constexpr (DUAL)  // and a_ix0 as the offset of the same output pixel in the second tensor
            a_ix0[i] = m < M ? (int)((unsigned)(oy * a.stride2 * a.W2 + ox * a.stride2) * (unsigned)(a.Cin2 * ES) + c4 * 16u) : (int)OOB;
    }
    const auto in2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(DUAL ? a.in2 : a.in), 0,
                                                            DUAL ? (unsigned)((size_t)a.H2 * a.W2 * a.Cin2 * ES) : 0u, 0x00020000);
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        const int row = st_row(i, B_IT);
        const int n = n0 + row;
        b_off[i] = n < a.Cout ? (unsigned)n * (unsigned)(Ktot * ES) + st_chunk(row) * 16u : OOB;
    }
    // DMA path: the descriptors as plain SGPR quadruples for the asm statement, and the LDS byte address of smem
    u32x4r in_v = {}, in2_v = {}, wt_v = {};
    unsigned lds0 = 0;
    int ld_buf = 0;  // LDS image the next load_step fills
    if constexpr (GLDS) {
        auto mk = [](const void* p, unsigned bytes) {
            const unsigned long long v = reinterpret_cast<unsigned long long>(p);
            u32x4r r;
            r.x = __builtin_amdgcn_readfirstlane((unsigned)v);
            r.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
            r.z = __builtin_amdgcn_readfirstlane(bytes);
            r.w = 0x00020000u;
            return r;
        };
        in_v = mk(in_base, (unsigned)((size_t)a.H * a.W * a.Cin * ES));
        wt_v = mk(wt_base, (unsigned)((size_t)a.Cout * Ktot * ES));
        in2_v = mk(DUAL ? a.in2 : a.in, DUAL ? (unsigned)((size_t)a.H2 * a.W2 * a.Cin2 * ES) : 0u);
        lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    }
    // one 16-byte piece of operand A (row group i of this wave) / B: register load, or DMA into image ld_buf
    auto dst_a = [&](int i) { return __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(ld_buf * BM * LR + (wave * A_IT + i) * 1024)); };
    auto dst_b = [&](int i) { return __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(NIMG * BM * LR + ld_buf * BN * LR + (wave * B_IT + i) * 1024)); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // staging registers: K step k + 1 waits here while step k is multiplied; the global loads of
    // step k + 2 refill them right after they were written to LDS.  (A second set -- loads three
    // steps ahead -- was measured for the split mode: no gain, and 128x128 tiles spill.)
    u32x4 ra[A_IT], rb[B_IT];
    const int cchunks = a.Cin / BK;  // K steps per filter tap
    const int ksteps = DUAL ? cchunks + a.Cin2 / BK : a.KH * a.KW * cchunks;
    int ky = 0, kx = 0, cc = 0;  // coordinates of the K step being LOADED

    int kload = 0;  // G1: K step the next load_a fetches
    auto load_a = [&]() {
        if constexpr (G1) {
            if (DUAL && kload >= cchunks) {  // wave-uniform: the second tensor's K steps
                const unsigned so = (unsigned)(kload - cchunks) * (unsigned)ROW_BYTES;
#pragma unroll
                for (int i = 0; i < A_IT; i++) {
                    if constexpr (GLDS)
                        dma16(in2_v, dst_a(i), (unsigned)a_ix0[i], so);
                    else
                        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(in2_rsrc, (unsigned)a_ix0[i], so, 0);
                }
            } else {
                const unsigned so = (unsigned)kload * (unsigned)ROW_BYTES;
#pragma unroll
                for (int i = 0; i < A_IT; i++) {
                    if constexpr (GLDS)
                        dma16(in_v, dst_a(i), (unsigned)a_iy0[i], so);
                    else
                        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (unsigned)a_iy0[i], so, 0);
                }
            }
            kload++;
            return;
        }
        const int dy = ky * a.dil, dx = kx * a.dil;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const unsigned coff = (unsigned)(cc * ROW_BYTES + st_chunk(st_row(i, A_IT)) * 16);
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = (unsigned)(iy * a.W + ix) * (unsigned)(a.Cin * ES) + coff;
            if constexpr (GLDS)
                dma16(in_v, dst_a(i), ok ? off : OOB, 0u);
            else
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? off : OOB, 0, 0);
        }
        // advance (ky,kx,cc) to the next K step, branch-free (all wave-uniform scalars)
        if constexpr (CMAJ) {
            kx += 1;
            const int w1 = kx == a.KW;
            kx = w1 ? 0 : kx;
            ky += w1;
            const int w2 = ky == a.KH;
            ky = w2 ? 0 : ky;
            cc += w2;
        } else {
            cc += 1;
            const int w1 = cc == cchunks;
            cc = w1 ? 0 : cc;
            kx += w1;
            const int w2 = kx == a.KW;
            kx = w2 ? 0 : kx;
            ky += w2;
        }
    };
    auto load_b = [&](int ks) {
        unsigned koff = (unsigned)ks * (unsigned)ROW_BYTES;
        if constexpr (CMAJ) {  // K step ks = (chunk, tap), the weight row is [tap][chunk]
            const int taps = a.KH * a.KW, c_ = ks / taps, t_ = ks - c_ * taps;
            koff = (unsigned)(t_ * cchunks + c_) * (unsigned)ROW_BYTES;
        }
#pragma unroll
        for (int i = 0; i < B_IT; i++) {  // K step in the scalar offset
            if constexpr (GLDS)
                dma16(wt_v, dst_b(i), b_off[i], koff);
            else
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc, b_off[i], koff, 0);
        }
    };
    auto load_step = [&](int ks) {
        load_a();
        load_b(ks);
    };
    const int c4 = tid & 7;  // register path: which 16-byte chunk of the 128-byte row this thread stages
    auto store_a = [&](int buf) {
        if constexpr (GLDS) return;  // the DMA wrote the image
        char* Ab = As + buf * BM * LR;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            if constexpr (FP8X)
                store_split_fp8(Ab + row * LR, c4, ra[i], a.a_scale);
            else if constexpr (SPLIT)
                store_split(Ab + row * LR + c4 * 8, ra[i], a.a_scale);
            else
                *reinterpret_cast<u32x4*>(Ab + row * LR + c4 * 16) = ra[i];
        }
    };
    auto store_b = [&](int buf) {
        if constexpr (GLDS) return;
        char* Bb = Bs + buf * BN * LR;
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            const int row = (tid >> 3) + i * (NT / 8);
            *reinterpret_cast<u32x4*>(Bb + row * LR + c4 * 16) = rb[i];  // SPLIT: split at load time
        }
    };
    auto store_step = [&](int buf) {
        store_a(buf);
        store_b(buf);
    };

    // LDS -> register fragments for one 32-byte k slice of buffer `buf`: lanes 0-31 take the
    // first 16 bytes (4 f32 / 8 f16 consecutive k), lanes 32-63 the second
    const int a_lds = (wm * TM * 32 + (lane & 31)) * LR + (GLDS ? 0 : (lane >> 5) * 16);
    const int b_lds = (wn * TN * 32 + (lane & 31)) * LR + (GLDS ? 0 : (lane >> 5) * 16);
    // DMA image: chunk q of a row sits at position q ^ swz(row); a fragment's rows are 32 apart, which leaves swz unchanged
    const int a_swz = lds_swz(lane & 31), b_swz = a_swz;
    auto read_frags = [&](int buf, int kk, float4 (&fa)[TM * NF], float4 (&fb)[TN * NF]) {
        const char* Ab = As + buf * BM * LR + a_lds + (GLDS ? (((2 * kk + (lane >> 5)) ^ a_swz) * 16) : kk * 32);
        const char* Bb = Bs + buf * BN * LR + b_lds + (GLDS ? (((2 * kk + (lane >> 5)) ^ b_swz) * 16) : kk * 32);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int p = 0; p < NF; p++) fa[i * NF + p] = *reinterpret_cast<const float4*>(Ab + i * 32 * LR + p * 64);
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int p = 0; p < NF; p++) fb[j * NF + p] = *reinterpret_cast<const float4*>(Bb + j * 32 * LR + p * 64);
    };

    float4 fa[TM * NF], fb[TN * NF], fa_n[TM * NF], fb_n[TN * NF];
    // FP8X: the e5m2 planes of the K step -- lane (row r, half h) takes the 32 bytes [64 + 32 h, 96 + 32 h) of its row:
    // h = 0 the hi / 2 plane (weights: lo * 2^11), h = 1 the lo * 2^10 plane (weights: hi), so that the 64-deep bf8
    // dot product is sum_k ah bl + al bh, scaled by 2^10 in both halves (undone by the instruction's 2^-10 block scale)
    float4 fa8[FP8X ? TM : 1][2], fb8[FP8X ? TN : 1][2];
    const int a_lds8 = (wm * TM * 32 + (lane & 31)) * LR + 64 + (lane >> 5) * 32;
    const int b_lds8 = (wn * TN * 32 + (lane & 31)) * LR + 64 + (lane >> 5) * 32;
    auto read_frags8 = [&](int buf) {
        if constexpr (FP8X) {
            const char* Ab = As + buf * BM * LR + a_lds8;
            const char* Bb = Bs + buf * BN * LR + b_lds8;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                fa8[i][0] = *reinterpret_cast<const float4*>(Ab + i * 32 * LR);
                fa8[i][1] = *reinterpret_cast<const float4*>(Ab + i * 32 * LR + 16);
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                fb8[j][0] = *reinterpret_cast<const float4*>(Bb + j * 32 * LR);
                fb8[j][1] = *reinterpret_cast<const float4*>(Bb + j * 32 * LR + 16);
            }
        }
    };

    // the MFMAs of one slice.  D rows = output channels, D cols = pixels (operands swapped on purpose)
    auto mma_slice = [&](const int kk) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                if constexpr (FP8X) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa[i]), acc[i][j], 0, 0, 0);
                    if (kk == 0) {
                        struct F2 { float4 lo, hi; };
                        const i32x8 a8 = __builtin_bit_cast(i32x8, (F2{fa8[i][0], fa8[i][1]}));
                        const i32x8 b8 = __builtin_bit_cast(i32x8, (F2{fb8[j][0], fb8[j][1]}));
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[i][j], kFp8Fmt, kFp8Fmt, 0, kFp8CrossScaleA, 0, 127);
                    }
                } else if constexpr (I8) {
                    // activations are u8, the MFMA is signed: x ^ 0x80 = x - 128 as s8 (the -128 * sum w is in q_bias)
                    const i32x4q ax = __builtin_bit_cast(i32x4q, fa[i]) ^ (int)0x80808080;
                    acc[i][j] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4q, fb[j]), ax,
                                                                                                  __builtin_bit_cast(i32x16q, acc[i][j]), 0, 0, 0));
                } else if constexpr (F32) {
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
            if (kk == 0) read_frags8(buf);
            mma_slice(kk);
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

    if constexpr (GLDS) {
        ld_buf = 0;
        load_step(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        load_step(0);
        store_step(0);
        if (ksteps > 1) load_step(1);
        __syncthreads();
    }
#ifdef KTRACE
    const unsigned long long kt_loop = __builtin_amdgcn_s_memtime();
#endif
    // epilogue geometry: a wave's 32-pixel row block is written out with LPR lanes (CPL channels each)
    // on every pixel row, RPI rows per instruction.  CPL = 4 (16 bytes of f32 read from the staging row); a u8 output takes
    // 16 channels per lane: its 16-byte stores and residual loads are a quarter of the instructions of the 4-byte form
    // (which ran the quantised 1x1 expansions at 2 TB/s, instruction-bound in the epilogue)
    constexpr bool Q8 = I8 && std::is_same<OutT, unsigned char>::value;
    // f16 -> f16: 8 channels per lane (two float4 of the staging row, one 16-byte store / residual load): half the epilogue's
    // iterations of the 4-channel form, whose 8-byte stores were 12 % of the 4K layer3 conv2's workgroup life (scripts/ktrace.py)
    constexpr bool H8 = std::is_same<T, _Float16>::value && std::is_same<OutT, _Float16>::value;
    constexpr int CPL = Q8 ? 16 : (H8 ? 8 : 4);
    constexpr int LPR = TN * 32 / CPL;
    constexpr int RPI = 64 / LPR;
    const int e_row = lane / LPR, e_col = lane % LPR;
    const int e_n = n0 + wn * TN * 32 + e_col * CPL;
    const T* res = static_cast<const T*>(a.res);
    using ResV = typename std::conditional<std::is_same<T, float>::value, float4,
                                           typename std::conditional<Q8, u32x4, typename std::conditional<I8, unsigned,
                                           typename std::conditional<H8, f16x8, f16x4>::type>::type>::type>::type;
    ResV rres[RESPF ? TM : 1][RESPF ? 32 / RPI : 1];
    if constexpr (RESPF) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int it = 0; it < 32 / RPI; it++) {
                const int m = m0 + wm * TM * 32 + i * 32 + it * RPI + e_row;
                ResV r = {};
                if (m < M && e_n < a.Cout) r = *reinterpret_cast<const ResV*>(res + (size_t)m * a.Cout + e_n);
                rres[i][it] = r;
            }
    }

    if constexpr (NBUF == 3) {
        // two LDS images, fragments read slice by slice into ONE register set (the big-tile form: the
        // 8 waves of a 256x256 tile keep 128 accumulators each; the second wave on the SIMD covers the
        // ds_read latency).  Stores of K step ks+1 go to the other image during slice 0; one barrier
        // ends the step.
        for (int ks = 0; ks < ksteps; ks++) {
            const int buf = ks & 1;
#pragma unroll
            for (int kk = 0; kk < NSL; kk++) {
                read_frags(buf, kk, fa, fb);
                if (kk == 0) read_frags8(buf);
                mma_slice(kk);
                // (staging the activations at slice 0 and the weights at slice 1 instead measured the same: these are
                //  ordinary loads and stores, the compiler spreads them either way)
                if (kk == 0 && ks + 1 < ksteps) {
                    store_step(buf ^ 1);
                    if (ks + 2 < ksteps) load_step(ks + 2);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else if constexpr (NBUF >= 4) {
        // LDS-DMA form: the DMA of K step ks + 1 goes into the other image and lands while this step's 32 MFMAs per
        // wave run; every wave waits for its own pieces (vmcnt) and its fragment reads (lgkmcnt) before the barrier that
        // ends the step -- after it the other image is complete and this one may be overwritten.
        // WHERE the pieces are issued is the NBUF 4 / 5 difference:
        //   4: all of them before slice 0.  Most bytes in flight for the longest time: best for the HBM-bound 1x1 convs.
        //   5: the activation pieces after the fragment reads of slice 0, the weight pieces after those of slice 1.  An
        //      LDS-DMA instruction costs its wave 60-180 issue cycles (MI355X_MICROARCH.md); issued up front, the eight
        //      of them keep the wave -- and, the waves of a workgroup running in step, the whole SIMD -- off the matrix
        //      pipe at the top of every K step.  Behind a slice's ds_reads they issue in the shadow of MFMAs that already
        //      have their operands.  4K FCN-ResNet101: classifier.0 1241 -> 1338 TFLOP/s, layer4 conv2 1244 -> 1321,
        //      layer3 conv2 1053-1092 -> 1150; the HBM-bound conv1 (1024 -> 256) gets SLOWER (0.084-0.089 -> 0.092 ms),
        //      so both forms are configurations and the tuner picks per shape.  (Other placements measured: one slice
        //      later, or after each slice's MFMAs instead of before: +2-4 % only; the weight pieces after slice 0's MFMAs,
        //      or s_setprio(1) around the MFMAs of a slice: 4 % slower than this form; 3 + 3 + 2 or 2 + 4 + 2 pieces over
        //      slices 0-2: the same within noise; all eight after the reads of slice 0: as slow as up front.)
        // (Reading the next slice's fragments ahead of this slice's MFMAs -- the register path's prefetch -- was
        // measured here and is slower: 1187 vs 1225 TFLOP/s on the 4K classifier.0; the second wave on the SIMD already
        // covers the ds_read latency and the extra register set costs more than it hides.)
        constexpr bool SPREAD = NBUF == 5;
        for (int ks = 0; ks < ksteps; ks++) {
            const int buf = ks & 1;
            const bool more = ks + 1 < ksteps;
            ld_buf = buf ^ 1;
            if (!SPREAD && more) load_step(ks + 1);
#pragma unroll
            for (int kk = 0; kk < NSL; kk++) {
                read_frags(buf, kk, fa, fb);
                if (SPREAD && more && kk == 0) load_a();
                if (SPREAD && more && kk == 1) load_b(ks + 1);
                mma_slice(kk);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else if constexpr (NBUF == 2) {
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
                if (kk == 0) read_frags8(0);
                mma_slice(kk);
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
    // output channels.  Stored straight from that layout a wave instruction would touch 32 pixels
    // x 32 bytes (32 cache lines); instead each wave passes its 32-pixel row block through its
    // own slice of the (now idle) operand LDS and stores / loads the residual with 16 lanes on the
    // 256 contiguous bytes of a pixel: 8 full lines per instruction, a quarter of the TA work.
    OutT* out = reinterpret_cast<OutT*>(static_cast<char*>(a.out) + (size_t)bidx * a.out_bs);
    // split modes: one accumulator scale per launch, or one per batched problem (the Winograd planes carry their own weight scale)
    const float acc_scale = a.acc_scale_b ? a.acc_scale_b[bidx] : a.acc_scale;
    const bool has_bias = a.bias != nullptr;
    const bool vec_ok = I8 || (a.Cout & (CPL - 1)) == 0;  // (the quantised epilogue guards its 4 channels one by one: 21-class logits too)
#ifdef KTRACE
    const unsigned long long kt_epi = __builtin_amdgcn_s_memtime();
    struct KtEnd {
        unsigned long long t0, t1, t2; int on, slot;
        __device__ ~KtEnd() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            if (on) { g_ktrace[slot * 4 + 0] = t1 - t0; g_ktrace[slot * 4 + 1] = t2 - t1; g_ktrace[slot * 4 + 2] = t3 - t2; }
        }
    } kt_end{kt_start, kt_loop, kt_epi, (a.Cin == KT_CIN && a.Cout == KT_COUT && blockIdx.x < 8 && lane == 0) ? 1 : 0, (int)(blockIdx.x * 8 + wave)};
#endif
    if (vec_ok) {
        constexpr int ROWB = TN * 128 + 16;  // staged row: TN*32 f32 + pad (conflict-free b128 writes and reads)
        static_assert(NT / 64 * 32 * ROWB <= lds_bytes(BM, BN, WM, WN, NBUF), "epilogue staging exceeds the LDS allocation");
        __syncthreads();  // every wave is done with the operand tiles
        char* stage = smem + wave * 32 * ROWB;
        const int rrow = e_row, rcol = e_col;
        const int n = e_n;
        const bool n_ok = n < a.Cout;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bv2 = bv;
        if (has_bias && n_ok) {
            bv = *reinterpret_cast<const float4*>(a.bias + n);
            if constexpr (H8) bv2 = *reinterpret_cast<const float4*>(a.bias + n + 4);
        }
        int qb[CPL];    // I8: folded bias and requantisation multiplier of this lane's channels
        float qm[CPL];
#pragma unroll
        for (int t = 0; t < CPL; t++) {
            qb[t] = 0;
            qm[t] = 0.f;
        }
        if constexpr (Q8) {
            // (a u8 output has Cout % 16 == 0 and n is a multiple of 16: the lane's 16 channels exist together -- eight 16-byte loads
            //  instead of 32 guarded dword loads at the head of every epilogue)
            if (n_ok) {
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) {
                    const qi4 b4 = *reinterpret_cast<const qi4*>(a.q_bias + n + 4 * t4);
                    const qf4 m4 = *reinterpret_cast<const qf4*>(a.q_mult + n + 4 * t4);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        qb[4 * t4 + t] = b4[t];
                        qm[4 * t4 + t] = m4[t];
                    }
                }
            }
        } else if constexpr (I8) {
#pragma unroll
            for (int t = 0; t < CPL; t++)
                if (n + t < a.Cout) {
                    qb[t] = a.q_bias[n + t];
                    qm[t] = a.q_mult[n + t];
                }
        }
        const float q_yzpf = (float)a.q_yzp, q_lo = -q_yzpf, q_hi = 255.f - q_yzpf;
        const QEpi qe = {q_yzpf, q_lo, q_hi, a.q_ra, a.q_rb, (float)a.q_bzp, (float)a.q_czp};
        float vmax = 0.f;  // SPLIT: largest |output| of this lane (range monitor)
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int jj = 0; jj < TN; jj++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float4 v = make_float4(acc[i][jj][4 * g + 0], acc[i][jj][4 * g + 1], acc[i][jj][4 * g + 2], acc[i][jj][4 * g + 3]);
                    if constexpr (SPLIT) {
                        v.x *= acc_scale; v.y *= acc_scale; v.z *= acc_scale; v.w *= acc_scale;
                    }
                    *reinterpret_cast<float4*>(stage + (lane & 31) * ROWB + (jj * 32 + 8 * g + 4 * (lane >> 5)) * 4) = v;
                }
            // the slice is private to this wave and LDS serves a wave's operations in order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int mb = m0 + wm * TM * 32 + i * 32;
            // Tiles too big for RESPF (256x256: 128 accumulators per lane): all residual loads of this 32-row block are
            // issued here, before the first of them is needed -- inside the loop below each one would wait out its full
            // latency behind the branch, with one workgroup per CU and nothing else to run (what made the first version
            // of conv1x1_areg.hip twice as slow).  The accumulators of block i are already in LDS: their registers are free.
            ResV rlate[RESPF ? 1 : 32 / RPI];
            if constexpr (!RESPF) {
                if (res) {
#pragma unroll
                    for (int it = 0; it < 32 / RPI; it++) {
                        const int m = mb + it * RPI + rrow;
                        ResV r = {};
                        if (m < M && n_ok) r = *reinterpret_cast<const ResV*>(res + (size_t)m * a.Cout + n);
                        rlate[it] = r;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 32 / RPI; it++) {
                const int row = it * RPI + rrow;
                const int m = mb + row;
                float4 v = *reinterpret_cast<const float4*>(stage + row * ROWB + rcol * (CPL * 4));
                if constexpr (Q8) {
                    // (the launcher admits a u8 output only with Cout % 16 == 0: channel padding of the quantised tensors)
                    i32x4q ai[4];
                    ai[0] = __builtin_bit_cast(i32x4q, v);
#pragma unroll
                    for (int t4 = 1; t4 < 4; t4++) ai[t4] = *reinterpret_cast<const i32x4q*>(stage + row * ROWB + rcol * 64 + t4 * 16);
                    if (m < M && n_ok) {
                        u32x4 rv = {0u, 0u, 0u, 0u};
                        const bool has_res = RESPF || res;
                        if (has_res) {
                            if constexpr (RESPF)
                                rv = rres[i][it];
                            else
                                rv = rlate[it];
                        }
                        u32x4 pk;
#pragma unroll
                        for (int t4 = 0; t4 < 4; t4++) {  // (qepilogue.h: four outputs per call, packed multiplies / adds)
                            const qi4 a4 = {ai[t4][0] + qb[4 * t4], ai[t4][1] + qb[4 * t4 + 1], ai[t4][2] + qb[4 * t4 + 2], ai[t4][3] + qb[4 * t4 + 3]};
                            const qf4 m4 = {qm[4 * t4], qm[4 * t4 + 1], qm[4 * t4 + 2], qm[4 * t4 + 3]};
                            pk[t4] = has_res ? q_word<true>(a4, m4, rv[t4], qe) : q_word<false>(a4, m4, 0u, qe);
                        }
                        *reinterpret_cast<u32x4*>(out + (size_t)m * a.Cout + n) = pk;
                    }
                } else if constexpr (I8) {
                    if (m < M && n_ok) {
                        const size_t o = (size_t)m * a.Cout + n;
                        const i32x4q ai = __builtin_bit_cast(i32x4q, v);
                        float q[4];  // y - y_zp
#pragma unroll
                        for (int t = 0; t < 4; t++) q[t] = q_requant_c(ai[t] + qb[t], qm[t], q_lo, q_hi);
                        {  // f32 output: the dequantised logits (a logit conv has no residual: launch_cfg)
#pragma clang fp contract(off)
                            float4 d;
                            // (+ q_dq_off, 0.0f for DequantizeLinear: the centred value may be -0.0 where the operator's f32(q - zp) is +0.0)
                            d.x = (q[0] + a.q_dq_off) * a.q_dq; d.y = (q[1] + a.q_dq_off) * a.q_dq;
                            d.z = (q[2] + a.q_dq_off) * a.q_dq; d.w = (q[3] + a.q_dq_off) * a.q_dq;
                            if ((a.Cout & 3) == 0) {
                                *reinterpret_cast<float4*>(out + o) = d;
                            } else {  // (21 logits per pixel: rows are not 16-byte aligned, the last group is partial)
                                const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                                for (int t = 0; t < 4; t++)
                                    if (n + t < a.Cout) out[o + t] = dd[t];
                            }
                        }
                    }
                } else if constexpr (H8) {
                    if (m < M && n_ok) {  // the same steps per value as the 4-channel form below: + bias, + residual, ReLU, round to f16
                        const float4 v2 = *reinterpret_cast<const float4*>(stage + row * ROWB + rcol * (CPL * 4) + 16);
                        float x[8] = {v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w, v2.x + bv2.x, v2.y + bv2.y, v2.z + bv2.z, v2.w + bv2.w};
                        if (RESPF || res) {
                            ResV rv;
                            if constexpr (RESPF)
                                rv = rres[i][it];
                            else
                                rv = rlate[it];
#pragma unroll
                            for (int t = 0; t < 8; t++) x[t] += (float)rv[t];
                        }
                        if (a.relu) {
#pragma unroll
                            for (int t = 0; t < 8; t++) x[t] = fmaxf(x[t], 0.f);
                        }
                        const f16x8 hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3],
                                          (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
                        *reinterpret_cast<f16x8*>(out + (size_t)m * a.Cout + n) = hv;
                    }
                } else if (m < M && n_ok) {
                    const size_t o = (size_t)m * a.Cout + n;
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (RESPF || res) {
                        ResV rv;
                        if constexpr (RESPF)
                            rv = rres[i][it];
                        else
                            rv = rlate[it];
                        if constexpr (std::is_same<T, float>::value) {
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        } else {
                            v.x += (float)rv[0]; v.y += (float)rv[1]; v.z += (float)rv[2]; v.w += (float)rv[3];
                        }
                    }
                    if (a.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if constexpr (SPLIT) vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                    if constexpr (std::is_same<OutT, float>::value) {
                        *reinterpret_cast<float4*>(out + o) = v;
                    } else {
                        f16x4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                        *reinterpret_cast<f16x4*>(out + o) = hv;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (SPLIT) {
            if (a.amax) {  // non-negative floats order like their bit patterns; the atomic is skipped unless this wave
                           // raises the maximum (tens of thousands of same-address atomics cost 0.17 ms per launch)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
                if (lane == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(a.amax)) atomicMax(a.amax, __float_as_uint(vmax));
            }
        }
        return;
    }
    // Cout not a multiple of 4 (the 21-class logits): element-wise from the accumulator layout
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
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (n + t >= a.Cout) break;
                    if constexpr (I8) {
                        // (never reached: the quantised epilogue above handles every Cout)
                    } else {
                        float x = acc[i][j][4 * g + t];
                        if constexpr (SPLIT) x *= acc_scale;
                        x += has_bias ? a.bias[n + t] : 0.f;
                        if (res) x += (float)res[o + t];
                        if (a.relu) x = fmaxf(x, 0.f);
                        out[o + t] = (OutT)x;
                    }
                }
            }
        }
    }
}

template <typename T, typename OutT, bool SPLIT, bool FP8X, bool G1, bool RESPF, bool DUAL, int BM, int BN, int WM, int WN, int NBUF>
static hipError_t launch_cfg_g(const ConvArgs& a, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + BM - 1) / BM;
    const int ntiles = (a.Cout + BN - 1) / BN;
    const size_t lds = (size_t)lds_bytes(BM, BN, WM, WN, NBUF);
    auto k = conv_igemm_kernel<T, OutT, BM, BN, WM, WN, NBUF, SPLIT, G1, RESPF, DUAL, FP8X>;
    // > 64 KB of dynamic LDS needs the attribute once per kernel AND per device (a process may hold contexts on
    // several GPUs, each driven from its own thread -- infur_group): the flags are atomics, and two threads that both
    // find a flag clear simply both make the (idempotent) call
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(mtiles * ntiles * (a.batch > 1 ? a.batch : 1)), dim3(WM * WN * 64), lds, s, a, mtiles, ntiles);
    return hipGetLastError();
}

template <typename T, typename OutT, bool SPLIT, bool FP8X, int BM, int BN, int WM, int WN, int NBUF = 2>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    const bool g1 = a.KH == 1 && a.KW == 1 && a.pad == 0;
    // the f16 -> f32 (and u8 -> f32) kernel only ever runs the classifier; an i8 convolution stores u8 activations
    constexpr bool kI8 = std::is_same<T, signed char>::value;
    constexpr bool kSameType = std::is_same<T, OutT>::value || (kI8 && std::is_same<OutT, unsigned char>::value);
    if (kI8 && a.in2) return hipErrorInvalidValue;  // (the two convolutions of a quantised block requantise separately)
    if (kI8 && std::is_same<OutT, unsigned char>::value && (a.Cout & 15)) return hipErrorInvalidValue;  // 16-byte epilogue stores
    if (kI8 && std::is_same<OutT, float>::value && a.res) return hipErrorInvalidValue;  // (dequantised logits: no residual sum)
    if constexpr (kSameType) {
        if (a.in2) {
            if (!g1 || a.stride != 1 || a.res) return hipErrorInvalidValue;
            return launch_cfg_g<T, OutT, SPLIT, FP8X, true, false, true, BM, BN, WM, WN, NBUF>(a, s);
        }
        // residual prefetch: only where a lane holds <= 64 accumulators (room for 64 more registers) -- a
        // residual only ever enters a 1x1 conv, so this is a G1 form
        constexpr bool kCanPf = (BM / WM) * (BN / WN) <= 64 * 64;
        if constexpr (kCanPf) {
            // (the prefetch reads a lane's whole channel group: 4 channels, 8 in the f16 -> f16 epilogue)
            constexpr int kGroup = (std::is_same<T, _Float16>::value && std::is_same<OutT, _Float16>::value) ? 8 : 4;
            if (a.res && g1 && (a.Cout & (kGroup - 1)) == 0) return launch_cfg_g<T, OutT, SPLIT, FP8X, true, true, false, BM, BN, WM, WN, NBUF>(a, s);
        }
    } else if (a.in2) {
        return hipErrorInvalidValue;
    }
    // 1x1 without padding: the plain-GEMM addressing form (no vector address arithmetic in the K loop)
    if (g1) return launch_cfg_g<T, OutT, SPLIT, FP8X, true, false, false, BM, BN, WM, WN, NBUF>(a, s);
    return launch_cfg_g<T, OutT, SPLIT, FP8X, false, false, false, BM, BN, WM, WN, NBUF>(a, s);
}

template <typename T, typename OutT, bool SPLIT = false, bool FP8X = false>
static hipError_t launch_t(const ConvArgs& a, int cfg, hipStream_t s) {
    constexpr size_t ES = sizeof(T);
    if (a.Cin % (int)(ROW_BYTES / ES) != 0 || (a.in2 && a.Cin2 % (int)(ROW_BYTES / ES) != 0)) return hipErrorInvalidValue;
    // 32-bit buffer offsets with 0x80000000 as the out-of-range marker
    if ((size_t)a.H * a.W * a.Cin * ES >= 0x80000000ull || (size_t)a.Cout * a.KH * a.KW * a.Cin * ES >= 0x80000000ull)
        return hipErrorInvalidValue;
    if (cfg < 0) cfg = conv_igemm_default_config(a);
    switch (cfg) {
        case 0: return launch_cfg<T, OutT, SPLIT, FP8X, 128, 128, 2, 2>(a, s);
        case 1: return launch_cfg<T, OutT, SPLIT, FP8X, 64, 128, 2, 2>(a, s);
        case 2: return launch_cfg<T, OutT, SPLIT, FP8X, 128, 64, 2, 2>(a, s);
        case 3: return launch_cfg<T, OutT, SPLIT, FP8X, 64, 64, 2, 2>(a, s);
        case 4: return launch_cfg<T, OutT, SPLIT, FP8X, 256, 32, 4, 1>(a, s);
        case 5: return launch_cfg<T, OutT, SPLIT, FP8X, 128, 256, 2, 4>(a, s);
        case 6: return launch_cfg<T, OutT, SPLIT, FP8X, 256, 128, 4, 2>(a, s);
        case 7: return launch_cfg<T, OutT, SPLIT, FP8X, 128, 128, 2, 2, 1>(a, s);
        case 8: return launch_cfg<T, OutT, SPLIT, FP8X, 128, 64, 2, 2, 1>(a, s);
        case 9: return launch_cfg<T, OutT, SPLIT, FP8X, 64, 128, 2, 2, 1>(a, s);
        case 10: return launch_cfg<T, OutT, SPLIT, FP8X, 64, 64, 2, 2, 1>(a, s);
        case 11: return launch_cfg<T, OutT, SPLIT, FP8X, 256, 256, 2, 4, 3>(a, s);  // 8 waves of 128x64, one fragment set
        case 12: return launch_cfg<T, OutT, SPLIT, FP8X, 256, 128, 4, 2, 3>(a, s);  // 8 waves of 64x64, one fragment set
        case 13:
        case 14:
            if constexpr ((std::is_same<T, _Float16>::value || std::is_same<T, signed char>::value) && !SPLIT) {  // LDS-DMA staging
                if (cfg == 13) return launch_cfg<T, OutT, SPLIT, FP8X, 256, 256, 2, 4, 4>(a, s);
                return launch_cfg<T, OutT, SPLIT, FP8X, 256, 128, 4, 2, 4>(a, s);
            }
            return hipErrorInvalidValue;
        case 16:
        case 17:
            if constexpr ((std::is_same<T, _Float16>::value || std::is_same<T, signed char>::value) && !SPLIT) {
                if (cfg == 16) return launch_cfg<T, OutT, SPLIT, FP8X, 256, 256, 2, 4, 5>(a, s);
                return launch_cfg<T, OutT, SPLIT, FP8X, 256, 128, 4, 2, 5>(a, s);
            }
            return hipErrorInvalidValue;
        case 15:
            if constexpr (std::is_same<T, _Float16>::value && std::is_same<OutT, _Float16>::value && !SPLIT) {
                if (conv1x1_areg_valid(a, 1, 0)) return launch_conv1x1_areg(a, s);
            }
            if constexpr (std::is_same<T, signed char>::value && std::is_same<OutT, unsigned char>::value) {
                if (conv1x1_q8_valid(a, 4, 0)) return launch_conv1x1_q8(a, 1, s);
            }
            return hipErrorInvalidValue;
        case 19:
        case 20:
            if constexpr (std::is_same<T, _Float16>::value && std::is_same<OutT, _Float16>::value && !SPLIT) return launch_conv3x3_halo(a, cfg == 19 ? 128 : 256, s);
            if constexpr (std::is_same<T, signed char>::value && std::is_same<OutT, unsigned char>::value) return launch_conv3x3_halo_q(a, cfg == 19 ? 128 : 256, s);
            return hipErrorInvalidValue;
        case 21:
            if constexpr (std::is_same<T, _Float16>::value && std::is_same<OutT, _Float16>::value && !SPLIT) return launch_conv3x3_halo4(a, s);
            return hipErrorInvalidValue;
        case 18:
            if constexpr (std::is_same<T, signed char>::value && std::is_same<OutT, unsigned char>::value) {
                if (conv1x1_q8_valid(a, 4, 0) && conv1x1_q8_nsplit(a) > 1) return launch_conv1x1_q8(a, conv1x1_q8_nsplit(a), s);
            }
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}


}  // namespace infur
