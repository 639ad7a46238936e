// conv1x1_b2b.hip -- two 1x1 convolutions back to back in ONE kernel, f16 operands:
//     y   = ReLU(W3 * t2 + b3 + x)        a bottleneck's conv3 + residual      (C2 -> 4*C2 channels, written once)
//     t1' = ReLU(W1' * y + b1')           the NEXT bottleneck's conv1          (4*C2 -> C2 channels)
// for the same pixels, so y -- the widest tensor of the stage (265 MB per launch at 4K in FCN-ResNet101's layer3) -- is
// written once as the next block's residual and never read back by conv1'.  Replaces two of the Conv/Add/Relu node
// groups ONNX Runtime executes inside `session.run` (infur/src/predict_onnx.rs:138).
//
// Why: in the f16-rate modes these 1x1 GEMMs are HBM-bound (LAB_NOTES.md 3.3): conv3 reads t2 + x and writes y, conv1' reads
// y again and writes t1' -- 927 MB of traffic per pair at 4K for 662 MB of compulsory bytes once the re-read is gone.
//
// Shape of the kernel: a workgroup owns 256 pixels, each of its 8 waves 32 of them, privately: the wave keeps its 32 x C2
// activation fragment (<= 64 VGPRs) and its 32 x C2 accumulators of the SECOND GEMM (<= 128 VGPRs) in registers for the
// whole kernel and walks the 4*C2 channels of y in steps of 32:
//   GEMM 1   acc1[32 px x 32 ch] = t2 fragment x W3 slice              (C2 / 16 MFMAs, weight fragments from LDS)
//   epilogue + bias + residual, ReLU, f16 -- in the MFMA's C/D layout; these 16 values per lane ARE the activation
//            operand of GEMM 2's next two MFMA k-slices (no LDS round trip: the rows of the W3 slice are staged in a
//            permuted order so that a lane's accumulator registers hold 8 CONSECUTIVE channels per k-slice), and they go
//            out to y through the wave's own LDS slot (4 lanes x 16 B = the 64 contiguous bytes of a pixel)
//   GEMM 2   acc2[32 px x C2] += y slice x W1' slice                   (C2 / 16 MFMAs)
// Everything streamed -- W3 slice, W1' slice (both from L2: 1 MB of weights per workgroup) and the residual slice (HBM)
// -- arrives by LDS-DMA two steps ahead in rings of three slots: no staging registers, one workgroup barrier per step.
//
// Same arithmetic as the unfused pair (conv1x1_areg / conv_igemm_kernel in the f16 mode): every output element sums its
// k in ascending 16-wide MFMA slices with the same operand slots, then (+ bias) + residual, ReLU, round to f16 -- y and
// t1' are BIT-IDENTICAL to the two-launch form (tests/test_gpu_b2b.py), so fusing is a speed decision only.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace infur {

typedef float f32x16b __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8b __attribute__((ext_vector_type(8)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_b;

// Instrumentation build (-DB2B_TRACE, scripts/b2b_trace.py): workgroup B2B_TRACE_WG of every launch accumulates, per wave,
// the shader cycles (s_memtime) of each phase of a step.  Not compiled into the product library.
#ifdef B2B_TRACE
#ifndef B2B_TRACE_WG
#define B2B_TRACE_WG 100
#endif
__device__ unsigned long long g_b2b_trace[8 * 8];
#define BB_T(k) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); tr[k] += now__ - tlast; tlast = now__; } while (0)
#else
#define BB_T(k) do { } while (0)
#endif

namespace {

constexpr int BB_BM = 256;   // pixels per workgroup
constexpr int BB_NT = 32;    // channels of y per step
constexpr int BB_NSLOT = 3;  // ring depth: step t computes while t + 1 and t + 2 are in flight
#ifndef B2B_PF4
#define B2B_PF4 8
#endif

// LDS: three rings, each contiguous (so that one lane-constant VGPR per operand plus an immediate reaches every slot),
// then the two bias tables.
template <int KS, int BM = BB_BM>
struct B2bLds {
    static constexpr int C2 = 64 * KS, C4 = 4 * C2, N2 = C2;
    static constexpr int W3_SLICE = BB_NT * C2 * 2;  // 32 rows x C2 f16: KS sub-images [32 rows][128 B]
    static constexpr int W1_SLICE = N2 * BB_NT * 2;  // N2 rows x 64 B
    static constexpr int RSLOT = BM * 64;            // [BM px][64 B]: residual in, y out (in place); a wave owns its rows
    static constexpr int OFF_W3 = 0;
    static constexpr int OFF_W1 = OFF_W3 + BB_NSLOT * W3_SLICE;
    static constexpr int OFF_R = OFF_W1 + BB_NSLOT * W1_SLICE;
    static constexpr int OFF_B3 = OFF_R + BB_NSLOT * RSLOT;
    static constexpr int OFF_B1 = OFF_B3 + C4 * 4;
    static constexpr int TOTAL = OFF_B1 + N2 * 4;
    static constexpr int FINAL = BM * N2 * 2;        // staging of t1' (over the idle rings)
    static_assert(FINAL <= OFF_B3, "the t1' staging must not reach the bias tables");
    static_assert(TOTAL <= 160 * 1024, "LDS");
    static_assert(BB_NSLOT * W3_SLICE <= 65536 && BB_NSLOT * W1_SLICE <= 65536 && BB_NSLOT * RSLOT <= 65536, "16-bit DS offsets");
};

// image row r of a 32-row weight block holds output channel pi(r): bits 2 and 3 of r swapped.  In the MFMA's C/D layout
// lane (pixel, hh) register 4g + e is row 8g + 4hh + e; with the permuted rows that is channel 16(g >> 1) + 8hh + 4(g & 1) + e,
// i.e. registers 8s .. 8s+7 are the 8 consecutive channels 16s + 8hh .. + 7: exactly lane-half hh's k slots of MFMA k-slice s.
__host__ __device__ constexpr int b2b_pi(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

__device__ __forceinline__ void bb_dma16(const u32x4b rsrc, const unsigned lds, const unsigned voff, const unsigned soff) {
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

__device__ __forceinline__ u32x4b bb_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    u32x4b r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

// The W3 copy this kernel streams (one-off at model load).  Every workgroup walks the channel blocks of y in the same order
// at about the same time, so all 256 CUs ask for the same 32 rows of W3 -- 16 KB contiguous in the plain [4*C2][C2] layout,
// i.e. a handful of L2 channels -- at once: measured 128k vs 116k cycles per workgroup (and 210k vs 167k in the one-wave-per-
// SIMD form) against a layout that puts the 32 rows of a step 16 KB apart.  dst[(r * NSTEP + t) * C2 + k] = src[(32 t + pi(r)) * C2 + k]:
// the row permutation of the C/D-layout trick (b2b_pi) is folded in.  (Walking the blocks from a different start per
// workgroup removes the hot spot as well, but changes the k order of GEMM 2 -- no longer bit-identical to the unfused conv1.)
__global__ void b2b_pack_w3_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, const int C2) {
    const int C4 = 4 * C2, nstep = C4 / BB_NT;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk of dst
    const size_t chunks_per_row = (size_t)C2 / 8;
    if (i >= (size_t)C4 * chunks_per_row) return;
    const int drow = (int)(i / chunks_per_row), ch = (int)(i % chunks_per_row);
    const int r = drow / nstep, t = drow % nstep;
    const int srow = BB_NT * t + b2b_pi(r);
    reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[(size_t)srow * chunks_per_row + ch];
}

// KS = C2 / 64 (2: layer2, 4: layer3 of a ResNet-50/101).  A wave owns RB blocks of 32 pixels; a workgroup NW x RB x 32 pixels.
//   <KS, 1, 8>: two waves per SIMD, <= 256 registers each (A fragment 16 KS, GEMM-2 accumulators 32 KS, one weight fragment
//               in flight: the two waves of a SIMD cover each other's LDS latency)
//   <KS, 2, 4>: one wave per SIMD with the whole 512-register file: every weight fragment feeds two MFMAs (half the LDS
//               reads) and there is room to keep several fragments in flight
//   <KS, 1, 4>: 128 pixels per workgroup (round 4): at 1080p the stride-8 map has M = 32,400 pixels = 127 workgroups of 256 --
//               half the chip idle for the whole launch; 254 workgroups of 128 put one on every CU.  One wave per SIMD and the
//               full weight stream per 128 pixels make the workgroup less efficient (nothing covers a wave held at a DMA issue),
//               but it does half the work: picked when the 256-pixel form would leave more than a third of the CUs empty
template <int KS, int RB, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) conv1x1_b2b_kernel(const B2bArgs a, const int mtiles, const int dma_phase) {
    constexpr int BM = NW * RB * 32;
    using L = B2bLds<KS, BM>;
    constexpr int C2 = L::C2, C4 = L::C4, N2 = L::N2, NJ = N2 / 32, NSTEP = C4 / BB_NT, NSL = KS * 4;
    constexpr int NT = NW * 64;
    constexpr int NPW = KS * 4 / NW;       // DMA pieces (1 KB) per wave and step for each weight slice
    constexpr int NPR = 2 * RB;            // ... and residual pieces (16 pixels x 64 B each)
    constexpr int PP = 2 * NPW + NPR;
    // weight fragments in flight per wave (one wave per SIMD has the whole register file: nothing else covers its ds_read latency)
    constexpr int PF = RB == 1 ? (NW == 4 ? B2B_PF4 : 2) : 6;
    static_assert(KS == 2 || KS == 4, "C2 = 128 or 256");
    static_assert((BM == 256 || BM == 128) && NPW >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: every per-wave address below stays in SGPRs
    const int r = lane & 31, hh = lane >> 5;
    const int M = a.M;
    int tile;
    {  // XCD-aware order: block b runs on XCD b % 8; every XCD gets a contiguous run of pixel tiles
        const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
        const int q = mtiles >> 3, rem = mtiles & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    }
    const int m0w = tile * BM + wave * (32 * RB);  // first pixel of this wave

    // ---- the wave's activation fragments, loaded once: pixel m0w + 32 ib + r, all C2 channels (rows >= M lie beyond
    //      num_records: zeros) ----
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (unsigned)((size_t)M * C2 * 2), 0x00020000);
    h16x8b areg[RB][NSL];
#pragma unroll
    for (int ib = 0; ib < RB; ib++) {
        const unsigned abase = (unsigned)(m0w + 32 * ib + r) * (unsigned)(C2 * 2) + (unsigned)hh * 16u;
#pragma unroll
        for (int sl = 0; sl < NSL; sl++)
            areg[ib][sl] = __builtin_bit_cast(h16x8b, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, abase, (unsigned)(sl * 32), 0));
    }
    // ---- bias tables into LDS (read back as broadcasts in the epilogues) ----
    for (int i = tid; i < C4 / 4; i += NT) reinterpret_cast<float4*>(smem + L::OFF_B3)[i] = reinterpret_cast<const float4*>(a.b3)[i];
    for (int i = tid; i < N2 / 4; i += NT) reinterpret_cast<float4*>(smem + L::OFF_B1)[i] = reinterpret_cast<const float4*>(a.b1)[i];

    // ---- the three DMA streams ----
    const u32x4b w3_v = bb_rsrc(a.w3, (unsigned)(C4 * C2 * 2));
    const u32x4b w1_v = bb_rsrc(a.w1, (unsigned)(N2 * C4 * 2));
    const u32x4b rs_v = bb_rsrc(a.res, (unsigned)((size_t)M * C4 * 2));
    const unsigned lds0 = (unsigned)(size_t)(lds_void_b*)smem;
    // W3 piece p: K sub-image p >> 2, rows 8 (p & 3) .. + 7, 8 chunks of 16 B each (chunk index XOR-swizzled on the source
    // side).  Source = the step-interleaved copy (b2b_pack_w3): image row `row` of step t is matrix row row * NSTEP + t.
    auto w3_off = [&](const int p) {
        const int row = 8 * (p & 3) + (lane >> 3);
        return (unsigned)(row * NSTEP * C2 * 2 + (p >> 2) * 128 + (((lane & 7) ^ ((row >> 1) & 7)) * 16));
    };
    // W1' piece p: rows 16 p .. + 15 of the N2 x 64 B slice, 4 chunks each
    auto w1_off = [&](const int p) {
        const int row = 16 * p + (lane >> 2);
        const int src = (row & ~31) | b2b_pi(row & 31);
        return (unsigned)src * (unsigned)(C4 * 2) + (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
    };
    unsigned w3_voff[NPW], w1_voff[NPW], rs_voff[NPR];
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        w3_voff[i] = w3_off(wave * NPW + i);
        w1_voff[i] = w1_off(wave * NPW + i);
    }
#pragma unroll
    for (int i = 0; i < NPR; i++) {  // residual piece i: pixels 16 i .. + 15 of the wave, the step's 64 bytes each
        const int row = 16 * i + (lane >> 2);
        rs_voff[i] = (unsigned)(m0w + row) * (unsigned)(C4 * 2) + (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
    }
    auto dma_step = [&](const int t, const int slot) {
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int p = wave * NPW + i;
            bb_dma16(w3_v, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::OFF_W3 + slot * L::W3_SLICE + p * 1024)), w3_voff[i],
                     __builtin_amdgcn_readfirstlane((unsigned)(t * C2 * 2)));
        }
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int p = wave * NPW + i;
            bb_dma16(w1_v, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::OFF_W1 + slot * L::W1_SLICE + p * 1024)), w1_voff[i],
                     __builtin_amdgcn_readfirstlane((unsigned)(t * BB_NT * 2)));
        }
#pragma unroll
        for (int i = 0; i < NPR; i++)
            bb_dma16(rs_v, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::OFF_R + slot * L::RSLOT + wave * (RB * 2048) + i * 1024)), rs_voff[i],
                     __builtin_amdgcn_readfirstlane((unsigned)(t * BB_NT * 2)));
    };
    dma_step(0, 0);
    if (NSTEP > 1) dma_step(1, 1);

    // lane-constant LDS addresses (ring base folded in: slot, K sub-image, row block are immediates of the DS instruction)
    const char* w3_lane[4];  // weight fragment of GEMM 1: row r of a [32][128 B] sub-image, chunk 2 q + hh
#pragma unroll
    for (int q = 0; q < 4; q++) w3_lane[q] = smem + L::OFF_W3 + r * 128 + (((2 * q + hh) ^ ((r >> 1) & 7)) * 16);
    int d_lane[2];           // row r of a [rows][64 B] image, chunk 2 s + hh (C/D layout of 16 channels)
#pragma unroll
    for (int s = 0; s < 2; s++) d_lane[s] = r * 64 + (((2 * s + hh) ^ ((r >> 2) & 3)) * 16);
    const char* w1_lane[2];  // weight fragment of GEMM 2
    char* r_lane[2];         // residual / y of this wave in the C/D layout
#pragma unroll
    for (int s = 0; s < 2; s++) {
        w1_lane[s] = smem + L::OFF_W1 + d_lane[s];
        r_lane[s] = smem + L::OFF_R + wave * (RB * 2048) + d_lane[s];
    }
    const int t_row = lane >> 2, t_chunk = lane & 3;  // y store: 4 lanes x 16 B per pixel, 16 pixels per instruction
    const char* t_lane = smem + L::OFF_R + wave * (RB * 2048) + t_row * 64 + ((t_chunk ^ ((t_row >> 2) & 3)) * 16);
    const float* b3_lane = reinterpret_cast<const float*>(smem + L::OFF_B3) + 8 * hh;

    const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)((size_t)M * C4 * 2), 0x00020000);
    const unsigned y_voff = (unsigned)(m0w + t_row) * (unsigned)(C4 * 2) + (unsigned)(t_chunk * 16);

    f32x16b acc2[RB][NJ];
#pragma unroll
    for (int ib = 0; ib < RB; ib++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc2[ib][j][e] = 0.0f;
    // Everything the compiler knows to be in flight (the activation fragments, the bias tables) is waited for HERE, with the
    // builtin, so that its own wait-count bookkeeping starts the loop empty: otherwise it guards the first use of every
    // fragment register inside the loop with s_waitcnt vmcnt(18) ... vmcnt(3) -- counts that know nothing of the DMA pieces
    // issued by inline asm and would drain the two steps of prefetch on every step.
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) expcnt(7) lgkmcnt(0): also puts the bias tables in LDS before the first barrier
    // (s_setprio(1) for waves 4-7 -- the younger half loses every issue arbitration on its SIMD -- evens out the two halves'
    //  phase times in the trace but not the kernel time: 173.5 vs 170.0 us; not used)
#ifdef B2B_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
    unsigned long long tlast = tstart;
#endif

    auto step = [&](auto SC, const int t) __attribute__((always_inline)) {
        constexpr int S = decltype(SC)::value;
        // this wave's pieces of step t have landed: younger than them are only the PP pieces of step t + 1 (and stores,
        // which can only make the wait longer: loads retire in order among themselves)
        if (t + 1 < NSTEP)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BB_T(0);
        __builtin_amdgcn_s_barrier();  // every wave's pieces of step t are in LDS, and every wave is done with step t - 1
        BB_T(1);
        // ---- GEMM 1: acc1[channel pi-row][pixel] over all C2 input channels; PF weight fragments in flight ----
        auto w3_frag = [&](const int sl) { return *reinterpret_cast<const h16x8b*>(w3_lane[sl & 3] + S * L::W3_SLICE + (sl >> 2) * 4096); };
        auto w1_frag = [&](const int f) { return *reinterpret_cast<const h16x8b*>(w1_lane[f / NJ] + S * L::W1_SLICE + (f % NJ) * 2048); };
        h16x8b fw[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) fw[i] = w3_frag(i);
        // The slot of step t - 1 takes step t + 2.  A DMA piece holds its wave at the issue stage until the texture-address
        // path has taken it (scripts/b2b_trace.py: the 48 pieces of a step occupy that path about as long as the step's
        // MFMAs take), so the two waves of a SIMD issue theirs at DIFFERENT points of the step: waves 0-3 here, waves 4-7
        // after their y stores (dma_phase 2) -- while one is held, the other computes.  4K layer3 pair, same box: all waves
        // here 192.8 us, waves 4-7 after GEMM 1 182.8, after the y stores 170.0 (the two launches it replaces: 239).
        const bool late = NW == 8 && wave >= 4 && dma_phase != 0;
        if (!late && t + 2 < NSTEP) dma_step(t + 2, (S + 2) % BB_NSLOT);
        BB_T(2);
        f32x16b acc1[RB];
#pragma unroll
        for (int ib = 0; ib < RB; ib++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc1[ib][e] = 0.0f;
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
#pragma unroll
            for (int ib = 0; ib < RB; ib++) acc1[ib] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[sl % PF], areg[ib][sl], acc1[ib], 0, 0, 0);
            if (sl + PF < NSL) fw[sl % PF] = w3_frag(sl + PF);
        }
        if (late && dma_phase == 1 && t + 2 < NSTEP) dma_step(t + 2, (S + 2) % BB_NSLOT);
        BB_T(3);
        // ---- epilogue 1: + bias, + residual, ReLU, f16 (same order as the unfused conv3) ----
        h16x8b yop[RB][2];
#pragma unroll
        for (int ib = 0; ib < RB; ib++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                char* rp = r_lane[s] + S * L::RSLOT + ib * 2048;
                const h16x8b rv = *reinterpret_cast<const h16x8b*>(rp);
                const float* bp = b3_lane + t * BB_NT + 16 * s;
                const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
                const float bias[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float v = acc1[ib][8 * s + e];
                    v += bias[e];
                    v += (float)rv[e];
                    v = fmaxf(v, 0.f);
                    yop[ib][s][e] = (_Float16)v;
                }
                *reinterpret_cast<h16x8b*>(rp) = yop[ib][s];  // in place: a lane reads and writes only its own chunks
            }
        BB_T(4);
        // the first fragments of GEMM 2 travel while y goes out
#pragma unroll
        for (int i = 0; i < PF; i++) fw[i] = w1_frag(i);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PF) : "memory");  // (the y chunks are in LDS; the PF fragment reads behind them may still fly)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2 * RB; it++) {
            const u32x4b yv = *reinterpret_cast<const u32x4b*>(t_lane + S * L::RSLOT + it * 1024);
            __builtin_amdgcn_raw_buffer_store_b128(yv, y_rsrc, y_voff + (unsigned)(it * 16 * C4 * 2), (unsigned)(t * BB_NT * 2), 0);
        }
        if (late && dma_phase == 2 && t + 2 < NSTEP) dma_step(t + 2, (S + 2) % BB_NSLOT);
        BB_T(5);
        // ---- GEMM 2: this step's 32 channels of y are two k-slices of conv1' ----
#pragma unroll
        for (int f = 0; f < 2 * NJ; f++) {
#pragma unroll
            for (int ib = 0; ib < RB; ib++) acc2[ib][f % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[f % PF], yop[ib][f / NJ], acc2[ib][f % NJ], 0, 0, 0);
            if (f + PF < 2 * NJ) fw[f % PF] = w1_frag(f + PF);
        }
        BB_T(6);
    };

    for (int t = 0; t < NSTEP; t += BB_NSLOT) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < NSTEP) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < NSTEP) step(std::integral_constant<int, 2>{}, t + 2);
    }

#ifdef B2B_TRACE
    if (blockIdx.x == B2B_TRACE_WG && lane == 0) {
        tr[7] = tlast - tstart;
#pragma unroll
        for (int k = 0; k < 8; k++) g_b2b_trace[wave * 8 + k] = tr[k];
    }
#endif
    // ---- epilogue 2: t1' = ReLU(acc2 + b1') as f16, through the wave's own staging slice (over the idle rings) so that a
    //      pixel's N2 * 2 contiguous bytes leave in 16-byte lanes ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* fs = smem + wave * (32 * RB * N2 * 2);
#pragma unroll
    for (int ib = 0; ib < RB; ib++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const float* bp = reinterpret_cast<const float*>(smem + L::OFF_B1) + 32 * j + 16 * s + 8 * hh;
                const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
                const float bias[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                h16x8b hv;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float v = acc2[ib][j][8 * s + e] + bias[e];
                    v = fmaxf(v, 0.f);
                    hv[e] = (_Float16)v;
                }
                const int c = 4 * j + 2 * s + hh;
                *reinterpret_cast<h16x8b*>(fs + (32 * ib + r) * (N2 * 2) + ((c ^ (r & 7)) * 16)) = hv;
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = N2 / 8, RPI = 64 / CPR;  // 16-byte chunks per pixel row, pixel rows per store instruction
    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out2, 0, (unsigned)((size_t)M * N2 * 2), 0x00020000);
#pragma unroll
    for (int it = 0; it < 32 * RB / RPI; it++) {
        const int row = it * RPI + lane / CPR, c = lane % CPR;
        const u32x4b v = *reinterpret_cast<const u32x4b*>(fs + row * (N2 * 2) + ((c ^ (row & 7)) * 16));
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rsrc, (unsigned)(m0w + row) * (unsigned)(N2 * 2) + (unsigned)(c * 16), 0, 0);
    }
}

template <int KS, int RB, int NW>
hipError_t launch_form(const B2bArgs& a, hipStream_t s) {
    constexpr int BM = NW * RB * 32;
    constexpr int kLds = B2bLds<KS, BM>::TOTAL;
    const int mtiles = (a.M + BM - 1) / BM;
    auto k = conv1x1_b2b_kernel<KS, RB, NW>;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    // (0 / 1 / 2 only: any other value would leave the late waves' pieces un-issued -- which, as an ablation, runs the 4K layer3 pair in
    //  157 instead of 188 us with HALF the weight / residual pieces missing: the step is bound by the DMA ingest, not by its MFMAs)
    static const int dma_env = getenv("INFUR_B2B_DMA") ? atoi(getenv("INFUR_B2B_DMA")) : 2;
    static const int dma_phase = dma_env < 0 || dma_env > 2 ? 2 : dma_env;
    hipLaunchKernelGGL(k, dim3(mtiles), dim3(NW * 64), kLds, s, a, mtiles, dma_phase);
    return hipGetLastError();
}

}  // namespace

bool conv1x1_b2b_valid(const B2bArgs& a) {
    return (a.C2 == 128 || a.C2 == 256) && a.M > 0 && a.relu1 && a.relu2 &&  // (both convs of a bottleneck end in ReLU)
           a.in && a.w3 && a.b3 && a.res && a.y && a.w1 && a.b1 && a.out2 &&
           // 32-bit buffer offsets; rows past M must stay below 2^32 as well
           ((size_t)a.M + BB_BM) * (size_t)a.C2 * 4 * 2 < 0x80000000ull;
}

#ifdef B2B_TRACE
extern "C" int32_t infur_debug_b2b_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_b2b_trace), sizeof(g_b2b_trace)) == hipSuccess ? 0 : 1;
}
#endif

hipError_t launch_b2b_pack_w3(const void* w3, void* w3i, int C2, hipStream_t s) {
    if (C2 != 128 && C2 != 256) return hipErrorInvalidValue;
    const size_t chunks = (size_t)4 * C2 * C2 / 8;
    hipLaunchKernelGGL(b2b_pack_w3_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const _Float16*)w3, (_Float16*)w3i, C2);
    return hipGetLastError();
}

hipError_t launch_conv1x1_b2b(const B2bArgs& a, hipStream_t s) {
    if (!conv1x1_b2b_valid(a)) return hipErrorInvalidValue;
    // measurement hook: INFUR_B2B_FORM=1 (default) two waves per SIMD x 32 pixels; 2 one wave per SIMD x 64 pixels with the
    // whole register file -- half the LDS fragment reads, but nothing covers the wave while it is held at a DMA issue or in
    // its epilogue: 237 against 170 us on the 4K layer3 pair (same box)
    // INFUR_B2B_FORM (measurement hook): 1 two waves per SIMD x 32 pixels; 2 one wave per SIMD x 64 pixels; 3 the 128-pixel workgroup;
    // default: 3 when 256-pixel workgroups would leave more than a third of the 256 CUs without one, else 1
    static const int form_env = getenv("INFUR_B2B_FORM") ? atoi(getenv("INFUR_B2B_FORM")) : 0;
    const int mt256 = (a.M + 255) / 256;
    const int form = form_env >= 1 && form_env <= 3 ? form_env : (mt256 <= 170 ? 3 : 1);
    if (form == 1) return a.C2 == 256 ? launch_form<4, 1, 8>(a, s) : launch_form<2, 1, 8>(a, s);
    if (form == 3) return a.C2 == 256 ? launch_form<4, 1, 4>(a, s) : launch_form<2, 1, 4>(a, s);
    return a.C2 == 256 ? launch_form<4, 2, 4>(a, s) : launch_form<2, 2, 4>(a, s);
}

}  // namespace infur
