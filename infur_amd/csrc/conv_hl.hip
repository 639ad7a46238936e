// conv_hl.hip -- INFUR_DTYPE_F16_HL: convolution as an implicit GEMM on THREE-BYTE tensors, two MFMA units per product.
//
// Replaces the Conv / Add / Relu nodes ONNX Runtime executes inside `session.run` (infur/src/predict_onnx.rs:138) in the one
// arithmetic of this library that is meant to satisfy BOTH halves of north_star's sentence -- logits within 1e-3 of the f32
// reference AND f16-matrix-core rate.  Round 4 measured why neither existing mode does: the f16 mode's
// 11-bit operands are ~3 bits short at every one of ~50 rounding sites, and the split modes that have the bits keep 4-byte
// tensors and stage them global -> VGPR -> split -> LDS, which bounds them at a CU's VMEM issue rate (MfmaUtil 26 %).
//
// Tensor format ("HL"): every activation tensor [pixels][C] is TWO planes
//      hi  [pixels][C] f16   = rne16(x)
//      lo  [pixels][C] e5m2  = rne8((x - hi) * kHlLoScale)          kHlLoScale = 2^11 * kHlDebias
// written by the PRODUCER's epilogue, so that a consumer stages both planes by LDS-DMA (buffer_load ... lds: no staging
// registers, no conversion, no ds_write) -- 3 bytes per element instead of 4 through HBM, L2 and the CU's ingest path.
// e5m2 has f16's exponent range: no tensor-level scale exists anywhere (round 4: e4m3 under per-tensor scales broke on
// heavy-tailed parameters).  Weights are split the same way once at model load (launch_hl_pack_weights), after the per-layer
// power-of-two scale of the split modes.
//
// Product (per 32-channel K step and 32x32 block):   a w  ~=  a_hi w_hi                          2 x v_mfma_f32_32x32x16_f16
//                                                       + t(a_hi) w_lo8 + a_lo8 t(w_hi)         1 x v_mfma_scale_f32_32x32x64_f8f6f4 (bf8)
// where t(.) is the TOP BYTE of the f16 -- an e5m2 value by truncation, extracted from the f16 fragment registers with one
// v_perm_b32 per four values: neither operand needs a third plane or a conversion.  Truncation shortens the magnitude by a
// factor whose mean is 1 / kHlDebias; that constant is folded into the lo planes (the term t(a_hi) w_lo8 carries it through
// w_lo8, the term a_lo8 t(w_hi) through a_lo8), which leaves the truncated byte with the error distribution of a ROUNDED one
// (scripts/sim_hl_assign.py: 9.3e-5 max-abs / 6.6e-3 per element on the hostile set against 1.0e-4 / 6.7e-3 with every byte
// rounded; 8.1e-3 with both bytes truncated and no third plane -- this kernel).  The bf8 MFMA's E8M0 block scale undoes the 2^11.
// Two MFMA units per product (the bf8 instruction runs 64-deep at twice the f16 rate), ~2^-14 relative per product.
//
// GEMM view as conv_igemm_kernel.h:  M = OH*OW pixels, N = Cout, K = KH*KW*Cin (tap-major, Cin inner); D rows = output
// channels, D columns = pixels.  K step = 32 channels: LDS rows are 64 B (hi) / 32 B (lo), a workgroup image is
// (BM + BN) * 96 bytes and THREE images form a ring (256x256: 144 KB): the DMA of step k+2 is in flight across the barrier
// of step k, every wave waits for its own pieces with a counted vmcnt.  Bank conflicts: a DMA piece lands lane-linear, so rows
// cannot be padded; the 16-byte chunk index is XOR-swizzled with (row >> 2) & 3 (64-byte rows) / (row >> 3) & 1 (32-byte rows),
// applied to the SOURCE address of the DMA and to the fragment reads -- conflict-free for the hardware's ds_read_b128 lane
// groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (MI355X_MICROARCH.md; tests/test_lds_layout_cpu.py checks the arithmetic).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "conv_hl_dev.h"

namespace infur {

#ifndef HL_TWOBAR
#define HL_TWOBAR 1
#endif

// Instrumentation build (make EXTRA="-DHL_TRACE -DHLT_CIN=2048 -DHLT_COUT=512", scripts/hl_trace.py): workgroup HLT_WG of the launches with
// that shape accumulates, per wave, the shader cycles (s_memtime) of each phase of the pipelined K loop.  Timestamps sit where no LDS
// read is outstanding (s_memtime shares lgkmcnt with them).  Not compiled into the product library.
#ifdef HL_TRACE
#ifndef HLT_WG
#define HLT_WG 100
#endif
__device__ unsigned long long g_hl_trace[8 * 8];
#define HLT_T(k) do { if (tr_on) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); tr[k] += now__ - tlast; tlast = now__; } } while (0)
#else
#define HLT_T(k) do { } while (0)
#endif

constexpr int hl_image_bytes(int bm, int bn) { return (bm + bn) * 96; }
constexpr int hl_lds_bytes(int bm, int bn, int wm, int wn, int nimg) {
    const int operands = nimg * hl_image_bytes(bm, bn);
    const int staging = wm * wn * 32 * (bn / wn * 4 + 16);
    return operands > staging ? operands : staging;
}
constexpr int hl_ceil_div(int a, int b) { return (a + b - 1) / b; }

// DUAL (with G1): the K loop runs over two activation tensors in turn (ConvArgs.in, then ConvArgs.in2 sampled with stride2) against
// one weight matrix whose rows are the two 1x1 kernels side by side -- a bottleneck's conv3 and the downsample branch of a stage's
// first block in one launch (conv_igemm_kernel.h: DUAL): the branch output is neither written nor re-read as a residual.
// NIMG: LDS images in the ring.  3 = the DMA of step k + 2 in flight across the barrier of step k (eight waves, one workgroup per
// CU).  2 = one step ahead: half-size tiles of FOUR waves with 128 x 64 wave tiles fit TWO independent workgroups per CU (72 KB
// each, one wave per SIMD each) -- the epilogue of one overlaps the K loop of the other, which is what the output-bound 1x1
// expansions (conv3: K = 256 / 512, a tile's epilogue moves as many bytes as its K loop ingests) are short of.
// PIPE (round 6): the K loop software-pipelined ACROSS the barrier.  The plain loop (PIPE = false) opens every step with its 12
// fragment reads and `s_waitcnt lgkmcnt(0)` -- all eight waves at once, right behind the barrier: ~96 KB through the LDS (~600 cycles)
// during which no SIMD has an MFMA to issue, then 2 x 1024 cycles of MFMAs: the 3.1 k cycles per 2 k of scripts/micro/ingest_rate.hip.
// The pipelined loop puts the step's one barrier in its MIDDLE, between the f16 MFMAs (hi x hi, two slices) and the bf8 MFMAs (cross
// terms), and the first thing after it is the read of the NEXT step's slice-0 fragments: they land under the 512 cycles of bf8 MFMAs
// the wave (and its SIMD partner) still has to issue; slice 1 and the lo planes are read under the slice-0 MFMAs.  After the barrier of
// step k every wave has finished ALL reads of image k (slice 0 before, slice 1 + lo during the first half of the step, drained by the
// lgkmcnt(0) in front of the barrier), so the DMA of step k + NIMG goes out there, into the image just freed -- the same two steps of
// cover as before (ring of three), one image less idle.  Same MFMAs on the same operands in the same order per accumulator:
// bit-identical to PIPE = false (tests/test_gpu_hl.py).
template <int BM, int BN, int WM, int WN, bool G1, bool OUTF32, bool DUAL = false, int NIMG = 3, bool PIPE = false>
__global__ void __launch_bounds__(WM* WN * 64, 2) conv_hl_kernel(const ConvArgs a, const int mtiles, const int ntiles) {
    static_assert(!DUAL || (G1 && !OUTF32), "DUAL is a form of the 1x1 GEMM addressing");
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int IMG = hl_image_bytes(BM, BN);
    constexpr int A_HI = 0, A_LO = BM * 64, B_HI = BM * 96, B_LO = BM * 96 + BN * 64;
    // DMA pieces (1 KB each) per plane and per wave; a plane whose pieces do not divide among the waves lets the surplus waves
    // repeat a piece (same bytes to the same place: harmless) so that every wave issues the same number -- the counted vmcnt
    constexpr int P_AH = BM / 16, P_AL = BM / 32, P_BH = BN / 16, P_BL = BN / 32;
    // LW: the waves that issue the DMA.  Pipelined loop with two waves per SIMD (eight waves): only the FIRST half -- waves w and w + 4 share a
    // SIMD, the older one wins the issue arbitration, finishes both halves of a step early and idles ~1.2 k cycles per step at the barrier
    // while the younger one, held at its DMA issues beside the other's MFMAs, is the step's critical path (s_memtime trace, LAB_NOTES
    // round 6): the pieces go where the slack is.
    constexpr int LW = (PIPE && NW == 8) ? 4 : NW;
    constexpr int I_AH = hl_ceil_div(P_AH, LW), I_AL = hl_ceil_div(P_AL, LW), I_BH = hl_ceil_div(P_BH, LW), I_BL = hl_ceil_div(P_BL, LW);
    constexpr int NPW = I_AH + I_AL + I_BH + I_BL;  // pieces per wave per K step
    static_assert(NPW <= 16, "vmcnt immediates below are written for <= 16 pieces per step");

    // MODE.FP16_OVFL = 1: an f32 -> f16 conversion that overflows clamps to +-65504 instead of producing inf
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order (conv_igemm_kernel.h): every XCD a contiguous run of tiles, n fastest
    const int nblk = mtiles * ntiles * (a.batch > 1 ? a.batch : 1);
    int tile;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, loc = b >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int per_batch = mtiles * ntiles;
    const int bidx = tile / per_batch;
    tile -= bidx * per_batch;
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: per-wave offsets live in SGPRs)
#ifdef HL_TRACE
    const bool tr_on = PIPE && a.Cin == HLT_CIN && a.Cout == HLT_COUT && blockIdx.x == HLT_WG;
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    const int wm = wave / WN, wn = wave % WN;
    const bool loader = LW == NW || wave < LW;  // (wave-uniform)
    const int M = a.OH * a.OW;
    const int Ktot = DUAL ? a.Cin + a.Cin2 : a.KH * a.KW * a.Cin;
    const int cchunks = a.Cin / HL_KC;
    const int ksteps = DUAL ? cchunks + a.Cin2 / HL_KC : a.KH * a.KW * cchunks;

    // descriptors of the four planes (batched use: plane b of the hi tensor starts at b * in_bs, of the lo tensor at b * in_bs / 2)
    auto mk = [](const void* p, unsigned bytes) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        u32x4r r;
        r.x = __builtin_amdgcn_readfirstlane((unsigned)v);
        r.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        r.z = __builtin_amdgcn_readfirstlane(bytes);
        r.w = 0x00020000u;
        return r;
    };
    const size_t in_elems = (size_t)a.H * a.W * a.Cin, wt_elems = (size_t)a.Cout * Ktot;
    const u32x4r ah_v = mk(static_cast<const char*>(a.in) + (size_t)bidx * a.in_bs, (unsigned)(in_elems * 2));
    const u32x4r al_v = mk(static_cast<const char*>(a.in_lo) + (size_t)bidx * (a.in_bs / 2), (unsigned)in_elems);
    const u32x4r bh_v = mk(static_cast<const char*>(a.wt) + (size_t)bidx * a.wt_bs, (unsigned)(wt_elems * 2));
    const u32x4r bl_v = mk(static_cast<const char*>(a.wt_lo) + (size_t)bidx * (a.wt_bs / 2), (unsigned)wt_elems);
    const size_t in2_elems = DUAL ? (size_t)a.H2 * a.W2 * a.Cin2 : 0;
    const u32x4r a2h_v = mk(DUAL ? a.in2 : a.in, (unsigned)(in2_elems * 2));
    const u32x4r a2l_v = mk(DUAL ? a.in2_lo : a.in_lo, (unsigned)in2_elems);
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;

    // per-lane source coordinates of the pieces this wave issues.  hi piece p: rows 16 p .. 16 p + 15, lane l = row l >> 2 at
    // LDS chunk position l & 3, i.e. data chunk (l & 3) ^ swz64(row); lo piece p: rows 32 p .. 32 p + 31, lane l = row l >> 1 at
    // position l & 1.  G1: one byte offset per piece for the whole K loop (the K step advances through the scalar offset);
    // otherwise the pixel's (iy0, ix0) and the chunk offset.
    int ah_y[I_AH], ah_x[I_AH], al_y[I_AL], al_x[I_AL];
    unsigned ah_dst[I_AH], al_dst[I_AL], bh_off[I_BH], bl_off[I_BL], bh_dst[I_BH], bl_dst[I_BL];
    unsigned ah_c[I_AH], al_c[I_AL];
#pragma unroll
    for (int i = 0; i < I_AH; i++) {
        const int p = (wave * I_AH + i) % P_AH;
        const int row = 16 * p + (lane >> 2);
        const unsigned ch = (unsigned)(((lane & 3) ^ hl_swz64(row)) * 16);
        const int m = m0 + row;
        const int oy = m / a.OW, ox = m - oy * a.OW;
        ah_dst[i] = (unsigned)(A_HI + p * 1024);
        ah_c[i] = ch;
        if constexpr (G1) {
            ah_y[i] = m < M ? (int)((unsigned)(oy * a.stride * a.W + ox * a.stride) * (unsigned)(a.Cin * 2) + ch) : (int)HL_OOB;
            ah_x[i] = 0;
            if constexpr (DUAL)  // the same output pixel in the second tensor
                ah_x[i] = m < M ? (int)((unsigned)(oy * a.stride2 * a.W2 + ox * a.stride2) * (unsigned)(a.Cin2 * 2) + ch) : (int)HL_OOB;
        } else {
            ah_y[i] = m < M ? oy * a.stride - a.pad : -0x100000;
            ah_x[i] = ox * a.stride - a.pad;
        }
    }
#pragma unroll
    for (int i = 0; i < I_AL; i++) {
        const int p = (wave * I_AL + i) % P_AL;
        const int row = 32 * p + (lane >> 1);
        const unsigned ch = (unsigned)(((lane & 1) ^ hl_swz32(row)) * 16);
        const int m = m0 + row;
        const int oy = m / a.OW, ox = m - oy * a.OW;
        al_dst[i] = (unsigned)(A_LO + p * 1024);
        al_c[i] = ch;
        if constexpr (G1) {
            al_y[i] = m < M ? (int)((unsigned)(oy * a.stride * a.W + ox * a.stride) * (unsigned)a.Cin + ch) : (int)HL_OOB;
            al_x[i] = 0;
            if constexpr (DUAL)
                al_x[i] = m < M ? (int)((unsigned)(oy * a.stride2 * a.W2 + ox * a.stride2) * (unsigned)a.Cin2 + ch) : (int)HL_OOB;
        } else {
            al_y[i] = m < M ? oy * a.stride - a.pad : -0x100000;
            al_x[i] = ox * a.stride - a.pad;
        }
    }
#pragma unroll
    for (int i = 0; i < I_BH; i++) {
        const int p = (wave * I_BH + i) % P_BH;
        const int row = 16 * p + (lane >> 2);
        const int n = n0 + row;
        bh_dst[i] = (unsigned)(B_HI + p * 1024);
        bh_off[i] = n < a.Cout ? (unsigned)n * 64u + (unsigned)(((lane & 3) ^ hl_swz64(row)) * 16) : HL_OOB;  // (K-block-major weights)
    }
#pragma unroll
    for (int i = 0; i < I_BL; i++) {
        const int p = (wave * I_BL + i) % P_BL;
        const int row = 32 * p + (lane >> 1);
        const int n = n0 + row;
        bl_dst[i] = (unsigned)(B_LO + p * 1024);
        bl_off[i] = n < a.Cout ? (unsigned)n * 32u + (unsigned)(((lane & 1) ^ hl_swz32(row)) * 16) : HL_OOB;
    }

    // K step being LOADED: its index, its tap and channel chunk, the ring image it goes to
    int kl = 0, ky = 0, kx = 0, cc = 0;
    unsigned ld_img = 0;  // byte offset of the image the next load fills
    auto sdst = [&](unsigned off) { return __builtin_amdgcn_readfirstlane(lds0 + ld_img + off); };
    auto load_a = [&]() {
        if constexpr (G1) {
            if (DUAL && kl >= cchunks) {  // wave-uniform: the second tensor's K steps
                const unsigned k2 = (unsigned)(kl - cchunks);
#pragma unroll
                for (int i = 0; i < I_AH; i++) hl_dma16(a2h_v, sdst(ah_dst[i]), (unsigned)ah_x[i], k2 * 64u);
#pragma unroll
                for (int i = 0; i < I_AL; i++) hl_dma16(a2l_v, sdst(al_dst[i]), (unsigned)al_x[i], k2 * 32u);
            } else {
#pragma unroll
                for (int i = 0; i < I_AH; i++) hl_dma16(ah_v, sdst(ah_dst[i]), (unsigned)ah_y[i], (unsigned)kl * 64u);
#pragma unroll
                for (int i = 0; i < I_AL; i++) hl_dma16(al_v, sdst(al_dst[i]), (unsigned)al_y[i], (unsigned)kl * 32u);
            }
        } else {
            const int dy = ky * a.dil, dx = kx * a.dil;
#pragma unroll
            for (int i = 0; i < I_AH; i++) {
                const int iy = ah_y[i] + dy, ix = ah_x[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off = (unsigned)(iy * a.W + ix) * (unsigned)(a.Cin * 2) + (unsigned)(cc * 64) + ah_c[i];
                hl_dma16(ah_v, sdst(ah_dst[i]), ok ? off : HL_OOB, 0u);
            }
#pragma unroll
            for (int i = 0; i < I_AL; i++) {
                const int iy = al_y[i] + dy, ix = al_x[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off = (unsigned)(iy * a.W + ix) * (unsigned)a.Cin + (unsigned)(cc * 32) + al_c[i];
                hl_dma16(al_v, sdst(al_dst[i]), ok ? off : HL_OOB, 0u);
            }
        }
    };
    auto load_b = [&]() {
#pragma unroll
        for (int i = 0; i < I_BH; i++) hl_dma16(bh_v, sdst(bh_dst[i]), bh_off[i], (unsigned)kl * (unsigned)(a.Cout * 64));
#pragma unroll
        for (int i = 0; i < I_BL; i++) hl_dma16(bl_v, sdst(bl_dst[i]), bl_off[i], (unsigned)kl * (unsigned)(a.Cout * 32));
    };
    auto load_next = [&]() {  // advance the load cursor (all wave-uniform scalars)
        kl++;
        cc++;
        const int w1 = cc == cchunks;
        cc = w1 ? 0 : cc;
        kx += w1;
        const int w2 = kx == a.KW;
        kx = w2 ? 0 : kx;
        ky += w2;
        ld_img = ld_img + IMG == NIMG * IMG ? 0u : ld_img + IMG;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // fragment addresses: lane (row r = lane & 31, half h = lane >> 5) reads hi chunk 2 h + kk (slice kk = 0, 1) and lo chunk h of
    // its row -- the lane half h covers channels 16 h .. 16 h + 15 of the step in BOTH planes, in channel order
    const int r = lane & 31, h = lane >> 5;
    const int a_hi0 = A_HI + (wm * TM * 32 + r) * 64 + (((2 * h) ^ hl_swz64(r)) * 16);
    const int a_hi1 = A_HI + (wm * TM * 32 + r) * 64 + (((2 * h + 1) ^ hl_swz64(r)) * 16);
    const int a_lo = A_LO + (wm * TM * 32 + r) * 32 + ((h ^ hl_swz32(r)) * 16);
    const int b_hi0 = B_HI + (wn * TN * 32 + r) * 64 + (((2 * h) ^ hl_swz64(r)) * 16);
    const int b_hi1 = B_HI + (wn * TN * 32 + r) * 64 + (((2 * h + 1) ^ hl_swz64(r)) * 16);
    const int b_lo = B_LO + (wn * TN * 32 + r) * 32 + ((h ^ hl_swz32(r)) * 16);

    // Residual prefetch (tiles with <= 64 accumulators per lane: room for the whole residual tile in registers): the loads go out at
    // the start of the first K step that issues no DMA any more (ks = ksteps - 2) -- every DMA piece still in flight is then OLDER
    // than they are, so the step-end wait becomes vmcnt(NRES) and the residual's HBM latency hides behind the last two steps
    // instead of standing in front of every 32-row block of the epilogue (conv_igemm_kernel.h: RESPF, measured there: the epilogue
    // of a K = 512 conv outlasted its K loop).  Unconditional loads (a guard's branch makes hipcc wait for each load in turn).
    constexpr int E_CPL = 8, E_LPR = TN * 32 / E_CPL, E_RPI = 64 / E_LPR, E_NIT = 32 / E_RPI;
    constexpr bool CANPF = !OUTF32 && TM * TN * 16 <= 64;
    constexpr int NRES = TM * E_NIT * 2;
    const bool pf = CANPF && a.res != nullptr;
    f16x8 prh[CANPF ? TM : 1][CANPF ? E_NIT : 1];
    u32x2 prl[CANPF ? TM : 1][CANPF ? E_NIT : 1];
    const int pf_step = ksteps >= 2 ? ksteps - 2 : 0;  // (ring of two: the residual loads are then OLDER than the last step's DMA)


    if constexpr (PIPE) {
        // ---- software-pipelined K loop (see the comment on PIPE above), rotated: one iteration = [barrier of step ks | next step's
        // slice-0 reads, the DMA pieces of step ks + NIMG one between every two MFMAs, bf8 MFMAs of step ks | slice-1 + lo reads and
        // the f16 MFMAs of step ks + 1].  The prefetched slice-0 set is defined and consumed inside one iteration; what crosses the
        // back edge is the bf8 operand set (a8 / b8), written unconditionally. ----
        int nload = 0;
#pragma unroll
        for (int p = 0; p < NIMG; p++)
            if (p < ksteps) {
                if (loader) {
                    load_a();
                    load_b();
                }
                load_next();
                nload++;
            }
        if (NIMG == 3 && nload == 3)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
        else if (nload == 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int pf_mid = ksteps >= NIMG ? ksteps - NIMG : 0;  // the residual loads go out in the first iteration that issues no DMA any more (they must be the YOUNGEST loads)
        i32x8 a8[TM], b8[TN];
        // one DMA piece of the step being loaded (p = 0 .. NPW - 1: A hi, A lo, B hi, B lo), for the interleave below
        auto load_piece = [&](const int p) {
            if (p < I_AH + I_AL) {
                const bool hi = p < I_AH;
                const int i = hi ? p : p - I_AH;
                if constexpr (G1) {
                    if (DUAL && kl >= cchunks) {
                        const unsigned k2 = (unsigned)(kl - cchunks);
                        if (hi) hl_dma16(a2h_v, sdst(ah_dst[i < I_AH ? i : 0]), (unsigned)ah_x[i < I_AH ? i : 0], k2 * 64u);
                        else hl_dma16(a2l_v, sdst(al_dst[i < I_AL ? i : 0]), (unsigned)al_x[i < I_AL ? i : 0], k2 * 32u);
                    } else {
                        if (hi) hl_dma16(ah_v, sdst(ah_dst[i < I_AH ? i : 0]), (unsigned)ah_y[i < I_AH ? i : 0], (unsigned)kl * 64u);
                        else hl_dma16(al_v, sdst(al_dst[i < I_AL ? i : 0]), (unsigned)al_y[i < I_AL ? i : 0], (unsigned)kl * 32u);
                    }
                } else {
                    const int dy = ky * a.dil, dx = kx * a.dil;
                    if (hi) {
                        const int q = i < I_AH ? i : 0;
                        const int iy = ah_y[q] + dy, ix = ah_x[q] + dx;
                        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        const unsigned off = (unsigned)(iy * a.W + ix) * (unsigned)(a.Cin * 2) + (unsigned)(cc * 64) + ah_c[q];
                        hl_dma16(ah_v, sdst(ah_dst[q]), ok ? off : HL_OOB, 0u);
                    } else {
                        const int q = i < I_AL ? i : 0;
                        const int iy = al_y[q] + dy, ix = al_x[q] + dx;
                        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        const unsigned off = (unsigned)(iy * a.W + ix) * (unsigned)a.Cin + (unsigned)(cc * 32) + al_c[q];
                        hl_dma16(al_v, sdst(al_dst[q]), ok ? off : HL_OOB, 0u);
                    }
                }
            } else {
                const int pb = p - I_AH - I_AL;
                if (pb < I_BH) hl_dma16(bh_v, sdst(bh_dst[pb < I_BH ? pb : 0]), bh_off[pb < I_BH ? pb : 0], (unsigned)kl * (unsigned)(a.Cout * 64));
                else {
                    const int q = pb - I_BH < I_BL ? pb - I_BH : 0;
                    hl_dma16(bl_v, sdst(bl_dst[q]), bl_off[q], (unsigned)kl * (unsigned)(a.Cout * 32));
                }
            }
        };
        // first half of step `img`: slice 1 and the lo planes are read under the slice-0 MFMAs; leaves the step's bf8 operands in a8 / b8
        constexpr int NMF = TM * TN;
        // DMA pieces issued in the bf8 half / in the f16 half of an iteration.  Ring of three: half and half (two steps of cover either way).
        // Ring of two: ALL of them in the bf8 half, right behind the barrier -- a piece issued in the f16 half would be waited for at the
        // end of that same half (one step of cover is all this ring has: the trace of a layer4 conv3 tile showed 0.75-1.0 k cycles of
        // counted-vmcnt wait per step, LAB_NOTES round 6)
        constexpr int P1 = NIMG == 2 ? NPW : NPW / 2, P2 = NPW - P1;
        // Ring of two, TWO barriers per step (round 6, after the trace of a layer4 conv3 tile: 0.75-1.0 k cycles of vmcnt wait per step): the
        // barrier in front of the bf8 MFMAs only frees the image of step k for the DMA of step k + 2; the wait for step k + 1 and the
        // barrier that publishes it come BEHIND the bf8 MFMAs -- half a step more cover for every piece -- and the next step's slice 0 is
        // read there, in front of its MFMAs.
        constexpr bool TWOBAR = NIMG == 2 && HL_TWOBAR;
        // With two waves per SIMD the ORDER still matters: a ds_read_b128 holds its wave ~16 cycles at issue, an LDS-DMA piece 50-60, an MFMA
        // covers 32 (f16) / 64 (bf8): twelve reads in a row in front of the first MFMA of a half leave the pipe to the partner alone --
        // which is in the same place of the same code behind the same barrier.  So every read and every piece sits behind an MFMA of
        // its own (conv3x3_halo.hip's one-filler-per-MFMA rule), pinned with sched_barrier.
        auto first_half = [&](const char* I, const uint4* f0a, const uint4* f0b, const bool issue) {
            uint4 f1a[TM], f1b[TN];
            constexpr int NRD = 2 * (TM + TN);  // slice-1 hi fragments, then the lo planes
            auto read_op = [&](const int k) {
                if (k < TM) f1a[k] = *reinterpret_cast<const uint4*>(I + a_hi1 + k * 32 * 64);
                else if (k < TM + TN) f1b[k - TM] = *reinterpret_cast<const uint4*>(I + b_hi1 + (k - TM) * 32 * 64);
                else if (k < 2 * TM + TN) {
                    const int i = k - TM - TN;
                    const uint4 t = *reinterpret_cast<const uint4*>(I + a_lo + i * 32 * 32);
                    a8[i][4] = (int)t.x; a8[i][5] = (int)t.y; a8[i][6] = (int)t.z; a8[i][7] = (int)t.w;
                } else {
                    const int j = k - 2 * TM - TN;
                    const uint4 t = *reinterpret_cast<const uint4*>(I + b_lo + j * 32 * 32);
                    b8[j][0] = (int)t.x; b8[j][1] = (int)t.y; b8[j][2] = (int)t.z; b8[j][3] = (int)t.w;
                }
            };
            // the second half of the step's DMA pieces goes out between these MFMAs (all of them behind one barrier crowd the CU's
            // one texture path: 48 pieces x >= 16 cycles each inside the ~1 k cycles of the bf8 half)
            auto pieces_behind = [&](const int m2) {
                if constexpr (P2 > 0) {
                    if (issue) {
#pragma unroll
                        for (int p = P1; p < NPW; p++)
                            if ((p - P1) * (2 * NMF) / P2 == m2) load_piece(p);
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int m = i * TN + j;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f0b[j]), __builtin_bit_cast(f16x8, f0a[i]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NRD; k++)
                        if (k * NMF / NRD == m) read_op(k);
                    pieces_behind(m);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int i = 0; i < TM; i++) {
                a8[i][0] = hl_top4(f0a[i].x, f0a[i].y);
                a8[i][1] = hl_top4(f0a[i].z, f0a[i].w);
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                b8[j][4] = hl_top4(f0b[j].x, f0b[j].y);
                b8[j][5] = hl_top4(f0b[j].z, f0b[j].w);
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f1b[j]), __builtin_bit_cast(f16x8, f1a[i]), acc[i][j], 0, 0, 0);
                    pieces_behind(NMF + i * TN + j);
                }
#pragma unroll
            for (int i = 0; i < TM; i++) {
                a8[i][2] = hl_top4(f1a[i].x, f1a[i].y);
                a8[i][3] = hl_top4(f1a[i].z, f1a[i].w);
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                b8[j][6] = hl_top4(f1b[j].x, f1b[j].y);
                b8[j][7] = hl_top4(f1b[j].z, f1b[j].w);
            }
        };
        // the wait in front of the barrier of step ks: step ks + 1 has landed (this wave's pieces), every read of image ks is back in
        // registers; in flight behind it: the pieces of step ks + 2 (ring of three), the residual
        auto wait_mid = [&](const int ks) {
            const bool res_out = CANPF && pf && ks > pf_mid;
            if (NIMG == 3 && ks + 2 < ksteps)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPW) : "memory");
            else if (res_out)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NRES) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        };
        {
            uint4 f0a[TM], f0b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) f0a[i] = *reinterpret_cast<const uint4*>(smem + a_hi0 + i * 32 * 64);
#pragma unroll
            for (int j = 0; j < TN; j++) f0b[j] = *reinterpret_cast<const uint4*>(smem + b_hi0 + j * 32 * 64);
            first_half(smem, f0a, f0b, false);
            if constexpr (TWOBAR)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else
                wait_mid(0);
        }
        HLT_T(0);  // prologue + the first step's f16 half
        unsigned cur = 0;
        for (int ks = 0;; ks++) {
            __builtin_amdgcn_s_barrier();
            HLT_T(1);  // at the barrier
            const unsigned nxt = cur + IMG == NIMG * IMG ? 0u : cur + IMG;
            // next step's slice 0 (unconditional: behind the last step it reads an image nobody uses any more), one read behind each of
            // the first bf8 MFMAs; the first half of the DMA pieces of step ks + NIMG behind the MFMAs as well
            uint4 f0a[TM], f0b[TN];
            const bool more = kl < ksteps;  // step ks + NIMG exists: its pieces go out between the MFMAs, into the image of step ks
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j], a8[i], acc[i][j], 1, 1, 0, 127 - kHlLoShift, 0, 127);
                    const int m = i * TN + j;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < TM + TN; k++)
                        if (!TWOBAR && k * NMF / (TM + TN) == m) {
                            if (k < TM) f0a[k] = *reinterpret_cast<const uint4*>(smem + nxt + a_hi0 + k * 32 * 64);
                            else f0b[k - TM] = *reinterpret_cast<const uint4*>(smem + nxt + b_hi0 + (k - TM) * 32 * 64);
                        }
                    if (more && loader) {
#pragma unroll
                        for (int p = 0; p < P1; p++)
                            if (p * NMF / P1 == m) load_piece(p);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            if constexpr (CANPF) {
                if (pf && ks == pf_mid) {
                    const int e_row = lane / E_LPR, e_col = lane % E_LPR;
                    const int n = n0 + wn * TN * 32 + e_col * E_CPL;
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int it = 0; it < E_NIT; it++) {
                            const int m = m0 + wm * TM * 32 + i * 32 + it * E_RPI + e_row;
                            const size_t e = (m < M && n < a.Cout) ? (size_t)m * a.Cout + n : 0;
                            prh[i][it] = *reinterpret_cast<const f16x8*>(static_cast<const _Float16*>(a.res) + e);
                            prl[i][it] = *reinterpret_cast<const u32x2*>(static_cast<const unsigned char*>(a.res_lo) + e);
                        }
                }
            }
            if (ks + 1 == ksteps) break;  // (more is false here: nothing is left half-issued)
            HLT_T(2);  // slice-0 reads of the next step, DMA issue, bf8 MFMAs
            if constexpr (TWOBAR) {
                // step ks + 1 has landed: in flight behind it only what this iteration issued (the pieces of step ks + 2, or the residual)
                if (more)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
                else if (CANPF && pf && ks >= pf_mid)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < TM; i++) f0a[i] = *reinterpret_cast<const uint4*>(smem + nxt + a_hi0 + i * 32 * 64);
#pragma unroll
                for (int j = 0; j < TN; j++) f0b[j] = *reinterpret_cast<const uint4*>(smem + nxt + b_hi0 + j * 32 * 64);
            }
            first_half(smem + nxt, f0a, f0b, more && loader);
            if (more) load_next();
            HLT_T(3);  // slice-1 + lo reads, f16 MFMAs
            if constexpr (TWOBAR)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (every read of image ks + 1 is back: the barrier at the loop's top frees it)
            else
                wait_mid(ks + 1);
            HLT_T(4);  // counted vmcnt + lgkmcnt(0)
            cur = nxt;
        }
        // (the last step's barrier stands between every wave's last LDS read and the epilogue's staging stores -- but the unconditional
        //  slice-0 read behind it is still outstanding: drain it before the staging stores reuse the registers' LDS region)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HLT_T(5);  // the last step's second half
    } else {
        // prologue: steps 0 and 1 into images 0 and 1 (ring of two: step 0)
        load_a();
        load_b();
        load_next();
        if (NIMG == 3 && ksteps > 1) {
            load_a();
            load_b();
            load_next();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();

        unsigned cur = 0;  // byte offset of the image being multiplied
        for (int ks = 0; ks < ksteps; ks++) {
            if constexpr (CANPF) {
                if (pf && ks == pf_step) {
                    const int e_row = lane / E_LPR, e_col = lane % E_LPR;
                    const int n = n0 + wn * TN * 32 + e_col * E_CPL;
    #pragma unroll
                    for (int i = 0; i < TM; i++)
    #pragma unroll
                        for (int it = 0; it < E_NIT; it++) {
                            const int m = m0 + wm * TM * 32 + i * 32 + it * E_RPI + e_row;
                            const size_t e = (m < M && n < a.Cout) ? (size_t)m * a.Cout + n : 0;
                            prh[i][it] = *reinterpret_cast<const f16x8*>(static_cast<const _Float16*>(a.res) + e);
                            prl[i][it] = *reinterpret_cast<const u32x2*>(static_cast<const unsigned char*>(a.res_lo) + e);
                        }
                }
            }
            const char* I = smem + cur;
            const bool more = kl < ksteps;  // step ks + 2 (ring of two: ks + 1) exists: its pieces go out between the slices
            uint4 fa[TM], fb[TN], fal[TM], fbl[TN];
            i32x8 a8[TM], b8[TN];
            // slice 0
    #pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const uint4*>(I + a_hi0 + i * 32 * 64);
    #pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const uint4*>(I + b_hi0 + j * 32 * 64);
    #pragma unroll
            for (int i = 0; i < TM; i++) fal[i] = *reinterpret_cast<const uint4*>(I + a_lo + i * 32 * 32);
    #pragma unroll
            for (int j = 0; j < TN; j++) fbl[j] = *reinterpret_cast<const uint4*>(I + b_lo + j * 32 * 32);
            if (more) load_a();
    #pragma unroll
            for (int i = 0; i < TM; i++) {
                a8[i][0] = hl_top4(fa[i].x, fa[i].y);
                a8[i][1] = hl_top4(fa[i].z, fa[i].w);
            }
    #pragma unroll
            for (int j = 0; j < TN; j++) {
                b8[j][4] = hl_top4(fb[j].x, fb[j].y);
                b8[j][5] = hl_top4(fb[j].z, fb[j].w);
            }
    #pragma unroll
            for (int i = 0; i < TM; i++)
    #pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa[i]), acc[i][j], 0, 0, 0);
            // slice 1
    #pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const uint4*>(I + a_hi1 + i * 32 * 64);
    #pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const uint4*>(I + b_hi1 + j * 32 * 64);
            if (more) {
                load_b();
                load_next();
            }
    #pragma unroll
            for (int i = 0; i < TM; i++) {
                a8[i][2] = hl_top4(fa[i].x, fa[i].y);
                a8[i][3] = hl_top4(fa[i].z, fa[i].w);
                a8[i][4] = (int)fal[i].x; a8[i][5] = (int)fal[i].y; a8[i][6] = (int)fal[i].z; a8[i][7] = (int)fal[i].w;
            }
    #pragma unroll
            for (int j = 0; j < TN; j++) {
                b8[j][6] = hl_top4(fb[j].x, fb[j].y);
                b8[j][7] = hl_top4(fb[j].z, fb[j].w);
                b8[j][0] = (int)fbl[j].x; b8[j][1] = (int)fbl[j].y; b8[j][2] = (int)fbl[j].z; b8[j][3] = (int)fbl[j].w;
            }
    #pragma unroll
            for (int i = 0; i < TM; i++)
    #pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa[i]), acc[i][j], 0, 0, 0);
            // cross terms: [t(a_hi) | a_lo8] . [w_lo8 | t(w_hi)], 64 deep, e5m2 x e5m2, block scale 2^-11
    #pragma unroll
            for (int i = 0; i < TM; i++)
    #pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j], a8[i], acc[i][j], 1, 1, 0, 127 - kHlLoShift, 0, 127);
            if (NIMG == 3 && more)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPW) : "memory");
            else if (pf && !more)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NRES) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur = cur + IMG == NIMG * IMG ? 0u : cur + IMG;
        }

    }

    // ---- epilogue: * acc_scale, + bias, + residual (HL), ReLU; HL planes or f32 out.  Each wave passes its 32-pixel row blocks
    // through its own slice of the idle operand LDS (conv_igemm_kernel.h) and stores with LPR lanes per pixel row ----
    const float acc_scale = a.acc_scale_b ? a.acc_scale_b[bidx] : a.acc_scale;
    const bool has_bias = a.bias != nullptr;
    constexpr int CPL = OUTF32 ? 4 : 8;
    const bool vec_ok = (a.Cout & (CPL - 1)) == 0;
    if (vec_ok) {
        constexpr int ROWB = TN * 128 + 16;
        static_assert(NW * 32 * ROWB <= hl_lds_bytes(BM, BN, WM, WN, NIMG), "epilogue staging exceeds the LDS allocation");
        constexpr int LPR = TN * 32 / CPL, RPI = 64 / LPR;
        const int e_row = lane / LPR, e_col = lane % LPR;
        const int n = n0 + wn * TN * 32 + e_col * CPL;
        const bool n_ok = n < a.Cout;
        char* stage = smem + wave * 32 * ROWB;
        float bv[CPL];
#pragma unroll
        for (int t = 0; t < CPL; t++) bv[t] = (has_bias && n_ok) ? a.bias[n + t] : 0.f;
        const _Float16* res_hi = static_cast<const _Float16*>(a.res);
        const unsigned char* res_lo = static_cast<const unsigned char*>(a.res_lo);
        // Tiles too big for the prefetch above (256 x 256: 128 accumulators per lane): ALL residual loads of the wave go out here, at
        // once -- the fragment registers of the K loop are dead by now, so there is room for them -- instead of per 32-row block:
        // one HBM round trip per wave instead of TM of them (one workgroup per CU: nothing else would cover the other three).
        constexpr bool LATE = !OUTF32 && !CANPF;
        f16x8 lrh[LATE ? TM : 1][LATE ? E_NIT : 1];
        u32x2 lrl[LATE ? TM : 1][LATE ? E_NIT : 1];
        if constexpr (LATE) {
            if (res_hi) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int it = 0; it < E_NIT; it++) {
                        const int m = m0 + wm * TM * 32 + i * 32 + it * E_RPI + (lane / E_LPR);
                        const size_t e = (m < M && n_ok) ? (size_t)m * a.Cout + n : 0;
                        lrh[i][it] = *reinterpret_cast<const f16x8*>(res_hi + e);
                        lrl[i][it] = *reinterpret_cast<const u32x2*>(res_lo + e);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int jj = 0; jj < TN; jj++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float4 v = make_float4(acc[i][jj][4 * g + 0], acc[i][jj][4 * g + 1], acc[i][jj][4 * g + 2], acc[i][jj][4 * g + 3]);
                    v.x *= acc_scale; v.y *= acc_scale; v.z *= acc_scale; v.w *= acc_scale;
                    *reinterpret_cast<float4*>(stage + (lane & 31) * ROWB + (jj * 32 + 8 * g + 4 * (lane >> 5)) * 4) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int mb = m0 + wm * TM * 32 + i * 32;
            if constexpr (OUTF32) {
                float* out = reinterpret_cast<float*>(static_cast<char*>(a.out) + (size_t)bidx * a.out_bs);
#pragma unroll
                for (int it = 0; it < 32 / RPI; it++) {
                    const int row = it * RPI + e_row, m = mb + row;
                    float4 v = *reinterpret_cast<const float4*>(stage + row * ROWB + e_col * 16);
                    if (m < M && n_ok) {
                        v.x += bv[0]; v.y += bv[1]; v.z += bv[2]; v.w += bv[3];
                        if (a.relu) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        *reinterpret_cast<float4*>(out + (size_t)m * a.Cout + n) = v;
                    }
                }
            } else {
                _Float16* out_hi = static_cast<_Float16*>(a.out);
                unsigned char* out_lo = static_cast<unsigned char*>(a.out_lo);
                // all residual loads of the row block first (conv_igemm_kernel.h: rlate)
                f16x8 rh[32 / RPI];
                u32x2 rl[32 / RPI];
                if (res_hi) {
#pragma unroll
                    for (int it = 0; it < 32 / RPI; it++) {
                        if constexpr (CANPF) {
                            rh[it] = prh[i][it];
                            rl[it] = prl[i][it];
                        } else {
                            rh[it] = lrh[i][it];
                            rl[it] = lrl[i][it];
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < 32 / RPI; it++) {
                    const int row = it * RPI + e_row, m = mb + row;
                    const float4 v0 = *reinterpret_cast<const float4*>(stage + row * ROWB + e_col * 32);
                    const float4 v1 = *reinterpret_cast<const float4*>(stage + row * ROWB + e_col * 32 + 16);
                    if (m < M && n_ok) {
                        float x[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
                        if (res_hi) {
                            float lo[8];
                            hl_lo8_to_f32(rl[it].x, lo);
                            hl_lo8_to_f32(rl[it].y, lo + 4);
#pragma unroll
                            for (int t = 0; t < 8; t++) x[t] += (float)rh[it][t] + lo[t];
                        }
                        f16x8 hv;
                        u32x2 lv;
                        hl_split8(x, hv, lv, a.relu ? 0.f : -kHlHiMax);  // (the ReLU is the split's lower clamp)
                        *reinterpret_cast<f16x8*>(out_hi + (size_t)m * a.Cout + n) = hv;
                        *reinterpret_cast<u32x2*>(out_lo + (size_t)m * a.Cout + n) = lv;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
#ifdef HL_TRACE
        if (tr_on) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            HLT_T(6);  // epilogue to the last store's completion
            if (lane == 0)
                for (int k = 0; k < 8; k++) g_hl_trace[wave * 8 + k] = tr[k];
        }
#endif
        return;
    }
    // Cout not a multiple of the vector width (the 21-class logits, f32 out): element-wise from the accumulator layout
    if constexpr (OUTF32) {
        float* out = reinterpret_cast<float*>(static_cast<char*>(a.out) + (size_t)bidx * a.out_bs);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = m0 + wm * TM * 32 + i * 32 + (lane & 31);
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int n = n0 + wn * TN * 32 + j * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (n + t >= a.Cout) break;
                        float x = acc[i][j][4 * g + t] * acc_scale + (has_bias ? a.bias[n + t] : 0.f);
                        if (a.relu) x = fmaxf(x, 0.f);
                        out[(size_t)m * a.Cout + n + t] = x;
                    }
                }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool G1, bool OUTF32, bool DUAL = false, int NIMG = 3, bool PIPE = false>
static hipError_t launch_hl_g(const ConvArgs& a, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + BM - 1) / BM, ntiles = (a.Cout + BN - 1) / BN;
    const size_t lds = (size_t)hl_lds_bytes(BM, BN, WM, WN, NIMG);
    auto k = conv_hl_kernel<BM, BN, WM, WN, G1, OUTF32, DUAL, NIMG, PIPE>;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(mtiles * ntiles * (a.batch > 1 ? a.batch : 1)), dim3(WM * WN * 64), lds, s, a, mtiles, ntiles);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int NIMG, bool PIPE>
static hipError_t launch_hl_p(const ConvArgs& a, int out_f32, hipStream_t s) {
    const bool g1 = a.KH == 1 && a.KW == 1 && a.pad == 0;
    if (a.in2) return launch_hl_g<BM, BN, WM, WN, true, false, true, NIMG, PIPE>(a, s);  // (conv_hl_config_valid: 1x1, stride 1, hi / lo out)
    if (out_f32) return g1 ? launch_hl_g<BM, BN, WM, WN, true, true, false, NIMG, PIPE>(a, s) : launch_hl_g<BM, BN, WM, WN, false, true, false, NIMG, PIPE>(a, s);
    return g1 ? launch_hl_g<BM, BN, WM, WN, true, false, false, NIMG, PIPE>(a, s) : launch_hl_g<BM, BN, WM, WN, false, false, false, NIMG, PIPE>(a, s);
}
// INFUR_HL_PIPE=0: the plain K loop (measurement / bisection hook; both loops are bit-identical)
static bool hl_pipe_on() {
    static const bool on = !(getenv("INFUR_HL_PIPE") && atoi(getenv("INFUR_HL_PIPE")) == 0);
    return on;
}
template <int BM, int BN, int WM, int WN, int NIMG = 3>
static hipError_t launch_hl_t(const ConvArgs& a, int out_f32, hipStream_t s) {
    return hl_pipe_on() ? launch_hl_p<BM, BN, WM, WN, NIMG, true>(a, out_f32, s) : launch_hl_p<BM, BN, WM, WN, NIMG, false>(a, out_f32, s);
}

// configurations of mode 5 (indices of conv_igemm.hip's table whose tile dimensions they share): 11 = 256x256 (8 waves of
// 128x64), 0 = 128x128 (4 waves of 64x64), 6 = 256x128 (8 waves of 64x64), 5 = 128x256 (8 waves of 64x64); 12 = 256x128 and
// 14 = 128x256 as FOUR waves of 128x64 with a ring of two images: two workgroups per CU
bool conv_hl_config_valid(const ConvArgs& a, int cfg, int out_f32) {
    if (cfg == 15) return conv_hl_areg_valid(a, out_f32);  // conv_hl_areg.hip: the activation fragment in registers (1x1 expansions)
    if (cfg != 11 && cfg != 0 && cfg != 6 && cfg != 5 && cfg != 12 && cfg != 14 && cfg != 13 && cfg != 16 && cfg != 17) return false;
    if (!a.in_lo || !a.wt_lo) return false;
    if (a.in2 && (!a.in2_lo || out_f32 || a.res || a.KH != 1 || a.KW != 1 || a.pad != 0 || a.stride != 1 || a.Cin2 % HL_KC != 0 || a.batch > 1 ||
                  (size_t)a.H2 * a.W2 * a.Cin2 * 2 >= 0x80000000ull))
        return false;
    if (a.Cin % HL_KC != 0) return false;
    if ((size_t)a.H * a.W * a.Cin * 2 >= 0x80000000ull || (size_t)a.Cout * a.KH * a.KW * a.Cin * 2 >= 0x80000000ull) return false;
    if (out_f32 ? (a.res != nullptr) : (!a.out_lo || (a.res != nullptr) != (a.res_lo != nullptr) || (a.Cout & 7))) return false;
    if (a.batch > 1 && (a.in_bs & 1 || a.wt_bs & 1 || !out_f32)) return false;
    if ((size_t)a.Cout * (a.in2 ? a.Cin + a.Cin2 : a.KH * a.KW * a.Cin) * 2 >= 0x80000000ull) return false;
    const int bn = (cfg == 11 || cfg == 5 || cfg == 14 || cfg == 13 || cfg == 16) ? 256 : 128;
    return bn <= a.Cout || bn == 128;  // Cout < 128 (layer1, the logits): the 128-wide N tile with its surplus rows out of range
}

hipError_t launch_conv_hl(const ConvArgs& a, int out_f32, int cfg, hipStream_t s) {
    if (cfg < 0) cfg = a.Cout >= 256 && (size_t)a.OH * a.OW * (a.batch > 1 ? a.batch : 1) >= 256 * 200 ? 11 : 0;
    if (!conv_hl_config_valid(a, cfg, out_f32)) return hipErrorInvalidValue;
    switch (cfg) {
        case 15: return launch_conv_hl_areg(a, s);
        case 11: return launch_hl_t<256, 256, 2, 4>(a, out_f32, s);
        case 0: return launch_hl_t<128, 128, 2, 2>(a, out_f32, s);
        case 6: return launch_hl_t<256, 128, 4, 2>(a, out_f32, s);
        case 5: return launch_hl_t<128, 256, 2, 4>(a, out_f32, s);
        case 12: return launch_hl_t<256, 128, 2, 2, 2>(a, out_f32, s);
        case 14: return launch_hl_t<128, 256, 1, 4, 2>(a, out_f32, s);
        case 13: return launch_hl_t<256, 256, 4, 2>(a, out_f32, s);  // 8 waves of 64 x 128: a wave's epilogue rows are 128 channels wide
        // round 6: the two-workgroups-per-CU forms with 64 x 128 wave tiles (the expansions' epilogue moves whole 256 / 128-byte rows
        // of the hi / lo planes per pixel instead of 128 / 64)
        case 16: return launch_hl_t<128, 256, 2, 2, 2>(a, out_f32, s);
        case 17: return launch_hl_t<256, 128, 4, 1, 2>(a, out_f32, s);
        default: return hipErrorInvalidValue;
    }
}

// ---- weights: f32 [rows][cols] (already in the kernel's K order) * scale -> hi f16 at dst, lo e5m2 at dst_lo, both K-BLOCK-MAJOR:
//      element (n, k) at ((k / 32) * rows + n) * 32 + k % 32.  The 32 channels of a K step of `BN` consecutive output channels -- one
//      weight image of the kernels -- are then ONE contiguous block (BN x 64 bytes hi, BN x 32 bytes lo) and every 1-KB DMA piece
//      is eight whole 128-byte lines instead of sixteen half lines 2 Ktot bytes apart: the weights are half of what a K step
//      ingests, and the re-layout costs nothing (done once, at model load).  cols % 32 != 0 (no such GEMM layer): row-major. ----
// `planes` tensors back to back (blockIdx.y = plane, scale per plane in the argument block): one launch for a layer's 16 / 36 / 64 Winograd
// planes (round 6: one launch per plane made model load ~950 launches of this kernel)
struct HlPlaneScales {
    float s[64];
};
__global__ void hl_pack_weights_kernel(const float* __restrict__ w, size_t n4, unsigned rows, unsigned cols, const HlPlaneScales sc, _Float16* __restrict__ hi,
                                       unsigned* __restrict__ lo) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const bool blocked = (cols & 31u) == 0;
    const size_t pl = blockIdx.y;
    const float scale = sc.s[pl];
    w += pl * n4 * 4;
    hi += pl * n4 * 4;
    lo += pl * n4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(w)[i];
        const float x[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hv;
        float rem[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            hv[t] = (_Float16)x[t];
            rem[t] = x[t] - (float)hv[t];
        }
        size_t o = i;  // in units of four elements
        if (blocked) {
            const size_t e = i * 4, n = e / cols, k = e - n * cols;
            o = (((k >> 5) * rows + n) * 32 + (k & 31)) >> 2;
        }
        reinterpret_cast<f16x4*>(hi)[o] = hv;
        lo[o] = hl_pack_lo4(rem);
    }
}

// planes tensors [rows][cols] f32 back to back at w, plane p scaled by scales[p] -> hi planes back to back at hi (2 bytes / element), lo
// planes back to back at lo (1 byte / element)
hipError_t launch_hl_pack_weights_planes(const float* w, size_t rows, size_t cols, int planes, const float* scales, void* hi, void* lo, hipStream_t s) {
    const size_t n = rows * cols;
    if (n & 3 || rows > 0xffffffffull || cols > 0xffffffffull || planes < 1 || planes > 64) return hipErrorInvalidValue;
    size_t blocks = (n / 4 + 255) / 256;
    const size_t cap = planes > 1 ? 256 : 4096;
    if (blocks > cap) blocks = cap;
    HlPlaneScales sc;
    for (int p = 0; p < 64; p++) sc.s[p] = p < planes ? scales[p] : 0.f;
    hipLaunchKernelGGL(hl_pack_weights_kernel, dim3((unsigned)blocks, (unsigned)planes), dim3(256), 0, s, w, n / 4, (unsigned)rows, (unsigned)cols, sc, static_cast<_Float16*>(hi),
                       static_cast<unsigned*>(lo));
    return hipGetLastError();
}

hipError_t launch_hl_pack_weights(const float* w, size_t rows, size_t cols, float scale, void* hi, void* lo, hipStream_t s) {
    return launch_hl_pack_weights_planes(w, rows, cols, 1, &scale, hi, lo, s);
}

#ifdef HL_TRACE
extern "C" int infur_debug_hltrace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hl_trace), sizeof(unsigned long long) * 64);
}
#endif

}  // namespace infur
