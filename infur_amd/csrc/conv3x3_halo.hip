// conv3x3_halo.hip -- stride-1 3x3 convolution (pad = dilation), f16 operands, f32 accumulation: the INPUT PATCH of a 16 x 16
// output tile stays in LDS for all nine taps.  Replaces Conv nodes ONNX Runtime executes inside `session.run`
// (infur/src/predict_onnx.rs:138): layer2 / layer3 conv2 and the heads of FCN-ResNet in the f16 mode.
//
// Why (round 4): the tiled implicit GEMM (conv_igemm_kernel.h) gathers the activation tile of every (tap, channel chunk) K step
// from global memory -- nine shifted copies of nearly the same pixels per chunk.  At 1080p the stride-8 feature map has M =
// 32,400 pixels, i.e. ONE 256-pixel tile per CU, and what bounds such a launch is not the MFMA but what a CU can take in from L2:
// a 256 x 128 tile pulls 32 KB of activations + 16 KB of weights per K step for 1,024 cycles of MFMA work per SIMD -- 47 B/clk/CU
// against the ~18-23 B/clk/CU the L2 -> LDS path sustains with all 256 CUs pulling (layer3 conv2: 53 us = 0.72 PFLOP/s).  Here a
// workgroup owns a 16 x 16 block of output pixels; per 64-channel chunk it brings the (16 + 2d)^2 input pixels of the block's
// halo into LDS ONCE (324 / 400 / 576 rows of 128 bytes for d = 1 / 2 / 4 instead of 9 x 256) and all nine taps read their
// activation fragments from that patch at a per-tap row offset; only the weight tile (BN x 128 B) is new per tap.  Activation
// ingest falls 5.8x (d = 2), the K step's L2 -> LDS traffic from 48 KB to ~22 KB, LDS write traffic with it.
//
// Shape (the 4-wave form with 128 x 128 wave tiles has its own section below): 8 waves, BN = 64 (layer1: waves 8 x 1, 32 pixels x 64
// channels each), BN = 128 (4 x 2, 64 x 64) or BN = 256 (2 x 4, 128 x 64);
// everything arrives by
// LDS-DMA (buffer_load ... lds, rows of 128 bytes, 16-byte chunk index XOR-swizzled by (row >> 1) & 7 on the SOURCE side, as in
// conv_igemm_kernel.h); two weight images (tap t + 1 lands while tap t is multiplied), and two patch images when they fit the
// 160 KB (the next chunk's patch lands piece by piece during this chunk's nine taps: one piece of 8 rows per wave and tap),
// else one image and one extra barrier per chunk.  Out-of-image halo pixels are zeros through the descriptor's bounds check.
//
// Arithmetic: for every output element the k order is (channel chunk, tap, 16-wide slice) ascending with the weight fragment as
// the MFMA's row operand -- the order conv_igemm_kernel.h uses for f16 KxK convolutions since round 4 (CMAJ) -- then + bias, ReLU,
// round to f16: BIT-IDENTICAL to every tiled configuration (tests/test_gpu_conv_configs.py), so the tuner picks it by speed only.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "qepilogue.h"

namespace infur {

typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8h __attribute__((ext_vector_type(8)));
typedef unsigned u32x4h __attribute__((ext_vector_type(4)));
typedef int i32x4h __attribute__((ext_vector_type(4)));
typedef int i32x16h __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_h;

namespace {

constexpr unsigned HOOB = 0x80000000u;
constexpr int H_ROWB = 2 * 128 + 16;  // epilogue staging row of a wave: 64 f32 + pad

__host__ __device__ constexpr int h_swz(int row) { return (row >> 1) & 7; }

// Which pixel of a 32-row MFMA block a lane's column stands for.  The hardware serves a ds_read_b128 in the lane groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} (and the same + 32: MI355X_MICROARCH.md, LDS), chosen so that 32 CONSECUTIVE rows put 16 different
// rows mod 16 into every group.  A 16-wide tile puts two tile rows into a block -- patch rows p0 + tx and p0 + PW + tx with PW = 18 /
// 20 / 24 -- and with lane = pixel a group straddles both: rows p0 + {0-3, 12-15} and p0 + PW + {4-11} collide mod 16 (8.5 M
// bank-conflict cycles per launch of the 4-wave form, profiles/r04_halo4_pmc.log).  So the lanes of one hardware group take the 16
// pixels of ONE tile row: lane r stands for pixel (r & 15) of tile row sel(r).  A permutation of GEMM rows: no arithmetic changes.
__device__ __forceinline__ int h_pix(const int r, const int tw) {
    return tw == 16 ? ((r & 15) | ((((((r >> 2) & 3) + 1) >> 1) & 1) ^ (r >> 4)) << 4) : r;
}

__device__ __forceinline__ void h_dma16(const u32x4h rsrc, const unsigned lds, const unsigned voff, const unsigned soff) {
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

__device__ __forceinline__ u32x4h h_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    u32x4h r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

// A workgroup's output tile is th x tw pixels, th * tw <= 256 (GEMM rows beyond th * tw are padding).  tw is 16 or 32 ONLY: the
// hardware serves a ds_read_b128 in groups of 16 lanes, a group is conflict-free iff its 16 patch rows differ mod 16, and with the
// (row >> 1) & 7 swizzle that holds when the 16 lanes are 16 consecutive pixels of ONE tile row -- a 17 x 15 tile (which cuts
// 1080p's 135 x 240 map into exactly 128 tiles) wraps inside every group: 4.9 M bank-conflict cycles per launch, every activation
// fragment read served in two passes (profiles/r04_f16_summary.md, first collection).  The count of workgroups is made to fit the
// chip with TWO REGIONS instead: 16 x 16 tiles over the first floor(H / 16) * 16 rows and, when no more than 8 rows remain, a
// strip of (H mod 16) x 32 tiles -- 135 x 240: 120 + 8 = 128 tiles instead of 135 (270 workgroups with two N tiles: two rounds on
// 256 CUs, the second 5 % full).
struct HaloGeom {
    int th1, tw1, tiles_x1, n1;  // region 1: rows [0, ysplit)
    int ysplit;
    int th2, tw2, tiles_x2;      // region 2: rows [ysplit, H); th2 == 0: none
    int mtiles;
};
__host__ __device__ constexpr int h_pieces(int th, int tw, int d) { return ((th + 2 * d) * (tw + 2 * d) + 7) / 8; }  // wave instructions of 8 rows
__host__ __device__ constexpr int h_pimg(int th, int tw, int d) { return h_pieces(th, tw, d) * 1024; }
// Weight steps and images.  A tap's 64-channel weight tile is BN x 128 bytes: 16 KB for BN = 128, 32 KB for BN = 256.  BN = 128 takes
// it whole (one step = one tap = 1,024 MFMA cycles per SIMD); BN = 256 takes it in two 32-channel HALVES (64-byte rows, one step =
// half a tap = again 1,024 cycles), so that both forms stream 16 KB steps through a ring of THREE images (48 KB) -- steps this short
// are shorter than the L2 latency, with one step of prefetch every step ended waiting for its successor's weights -- and two patch
// images still fit beside them for d <= 2 (BN = 256 with whole-tap images needed 64 KB for two of them and one patch image: an
// exposed patch load per channel chunk).
__host__ __device__ constexpr int h_bkb(int bn) { return bn <= 128 ? 128 : 64; }  // bytes of k per weight step and row
__host__ __device__ constexpr int h_nb(int) { return 3; }
__host__ __device__ constexpr int h_lds(int bn, int na, int th, int tw, int d) {
    const int operands = na * h_pimg(th, tw, d) + h_nb(bn) * bn * h_bkb(bn);
    const int staging = 8 * 32 * H_ROWB;
    return operands > staging ? operands : staging;
}

// I8 (round 5): the same kernel on the quantised model's tensors -- u8 activations x s8 weights on v_mfma_i32_32x32x32_i8, a 128-byte
// pixel row holds 128 channels instead of 64, the requantisation of QLinearConv in the epilogue (qepilogue.h, the Q8 form of
// conv_igemm_kernel.h: 16 channels per lane).  Every byte address, swizzle and DMA piece is the f16 form's; integer accumulation is
// exact in any order, so the result is the tiled `dmai` form's, byte for byte (tests/test_gpu_quant.py).
template <int BN, int NA, bool I8 = false>
__global__ void __launch_bounds__(512, 2) conv3x3_halo_kernel(const ConvArgs a, const HaloGeom g, const int ntiles) {
    constexpr int ES = I8 ? 1 : 2;           // bytes per element
    constexpr int WN = BN / 64, WM = 8 / WN;  // waves along N (64 channels each) / along M
    constexpr int TM = 256 / WM / 32, TN = 2;
    constexpr int BKB = h_bkb(BN);           // bytes of k per weight step: a whole tap's 128, or half a tap
    constexpr int HS = 128 / BKB;            // weight steps per tap
    constexpr int SL = 4 / HS;               // 16-wide MFMA slices per weight step
    constexpr int BIMG = BN * BKB;           // one weight image
    constexpr int B_IT = BIMG / 8192;        // weight pieces (1 KB) per wave and step
    constexpr int NB = h_nb(BN);             // weight images (ring)
    constexpr int SPC = 9 * HS;              // weight steps per channel chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, hh = lane >> 5;
    int tile;
    {  // XCD-aware order (block b runs on XCD b % 8): every XCD gets a contiguous run of tiles, N fastest, so the N tiles
       // that share a patch share one L2
        const int nblk = g.mtiles * ntiles;
        const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
        const int q = nblk >> 3, rem = nblk & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    }
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const bool r2 = mt >= g.n1;  // (workgroup-uniform) the strip region
    const int th = r2 ? g.th2 : g.th1, tw = r2 ? g.tw2 : g.tw1, tiles_x = r2 ? g.tiles_x2 : g.tiles_x1;
    const int mloc = r2 ? mt - g.n1 : mt;
    const int tyb = mloc / tiles_x, txb = mloc - tyb * tiles_x;
    const int y0 = (r2 ? g.ysplit : 0) + tyb * th, x0 = txb * tw, n0 = nt * BN;
    const int d = a.dil, PW = tw + 2 * d, P = PW * (th + 2 * d);
    const int npiece = (P + 7) / 8, pimg = npiece * 1024;
    const int npix = th * tw;              // GEMM rows that are pixels of the tile (the rest of the 256 is padding)
    const float rtw = 1.0f / (float)tw;    // row -> (ty, tx): exact for these small integers
    const int Kb = a.Cin * ES;     // bytes of a pixel's channels
    const int cchunks = Kb / 128;

    char* const As = smem;              // [NA][npiece * 8 rows][128]
    char* const Bs = smem + NA * pimg;  // [NB][BN][BKB]
    const unsigned lds0 = (unsigned)(size_t)(lds_void_h*)smem;
    const u32x4h in_v = h_rsrc(a.in, (unsigned)((size_t)a.H * a.W * Kb));
    const u32x4h wt_v = h_rsrc(a.wt, (unsigned)((size_t)a.Cout * 9 * Kb));

    // ---- patch pieces: piece g = 8 t + wave is issued by this wave at tap t; lane l brings row p = 8 g + (l >> 3), LDS chunk
    //      position l & 7 = data chunk (l & 7) ^ swz(p).  The per-lane byte offset holds for the whole K loop: the channel
    //      chunk advances through the scalar offset. ----
    unsigned a_voff[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int p = 8 * (8 * t + wave) + (lane >> 3);
        const int py = p / PW, px = p - py * PW;
        const int iy = y0 - d + py, ix = x0 - d + px;
        const bool ok = p < P && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        a_voff[t] = ok ? (unsigned)(iy * a.W + ix) * (unsigned)Kb + (unsigned)(((lane & 7) ^ h_swz(p)) * 16) : HOOB;
    }
    // weight pieces: 1 KB = 8 rows of 128 bytes (chunk index XOR (row >> 1) & 7) or 16 rows of 64 bytes (XOR (row >> 2) & 3): either
    // way a 16-lane fragment read group covers the 64 banks exactly once
    unsigned b_voff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        const int row = (BKB == 128 ? 8 : 16) * (wave * B_IT + i) + (BKB == 128 ? (lane >> 3) : (lane >> 2));
        const int chunk = BKB == 128 ? ((lane & 7) ^ h_swz(row)) : ((lane & 3) ^ ((row >> 2) & 3));
        const int n = n0 + row;
        b_voff[i] = n < a.Cout ? (unsigned)n * (unsigned)(9 * Kb) + (unsigned)(chunk * 16) : HOOB;
    }
    auto dma_a = [&](const int t, const int cc, const int img) {  // this wave's patch piece of tap slot t, channel chunk cc
        if (8 * t + wave < npiece)  // (wave-uniform)
            h_dma16(in_v, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(img * pimg + (8 * t + wave) * 1024)), a_voff[t],
                    __builtin_amdgcn_readfirstlane((unsigned)(cc * 128)));
    };
    auto dma_b = [&](const int cc, const int q, const int img) {  // weight step q = tap * HS + half of chunk cc: rows n0 .. n0 + BN - 1
        const int tap = q / HS, half = q - tap * HS;
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((tap * cchunks + cc) * 128 + half * BKB));
#pragma unroll
        for (int i = 0; i < B_IT; i++)
            h_dma16(wt_v, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(NA * pimg + img * BIMG + (wave * B_IT + i) * 1024)), b_voff[i], soff);
    };

    // ---- lane-constant fragment addressing ----
    int pbase[TM];  // patch row of this lane's pixel in block i at tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int R = (wm * TM + i) * 32 + h_pix(r, tw);
        const int ty = (int)(((float)R + 0.5f) * rtw), tx = R - ty * tw;
        pbase[i] = R < npix ? ty * PW + tx : 0;  // (padding rows read a valid patch row; their results are never stored)
    }
    const int b_lds = (wn * 64 + r) * BKB;
    const int b_sw = BKB == 128 ? h_swz(r) : ((r >> 2) & 3);  // fragment rows are 32 apart: the swizzle does not change

    f32x16h acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // ---- prologue: the whole patch of chunk 0, the weights of the first NB - 1 taps ----
#pragma unroll
    for (int t = 0; t < 9; t++) dma_a(t, 0, 0);
    dma_b(0, 0, 0);
    dma_b(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto wait_vm = [](const int n) {  // s_waitcnt takes an immediate
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        }
    };

    int bimg = 0;  // weight image of the current tap
    for (int cc = 0; cc < cchunks; cc++) {
        const int aimg = NA == 2 ? (cc & 1) : 0;
        const bool more_chunks = cc + 1 < cchunks;
        // (an opaque zero: without it the per-tap fragment addresses below are loop-invariant and the compiler keeps all 9 x TM of
        //  them -- and their swizzles -- in registers across the chunk loop: 90 spilled VGPRs in the BN = 256 form)
        int zero = 0;
        asm volatile("" : "+s"(zero));
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int toff = (ky * PW + kx) * d + zero;  // (scalar) patch-row offset of this tap
            // chunk (2 kk + hh) of row p sits at position (2 kk + hh) ^ swz(p) = (kk << 1) ^ (hh ^ swz(p)): one base address per
            // block and tap, one XOR with a constant per slice (the swizzle of a patch row depends on the tap's row offset, so
            // -- unlike the tiled kernel's -- it cannot be folded into the instruction's immediate)
            unsigned abase[TM];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int p = pbase[i] + toff;
                abase[i] = (unsigned)(aimg * pimg + p * 128) | (unsigned)((hh ^ h_swz(p)) << 4);
            }
#pragma unroll
            for (int half = 0; half < HS; half++) {
                const int q = tap * HS + half;
                // what this step issues: (first step of a tap) the tap's patch piece of the next chunk, and the weights NB - 1 steps ahead
                const bool issue_a = half == 0 && NA == 2 && more_chunks && 8 * tap + wave < npiece;
                const bool issue_b = more_chunks || q + NB - 1 < SPC;
                const int bnext = bimg + NB - 1 >= NB ? bimg - 1 : bimg + NB - 1;  // (bimg + NB - 1) % NB
                const char* Bb = Bs + bimg * BIMG + b_lds;
                auto read_slice = [&](const int kl, h16x8h (&fa)[TM], h16x8h (&fb)[TN]) {  // slice kl of this step = slice half * SL + kl of the tap
#pragma unroll
                    for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const h16x8h*>(As + (abase[i] ^ (unsigned)((half * SL + kl) << 5)));
#pragma unroll
                    for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const h16x8h*>(Bb + j * 32 * BKB + (((2 * kl + hh) ^ b_sw) * 16));
                };
                // fragments run ONE slice ahead of the MFMAs (two register sets): with a fence per slice and no prefetch every slice
                // paid its ds_read latency in full -- 3.8k cycles per tap for 1k of MFMA work in the first cut
                h16x8h fa[2][TM], fb[2][TN];
                read_slice(0, fa[0], fb[0]);
#pragma unroll
                for (int kl = 0; kl < SL; kl++) {
                    if (kl + 1 < SL) read_slice(kl + 1, fa[(kl + 1) & 1], fb[(kl + 1) & 1]);
                    // the DMA instructions go out behind a slice's fragment reads, in the shadow of MFMAs that already have their
                    // operands (the `dmai` placement of conv_igemm_kernel.h)
                    if (kl == 0 && issue_a) dma_a(tap, cc + 1, aimg ^ 1);
                    if (kl == (SL > 1 ? 1 : 0) && issue_b) dma_b(q + NB - 1 >= SPC ? cc + 1 : cc, q + NB - 1 >= SPC ? q + NB - 1 - SPC : q + NB - 1, bnext);
#pragma unroll
                    for (int i = 0; i < TM; i++) {
                        // (I8: activations are u8, the MFMA is signed: x ^ 0x80 = x - 128 as s8; the -128 * sum w is in q_bias, and an
                        //  out-of-image halo pixel -- zeros from the DMA's bounds check -- is x = 0 = the zero point of a padded tensor)
                        const i32x4h ax = __builtin_bit_cast(i32x4h, fa[kl & 1][i]) ^ (int)0x80808080;
#pragma unroll
                        for (int j = 0; j < TN; j++) {
                            if constexpr (I8)
                                acc[i][j] = __builtin_bit_cast(f32x16h, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4h, fb[kl & 1][j]), ax,
                                                                                                               __builtin_bit_cast(i32x16h, acc[i][j]), 0, 0, 0));
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[kl & 1][j], fa[kl & 1][i], acc[i][j], 0, 0, 0);
                        }
                    }
                }
                // The NEXT step's weights must have landed (and, at a chunk's last step, the next chunk's whole patch); loads retire in
                // order, so what may stay in flight is exactly what was issued after them: this step's weight pieces (for two steps
                // ahead) and this step's patch piece -- which is never issued in a chunk's last step.
                wait_vm((issue_b ? B_IT : 0) + ((issue_a && q + 1 < SPC) ? 1 : 0));
                __builtin_amdgcn_s_barrier();
                bimg = bimg + 1 == NB ? 0 : bimg + 1;
            }
        }
        if (NA == 1 && more_chunks) {  // one patch image: it is free only now
#pragma unroll
            for (int t = 0; t < 9; t++) dma_a(t, cc + 1, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue: + bias, ReLU, f16; each wave passes its 32-pixel blocks through its own LDS slice so that 8 lanes store the
    //      128 contiguous bytes of a pixel's 64 channels (the f16 -> f16 form of conv_igemm_kernel.h, same steps per value) ----
    char* stage = smem + wave * 32 * H_ROWB;
    if constexpr (I8) {
        // QLinearConv's requantisation, 16 channels per lane (four lanes on the 64 bytes of a pixel's channels of this wave): the Q8
        // epilogue of conv_igemm_kernel.h without the residual sum (a 3x3 conv has none)
        const int q_row = lane >> 2, q_col = lane & 3;
        const int nq = n0 + wn * 64 + q_col * 16;
        const bool nq_ok = nq < a.Cout;
        int qb[16];
        float qm[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            qb[t] = 0;
            qm[t] = 0.f;
        }
        if (nq_ok) {
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) {
                const qi4 b4 = *reinterpret_cast<const qi4*>(a.q_bias + nq + 4 * t4);
                const qf4 m4 = *reinterpret_cast<const qf4*>(a.q_mult + nq + 4 * t4);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    qb[4 * t4 + t] = b4[t];
                    qm[4 * t4 + t] = m4[t];
                }
            }
        }
        const float q_yzpf = (float)a.q_yzp;
        const QEpi qe = {q_yzpf, -q_yzpf, 255.f - q_yzpf, a.q_ra, a.q_rb, (float)a.q_bzp, (float)a.q_czp};
        unsigned char* outq = static_cast<unsigned char*>(a.out);
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int jj = 0; jj < TN; jj++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const float4 v = make_float4(acc[i][jj][4 * g4 + 0], acc[i][jj][4 * g4 + 1], acc[i][jj][4 * g4 + 2], acc[i][jj][4 * g4 + 3]);
                    *reinterpret_cast<float4*>(stage + h_pix(r, tw) * H_ROWB + (jj * 32 + 8 * g4 + 4 * hh) * 4) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int row = it * 16 + q_row;
                const int R = (wm * TM + i) * 32 + row;
                const int ty = (int)(((float)R + 0.5f) * rtw), tx = R - ty * tw;
                const int oy = y0 + ty, ox = x0 + tx;
                if (R < npix && oy < a.OH && ox < a.OW && nq_ok) {
                    u32x4h pk;
#pragma unroll
                    for (int t4 = 0; t4 < 4; t4++) {
                        const qi4 ai = *reinterpret_cast<const qi4*>(stage + row * H_ROWB + q_col * 64 + t4 * 16);
                        const qi4 a4 = {ai[0] + qb[4 * t4], ai[1] + qb[4 * t4 + 1], ai[2] + qb[4 * t4 + 2], ai[3] + qb[4 * t4 + 3]};
                        const qf4 m4 = {qm[4 * t4], qm[4 * t4 + 1], qm[4 * t4 + 2], qm[4 * t4 + 3]};
                        pk[t4] = q_word<false>(a4, m4, 0u, qe);
                    }
                    *reinterpret_cast<u32x4h*>(outq + ((size_t)oy * a.OW + ox) * a.Cout + nq) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    const int e_row = lane >> 3, e_col = lane & 7;
    const int n = n0 + wn * 64 + e_col * 8;
    const bool n_ok = n < a.Cout;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bv2 = bv;
    if (a.bias && n_ok) {
        bv = *reinterpret_cast<const float4*>(a.bias + n);
        bv2 = *reinterpret_cast<const float4*>(a.bias + n + 4);
    }
    _Float16* out = static_cast<_Float16*>(a.out);
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int jj = 0; jj < TN; jj++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 v = make_float4(acc[i][jj][4 * g + 0], acc[i][jj][4 * g + 1], acc[i][jj][4 * g + 2], acc[i][jj][4 * g + 3]);
                *reinterpret_cast<float4*>(stage + h_pix(r, tw) * H_ROWB + (jj * 32 + 8 * g + 4 * hh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 8 + e_row;
            const int R = (wm * TM + i) * 32 + row;
            const int ty = (int)(((float)R + 0.5f) * rtw), tx = R - ty * tw;
            const int oy = y0 + ty, ox = x0 + tx;
            if (R < npix && oy < a.OH && ox < a.OW && n_ok) {
                const float4 v = *reinterpret_cast<const float4*>(stage + row * H_ROWB + e_col * 32);
                const float4 v2 = *reinterpret_cast<const float4*>(stage + row * H_ROWB + e_col * 32 + 16);
                float x[8] = {v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w, v2.x + bv2.x, v2.y + bv2.y, v2.z + bv2.z, v2.w + bv2.w};
                if (a.relu) {
#pragma unroll
                    for (int t = 0; t < 8; t++) x[t] = fmaxf(x[t], 0.f);
                }
                const h16x8h hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3],
                                   (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
                *reinterpret_cast<h16x8h*>(out + ((size_t)oy * a.OW + ox) * a.Cout + n) = hv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// Tiling and patch images for one launch: fewest rounds of workgroups on the chip first, then the fewest tiles, then the smallest
// patch.  Candidates: uniform 16 x 16, uniform 8 x 32, and 16 x 16 with an (H mod 16) x 32 strip when H mod 16 <= 8.
struct HaloPlan {
    HaloGeom g{};
    int na = 0, lds = 0;
};
HaloPlan halo_plan(const ConvArgs& a, int bn, int es = 2) {
    HaloPlan best;
    const int kchunks = a.Cin * es / 128;  // 128-byte channel chunks per pixel (f16: 64 channels each, i8: 128)
    long best_cost[3] = {0, 0, 0};
    const int ntiles = a.Cout / bn, d = a.dil, H = a.OH, W = a.OW;
    auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
    HaloGeom cands[3];
    int nc = 0;
    cands[nc++] = HaloGeom{16, 16, cdiv(W, 16), cdiv(H, 16) * cdiv(W, 16), H, 0, 0, 0, cdiv(H, 16) * cdiv(W, 16)};
    cands[nc++] = HaloGeom{8, 32, cdiv(W, 32), cdiv(H, 8) * cdiv(W, 32), H, 0, 0, 0, cdiv(H, 8) * cdiv(W, 32)};
    if (H >= 16 && H % 16 != 0 && H % 16 <= 8) {
        const int n1 = (H / 16) * cdiv(W, 16), n2 = cdiv(W, 32);
        cands[nc++] = HaloGeom{16, 16, cdiv(W, 16), n1, (H / 16) * 16, H % 16, 32, n2, n1 + n2};
    }
    for (int ci = 0; ci < nc; ci++)
        // (one channel chunk: there is no next patch to prefetch; BN = 64 exists in the single-patch-image form only -- launch_conv3x3_halo
        //  runs launch_halo<64, 1> -- so its plan, LDS size and workgroups-per-CU estimate must describe that form: ADVICE r4)
        for (int na = ((kchunks == 1 || bn == 64) ? 1 : 2); na >= 1; na--) {
            const HaloGeom& g = cands[ci];
            int lds = h_lds(bn, na, g.th1, g.tw1, d), pieces = h_pieces(g.th1, g.tw1, d);
            long patch = (long)(g.th1 + 2 * d) * (g.tw1 + 2 * d);
            if (g.th2) {
                lds = lds > h_lds(bn, na, g.th2, g.tw2, d) ? lds : h_lds(bn, na, g.th2, g.tw2, d);
                pieces = pieces > h_pieces(g.th2, g.tw2, d) ? pieces : h_pieces(g.th2, g.tw2, d);
            }
            if (lds > 160 * 1024 || pieces > 72) continue;  // (72 = 9 tap slots x 8 waves of patch pieces)
            // two workgroups share a CU only in the BN = 128 form with one patch image (<= 128 VGPRs) and <= 80 KB of LDS
            const long per_cu = (bn <= 128 && na == 1 && lds <= 80 * 1024) ? 2 : 1;
            const long wgs = (long)g.mtiles * ntiles, slots = 256L * per_cu;
            // one patch image costs a barrier + an exposed patch load per channel chunk: ~10 % of a round
            const long cost[3] = {(wgs + slots - 1) / slots * ((na == 1 && kchunks > 1) ? 11 : 10), g.mtiles, patch};
            bool better = best.na == 0;
            for (int k = 0; k < 3 && !better; k++) {
                if (cost[k] < best_cost[k]) better = true;
                else if (cost[k] > best_cost[k]) break;
            }
            if (better) {
                best.g = g; best.na = na; best.lds = lds;
                for (int k = 0; k < 3; k++) best_cost[k] = cost[k];
            }
        }
    return best;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The 4-wave form (round 4, configuration 21): BN = 256 with ONE wave per SIMD and a 128 x 128 wave tile.
//
// Why: with 128 x 64 wave tiles (every 8-wave form of this file and of conv_igemm_kernel.h) a 64-deep K step reads 192 KB of
// fragments from LDS for 2,048 cycles of MFMA per SIMD -- with the DMA writes that is all the LDS can deliver, and the K loop sits
// at 73 % of MFMA issue (LAB_NOTES).  A 128 x 128 wave tile reads 128 KB for the same MFMAs (16 per fragment octet instead of 8),
// but its 256 accumulators leave room for only one wave per SIMD: nothing hides a wave's own latencies, so the schedule has to.
// The resident patch is what makes that possible -- no activation gather in the K loop, a tap is an LDS address offset:
//   * weight steps of half a tap (64-byte rows, 16 KB images) in a ring of THREE, the image of step s + 3 issued during step s
//     (four slices of 16 MFMAs = 2,048 cycles of slack before it is waited for);
//   * fragments run one 16-wide slice ahead of the MFMAs in two register sets, ACROSS steps: the barrier of a step sits between its
//     two slices -- after it the next image is published (every wave waited for its own pieces with a counted vmcnt) and the current
//     one is free (its last fragments are already in registers), so neither a DMA nor a ds_read latency is ever exposed;
//   * the next channel chunk's patch arrives one piece per wave and step in the first steps of the current chunk (patch images are
//     padded to a multiple of four pieces so that every wave issues the same count and the vmcnt immediates are uniform);
//   * the last three steps re-issue the first steps' weights of a chunk that does not exist (wrapped to chunk 0: valid memory,
//     never read back) instead of branching around the issue -- the counted waits stay the same to the end.
// Same k order, same MFMA operand slots, same epilogue as every other configuration: bit-identical (tests/test_gpu_halo.py).
constexpr int H4_BN = 256, H4_BKB = 64, H4_BIMG = H4_BN * H4_BKB, H4_NB = 3, H4_TMAX = 14;
constexpr int H4_ROWB = 4 * 128 + 16;  // epilogue staging row of a wave: 128 f32 + pad
__host__ __device__ constexpr int h4_slots(int th, int tw, int d) { return (h_pieces(th, tw, d) + 3) / 4; }  // patch pieces per wave
__host__ __device__ constexpr int h4_pimg(int th, int tw, int d) { return h4_slots(th, tw, d) * 4096; }
__host__ __device__ constexpr int h4_lds(int th, int tw, int d) {
    const int operands = 2 * h4_pimg(th, tw, d) + H4_NB * H4_BIMG;
    const int staging = 4 * 32 * H4_ROWB;
    return operands > staging ? operands : staging;
}

template <int ABL>
__global__ void __launch_bounds__(256, 1) conv3x3_halo4_kernel(const ConvArgs a, const HaloGeom g, const int ntiles) {
    constexpr int TM = 4, TN = 4, B_IT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    int tile;
    {
        const int nblk = g.mtiles * ntiles;
        const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
        const int q = nblk >> 3, rem = nblk & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    }
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const bool r2 = mt >= g.n1;
    const int th = r2 ? g.th2 : g.th1, tw = r2 ? g.tw2 : g.tw1, tiles_x = r2 ? g.tiles_x2 : g.tiles_x1;
    const int mloc = r2 ? mt - g.n1 : mt;
    const int tyb = mloc / tiles_x, txb = mloc - tyb * tiles_x;
    const int y0 = (r2 ? g.ysplit : 0) + tyb * th, x0 = txb * tw, n0 = nt * H4_BN;
    const int d = a.dil, PW = tw + 2 * d, P = PW * (th + 2 * d);
    const int T = ((P + 7) / 8 + 3) / 4;  // patch pieces per wave (workgroup-uniform)
    const int pimg = T * 4096;
    const int npix = th * tw;
    const float rtw = 1.0f / (float)tw;
    const int Kb = a.Cin * 2;
    const int cchunks = a.Cin / 64;

    char* const As = smem;             // [2][T * 32 rows][128]
    char* const Bs = smem + 2 * pimg;  // [3][256][64]
    const unsigned lds0 = (unsigned)(size_t)(lds_void_h*)smem;
    const u32x4h in_v = h_rsrc(a.in, (unsigned)((size_t)a.H * a.W * Kb));
    const u32x4h wt_v = h_rsrc(a.wt, (unsigned)((size_t)a.Cout * 9 * Kb));

    // patch piece of slot t: piece 4 t + wave, lane l brings row p = 8 (4 t + wave) + (l >> 3) (rows >= P: zeros into the padding)
    unsigned a_voff[H4_TMAX];
#pragma unroll
    for (int t = 0; t < H4_TMAX; t++) {
        const int p = 8 * (4 * t + wave) + (lane >> 3);
        const int py = p / PW, px = p - py * PW;
        const int iy = y0 - d + py, ix = x0 - d + px;
        const bool ok = p < P && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        a_voff[t] = ok ? (unsigned)(iy * a.W + ix) * (unsigned)Kb + (unsigned)(((lane & 7) ^ h_swz(p)) * 16) : HOOB;
    }
    unsigned b_voff[B_IT];  // weight piece i of this wave: rows 16 (4 wave + i) .. + 15 of the 256 x 64 B image
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        const int row = 16 * (wave * B_IT + i) + (lane >> 2);
        const int n = n0 + row;
        b_voff[i] = n < a.Cout ? (unsigned)n * (unsigned)(9 * Kb) + (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16) : HOOB;
    }
    auto dma_a = [&](const int t, const int cc, const int img) {
        h_dma16(in_v, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(img * pimg + (4 * t + wave) * 1024)), a_voff[t],
                __builtin_amdgcn_readfirstlane((unsigned)(cc * 128)));
    };
    auto dma_b = [&](const int cc, const int q, const int img) {  // weight step q = 2 tap + half of channel chunk cc
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(((q >> 1) * cchunks + cc) * 128 + (q & 1) * H4_BKB));
#pragma unroll
        for (int i = 0; i < B_IT; i++)
            h_dma16(wt_v, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(2 * pimg + img * H4_BIMG + (wave * B_IT + i) * 1024)), b_voff[i], soff);
    };

    int pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int R = (wm * TM + i) * 32 + h_pix(r, tw);
        const int ty = (int)(((float)R + 0.5f) * rtw), tx = R - ty * tw;
        pbase[i] = R < npix ? ty * PW + tx : 0;
    }
    const char* const b_lane = Bs + (wn * 128 + r) * H4_BKB;
    const int b_sw = (r >> 2) & 3;

    f32x16h acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // ---- prologue: the patch of chunk 0 and the first three weight steps ----
    for (int t = 0; t < T; t++) dma_a(t, 0, 0);
    dma_b(0, 0, 0);
    dma_b(0, 1, 1);
    // In the K loop nothing travels by LDS-DMA: an LDS-DMA instruction holds its wave at issue for 50-60 cycles, an MFMA covers 32, and
    // with one wave per SIMD nobody else feeds the pipe meanwhile (doubling the pieces of the first cut cost 24 %).  A plain 16-byte
    // buffer load and, two steps later, a ds_write_b128 of the same registers each issue inside one MFMA's shadow: weight step s + 4 is
    // loaded during step s into register set s & 1, written to the ring slot of step s + 2 during step s + 2 (free since that step's
    // barrier), published by the barrier of step s + 3 and read from then on -- two whole steps of latency budget, no counted waits
    // (the compiler tracks ordinary loads itself).
    const auto wt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), 0, (unsigned)((size_t)a.Cout * 9 * Kb), 0x00020000);
    const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (unsigned)((size_t)a.H * a.W * Kb), 0x00020000);
    u32x4h wreg[2][B_IT], preg = {0u, 0u, 0u, 0u};
    auto w_soff = [&](const int cc, const int q) { return (unsigned)(((q >> 1) * cchunks + cc) * 128 + (q & 1) * H4_BKB); };
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        wreg[0][i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rs, b_voff[i], w_soff(0, 2), 0);
        wreg[1][i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rs, b_voff[i], w_soff(0, 3), 0);
    }
    char* const w_dst = smem + 2 * pimg + wave * B_IT * 1024 + lane * 16;  // a piece lands lane-linear, as the DMA's did
    char* const p_dst = smem + wave * 1024 + lane * 16;

    // fragment addresses of all nine taps, kept for the whole kernel (36 VGPRs: with one wave per SIMD there is room, and the K loop
    // loses the ~20 VALU instructions per tap that sat in ONE MFMA gap): chunk (2 sl + hh) of patch row p sits at position
    // (2 sl + hh) ^ swz(p) = (sl << 1) ^ (hh ^ swz(p)); the patch image's offset is added by the same v_xad_u32 that applies the slice
    unsigned abase9[9][TM];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int p = pbase[i] + ((t / 3) * PW + (t % 3)) * d;
            abase9[t][i] = (unsigned)(p * 128) | (unsigned)((hh ^ h_swz(p)) << 4);
        }
    // (all of the above needs no LDS data and runs while the prologue's pieces are in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // MFMAs k0 .. k1 - 1 of a slice's sixteen (k = 4 i + j: consecutive MFMAs write different accumulators)
    auto mma = [&](const h16x8h (&fa)[TM], const h16x8h (&fb)[TN], const int k0, const int k1) __attribute__((always_inline)) {
#pragma unroll
        for (int k = k0; k < k1; k++)
            acc[k >> 2][k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[k & 3], fa[k >> 2], acc[k >> 2][k & 3], 0, 0, 0);
    };
    // (the image offset is added FIRST and the slice XORed in after it -- the two touch disjoint bits, so the order is free, but
    //  `(abase9 ^ slice)` alone is loop-invariant and the compiler would keep all 144 of them in registers: 30 spills)
    unsigned acur[TM];  // abase9[current tap] + offset of the current patch image
    auto read_a = [&](const int sl, h16x8h (&fa)[TM]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const h16x8h*>(As + (acur[i] ^ (unsigned)(sl << 5)));
    };
    auto read_b = [&](const int bs, const int kl, h16x8h (&fb)[TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const h16x8h*>(b_lane + bs * H4_BIMG + j * 32 * H4_BKB + (((2 * kl + hh) ^ b_sw) * 16));
    };
// (with one wave per SIMD the ORDER is the schedule: an LDS-DMA holds its wave at issue for tens of cycles, a clump of five of them
//  right after the barrier left the matrix pipe idle for most of that time -- every piece now sits alone between two MFMA pairs, the
//  fragment reads likewise, and sched_barrier pins it)
#define H4_PIN __builtin_amdgcn_sched_barrier(0)

    h16x8h fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) acur[i] = abase9[0][i];
    read_a(0, fa0);
    read_b(0, 0, fb0);
    for (int cc = 0; cc < cchunks; cc++) {
        const int aimg = cc & 1;
        const bool more_chunks = cc + 1 < cchunks;
        const int ccn = more_chunks ? cc + 1 : 0;
        const unsigned aoff = (unsigned)(aimg * pimg), aoff_n = (unsigned)((aimg ^ 1) * pimg);
        // (the eighteen steps are instantiated one by one: `#pragma unroll` gives up on a body with pinned scheduling regions, and a
        //  run-time q turns the slot / tap arithmetic and a_voff[q] into indexed register accesses -- 111 spilled VGPRs)
        auto step = [&](auto QC) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::value;
            constexpr int half = q & 1;
            // ---- slice 0 of step q: the fragments of slice 1 (same tap, same weight image) are read while set 0 is multiplied: ONE
            //      ds_read_b128 behind each of the first eight MFMAs (a read holds the wave ~16 cycles at issue, an MFMA covers 32; four
            //      reads in a row behind a pair of MFMAs left the pipe idle for half of them: 15 % of the kernel, INFUR_H4_ABL) ----
#define H4_RA(fa, sl, i) if (!(ABL & 4)) fa[i] = *reinterpret_cast<const h16x8h*>(As + (acur[i] ^ (unsigned)((sl) << 5)))
#define H4_RB(fb, bs, kl, j) if (!(ABL & 4)) fb[j] = *reinterpret_cast<const h16x8h*>(b_lane + (bs) * H4_BIMG + (j) * 32 * H4_BKB + (((2 * (kl) + hh) ^ b_sw) * 16))
#define H4_M(fa, fb, k) acc[(k) >> 2][(k) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[(k) & 3], fa[(k) >> 2], acc[(k) >> 2][(k) & 3], 0, 0, 0)
            H4_M(fa0, fb0, 0); H4_PIN; H4_RA(fa1, 2 * half + 1, 0); H4_PIN;
            H4_M(fa0, fb0, 1); H4_PIN; H4_RA(fa1, 2 * half + 1, 1); H4_PIN;
            H4_M(fa0, fb0, 2); H4_PIN; H4_RA(fa1, 2 * half + 1, 2); H4_PIN;
            H4_M(fa0, fb0, 3); H4_PIN; H4_RA(fa1, 2 * half + 1, 3); H4_PIN;
            H4_M(fa0, fb0, 4); H4_PIN; H4_RB(fb1, q % 3, 1, 0); H4_PIN;
            H4_M(fa0, fb0, 5); H4_PIN; H4_RB(fb1, q % 3, 1, 1); H4_PIN;
            H4_M(fa0, fb0, 6); H4_PIN; H4_RB(fb1, q % 3, 1, 2); H4_PIN;
            H4_M(fa0, fb0, 7); H4_PIN; H4_RB(fb1, q % 3, 1, 3); H4_PIN;
            // the next chunk's patch, one piece per wave and step: written here one step after it was asked for
            H4_M(fa0, fb0, 8); H4_PIN;
            if constexpr (q >= 1 && q - 1 < H4_TMAX)  // (steps 15 .. 17 never carry a patch piece: T <= H4_TMAX, enforced by the host)
                if (!(ABL & 1) && q - 1 < T && more_chunks) *reinterpret_cast<u32x4h*>(p_dst + (aimg ^ 1) * pimg + (q - 1) * 4096) = preg;
            H4_PIN;
            H4_M(fa0, fb0, 9); H4_PIN;
            if constexpr (q < H4_TMAX)  // (a_voff has H4_TMAX entries: no instantiation indexes past it)
                if (!(ABL & 1) && q < T && more_chunks) preg = __builtin_amdgcn_raw_buffer_load_b128(in_rs, a_voff[q], (unsigned)((cc + 1) * 128), 0);
            H4_PIN;
            mma(fa0, fb0, 10, 16);
            // ---- image q + 1 is published (written during step q - 1, before this barrier), image q is free ----
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
            // ---- slice 1: the next step's first fragments; weight step q + 2 from its registers into the ring; weight step q + 4 into them ----
            constexpr int qb = q + 4 >= 18 ? q + 4 - 18 : q + 4;
            const int cb = q + 4 >= 18 ? ccn : cc;
            constexpr int tn = q == 17 ? 0 : (q + 1) >> 1;  // the next step's tap (of the next chunk after the ninth)
            const unsigned ao_n = q == 17 ? aoff_n : aoff;
#define H4_AC(i) if (half == 1) acur[i] = abase9[tn][i] + ao_n
            H4_M(fa1, fb1, 0); H4_PIN; H4_AC(0); H4_RA(fa0, half == 1 ? 0 : 2, 0); H4_PIN;
            H4_M(fa1, fb1, 1); H4_PIN; H4_AC(1); H4_RA(fa0, half == 1 ? 0 : 2, 1); H4_PIN;
            H4_M(fa1, fb1, 2); H4_PIN; H4_AC(2); H4_RA(fa0, half == 1 ? 0 : 2, 2); H4_PIN;
            H4_M(fa1, fb1, 3); H4_PIN; H4_AC(3); H4_RA(fa0, half == 1 ? 0 : 2, 3); H4_PIN;
#undef H4_AC
            H4_M(fa1, fb1, 4); H4_PIN; H4_RB(fb0, (q + 1) % 3, 0, 0); H4_PIN;
            H4_M(fa1, fb1, 5); H4_PIN; H4_RB(fb0, (q + 1) % 3, 0, 1); H4_PIN;
            H4_M(fa1, fb1, 6); H4_PIN; H4_RB(fb0, (q + 1) % 3, 0, 2); H4_PIN;
            H4_M(fa1, fb1, 7); H4_PIN; H4_RB(fb0, (q + 1) % 3, 0, 3); H4_PIN;
#define H4_W(i) if (!(ABL & 1)) *reinterpret_cast<u32x4h*>(w_dst + ((q + 2) % 3) * H4_BIMG + (i) * 1024) = wreg[q & 1][i]
#define H4_L(i) if (!(ABL & 1)) wreg[q & 1][i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rs, b_voff[i], w_soff(cb, qb), 0)
            H4_M(fa1, fb1, 8); H4_PIN; H4_W(0); H4_PIN;
            H4_M(fa1, fb1, 9); H4_PIN; H4_W(1); H4_PIN;
            H4_M(fa1, fb1, 10); H4_PIN; H4_W(2); H4_PIN;
            H4_M(fa1, fb1, 11); H4_PIN; H4_W(3); H4_PIN;
            H4_M(fa1, fb1, 12); H4_PIN; H4_L(0); H4_PIN;
            H4_M(fa1, fb1, 13); H4_PIN; H4_L(1); H4_PIN;
            H4_M(fa1, fb1, 14); H4_PIN; H4_L(2); H4_PIN;
            H4_M(fa1, fb1, 15); H4_PIN; H4_L(3); H4_PIN;
#undef H4_W
#undef H4_L
#undef H4_RA
#undef H4_RB
#undef H4_M
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
        step(std::integral_constant<int, 9>{});
        step(std::integral_constant<int, 10>{});
        step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{});
        step(std::integral_constant<int, 13>{});
        step(std::integral_constant<int, 14>{});
        step(std::integral_constant<int, 15>{});
        step(std::integral_constant<int, 16>{});
        step(std::integral_constant<int, 17>{});
    }
#undef H4_PIN
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: + bias, ReLU, f16 (the steps per value of every other configuration); 16 lanes store the 256 contiguous bytes
    //      of a pixel's 128 channels ----
    if (ABL & 32) {  // (ablation: a sum over the accumulators and one conditional store instead of the epilogue)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) sum += acc[i][j][e];
        if (sum == 12345.f) static_cast<_Float16*>(a.out)[tid] = (_Float16)1.f;
        return;
    }
    char* stage = smem + wave * 32 * H4_ROWB;
    const int e_row = lane >> 4, e_col = lane & 15;
    const int n = n0 + wn * 128 + e_col * 8;
    const bool n_ok = n < a.Cout;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bv2 = bv;
    if (a.bias && n_ok) {
        bv = *reinterpret_cast<const float4*>(a.bias + n);
        bv2 = *reinterpret_cast<const float4*>(a.bias + n + 4);
    }
    _Float16* out = static_cast<_Float16*>(a.out);
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int jj = 0; jj < TN; jj++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const float4 v = make_float4(acc[i][jj][4 * gq + 0], acc[i][jj][4 * gq + 1], acc[i][jj][4 * gq + 2], acc[i][jj][4 * gq + 3]);
                *reinterpret_cast<float4*>(stage + h_pix(r, tw) * H4_ROWB + (jj * 32 + 8 * gq + 4 * hh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + e_row;
            const int R = (wm * TM + i) * 32 + row;
            const int ty = (int)(((float)R + 0.5f) * rtw), tx = R - ty * tw;
            const int oy = y0 + ty, ox = x0 + tx;
            if (R < npix && oy < a.OH && ox < a.OW && n_ok) {
                const float4 v = *reinterpret_cast<const float4*>(stage + row * H4_ROWB + e_col * 32);
                const float4 v2 = *reinterpret_cast<const float4*>(stage + row * H4_ROWB + e_col * 32 + 16);
                float x[8] = {v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w, v2.x + bv2.x, v2.y + bv2.y, v2.z + bv2.z, v2.w + bv2.w};
                if (a.relu) {
#pragma unroll
                    for (int t = 0; t < 8; t++) x[t] = fmaxf(x[t], 0.f);
                }
                const h16x8h hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3],
                                   (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
                if (!(ABL & 16)) *reinterpret_cast<h16x8h*>(out + ((size_t)oy * a.OW + ox) * a.Cout + n) = hv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// tiling for the 4-wave form: the candidates of halo_plan, two patch images always, one workgroup per CU
HaloPlan halo4_plan(const ConvArgs& a) {
    HaloPlan best;
    long best_cost[3] = {0, 0, 0};
    const int ntiles = a.Cout / H4_BN, d = a.dil, H = a.OH, W = a.OW;
    auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
    HaloGeom cands[3];
    int nc = 0;
    cands[nc++] = HaloGeom{16, 16, cdiv(W, 16), cdiv(H, 16) * cdiv(W, 16), H, 0, 0, 0, cdiv(H, 16) * cdiv(W, 16)};
    cands[nc++] = HaloGeom{8, 32, cdiv(W, 32), cdiv(H, 8) * cdiv(W, 32), H, 0, 0, 0, cdiv(H, 8) * cdiv(W, 32)};
    if (H >= 16 && H % 16 != 0 && H % 16 <= 8) {
        const int n1 = (H / 16) * cdiv(W, 16), n2 = cdiv(W, 32);
        cands[nc++] = HaloGeom{16, 16, cdiv(W, 16), n1, (H / 16) * 16, H % 16, 32, n2, n1 + n2};
    }
    for (int ci = 0; ci < nc; ci++) {
        const HaloGeom& g = cands[ci];
        int lds = h4_lds(g.th1, g.tw1, d), slots = h4_slots(g.th1, g.tw1, d);
        const long patch = (long)(g.th1 + 2 * d) * (g.tw1 + 2 * d);
        if (g.th2) {
            lds = lds > h4_lds(g.th2, g.tw2, d) ? lds : h4_lds(g.th2, g.tw2, d);
            slots = slots > h4_slots(g.th2, g.tw2, d) ? slots : h4_slots(g.th2, g.tw2, d);
        }
        if (lds > 160 * 1024 || slots > H4_TMAX) continue;
        const long wgs = (long)g.mtiles * ntiles;
        const long cost[3] = {(wgs + 255) / 256, g.mtiles, patch};
        bool better = best.na == 0;
        for (int k = 0; k < 3 && !better; k++) {
            if (cost[k] < best_cost[k]) better = true;
            else if (cost[k] > best_cost[k]) break;
        }
        if (better) {
            best.g = g; best.na = 2; best.lds = lds;
            for (int k = 0; k < 3; k++) best_cost[k] = cost[k];
        }
    }
    return best;
}

template <int BN, int NA, bool I8 = false>
hipError_t launch_halo(const ConvArgs& a, const HaloPlan& pl, hipStream_t s) {
    const int ntiles = a.Cout / BN;
    auto k = conv3x3_halo_kernel<BN, NA, I8>;
    // (the LDS size depends on the dilation and the tile shape: the attribute is raised to the largest footprint there is)
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(pl.g.mtiles * ntiles), dim3(512), pl.lds, s, a, pl.g, ntiles);
    return hipGetLastError();
}

}  // namespace

// configuration 19 asks for BN = 128; it runs with N tiles of 64 (8 waves x 32 pixels, two workgroups per CU) on the 64-channel convs
// of layer1 and wherever 128-wide tiles would leave more than a third of the CUs without a workgroup (layer2 conv2 at 1080p: 128
// tiles x ONE N tile)
static int halo_bn(const ConvArgs& a, int bn) {
    if (bn != 128) return bn;
    if (a.Cout == 64) return 64;
    static const int force = getenv("INFUR_HALO_BN") ? atoi(getenv("INFUR_HALO_BN")) : 0;  // measurement hook: 64 / 128
    if (force == 64 && a.Cout % 64 == 0) return 64;
    if (force == 128) return 128;
    const long tiles = (long)((a.OH + 15) / 16) * ((a.OW + 15) / 16);
    return (a.Cout % 128 == 0 && tiles * (a.Cout / 128) <= 170) ? 64 : 128;
}

bool conv3x3_halo_valid(const ConvArgs& a, int mode, int out_f32, int bn) {
    if (mode == 4) {  // the quantised model's 3x3 convs (u8 out): BN = 128 / 256, 128-byte rows = 128 channels
        if (out_f32 || (bn != 128 && bn != 256) || !a.q_mult || !a.q_bias) return false;
        if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != a.dil || (a.dil != 1 && a.dil != 2 && a.dil != 4)) return false;
        if (a.res || a.in2 || a.batch > 1 || a.OH != a.H || a.OW != a.W) return false;
        if (a.Cin % 128 != 0 || a.Cout % bn != 0) return false;
        if (halo_plan(a, bn, 1).na == 0) return false;
        return (size_t)a.H * a.W * a.Cin < 0x80000000ull && (size_t)a.Cout * 9 * a.Cin < 0x80000000ull;
    }
    bn = halo_bn(a, bn);
    if (mode != 1 || out_f32 || (bn != 64 && bn != 128 && bn != 256)) return false;
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != a.dil || (a.dil != 1 && a.dil != 2 && a.dil != 4)) return false;
    if (a.res || a.in2 || a.batch > 1 || a.OH != a.H || a.OW != a.W) return false;
    if (a.Cin % 64 != 0 || a.Cout % bn != 0 || (a.Cout & 7)) return false;
    if (halo_plan(a, bn).na == 0) return false;
    // 32-bit buffer offsets with 0x80000000 (+ the channel chunk's scalar offset) as the out-of-range marker
    return (size_t)a.H * a.W * a.Cin * 2 < 0x80000000ull && (size_t)a.Cout * 9 * a.Cin * 2 < 0x80000000ull;
}

hipError_t launch_conv3x3_halo_q(const ConvArgs& a, int bn, hipStream_t s) {
    if (!conv3x3_halo_valid(a, 4, 0, bn)) return hipErrorInvalidValue;
    const HaloPlan pl = halo_plan(a, bn, 1);
    if (bn == 128) return pl.na == 2 ? launch_halo<128, 2, true>(a, pl, s) : launch_halo<128, 1, true>(a, pl, s);
    return pl.na == 2 ? launch_halo<256, 2, true>(a, pl, s) : launch_halo<256, 1, true>(a, pl, s);
}

hipError_t launch_conv3x3_halo(const ConvArgs& a, int bn, hipStream_t s) {
    if (!conv3x3_halo_valid(a, 1, 0, bn)) return hipErrorInvalidValue;
    bn = halo_bn(a, bn);
    const HaloPlan pl = halo_plan(a, bn);
    if (bn == 64) return launch_halo<64, 1>(a, pl, s);
    if (bn == 128) return pl.na == 2 ? launch_halo<128, 2>(a, pl, s) : launch_halo<128, 1>(a, pl, s);
    return pl.na == 2 ? launch_halo<256, 2>(a, pl, s) : launch_halo<256, 1>(a, pl, s);
}


bool conv3x3_halo4_valid(const ConvArgs& a, int mode, int out_f32) {
    if (mode != 1 || out_f32) return false;
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != a.dil || (a.dil != 1 && a.dil != 2)) return false;
    if (a.res || a.in2 || a.batch > 1 || a.OH != a.H || a.OW != a.W) return false;
    if (a.Cin % 64 != 0 || a.Cout % H4_BN != 0) return false;
    if (halo4_plan(a).na == 0) return false;
    return (size_t)a.H * a.W * a.Cin * 2 < 0x80000000ull && (size_t)a.Cout * 9 * a.Cin * 2 < 0x80000000ull;
}

hipError_t launch_conv3x3_halo4(const ConvArgs& a, hipStream_t s) {
    if (!conv3x3_halo4_valid(a, 1, 0)) return hipErrorInvalidValue;
    const HaloPlan pl = halo4_plan(a);
    const int ntiles = a.Cout / H4_BN;
    // timing ablations (results WRONG; instrumentation build only: make EXTRA=-DH4_ABLATIONS, then INFUR_H4_ABL = 1 no staging in the K
    // loop, 2 no barrier, 4 no fragment reads, 7 all three, 16 no stores, 32 no epilogue)
#ifdef H4_ABLATIONS
    static const int abl = getenv("INFUR_H4_ABL") ? atoi(getenv("INFUR_H4_ABL")) : 0;
    auto k = abl == 1 ? conv3x3_halo4_kernel<1> : abl == 2 ? conv3x3_halo4_kernel<2> : abl == 4 ? conv3x3_halo4_kernel<4>
             : abl == 7 ? conv3x3_halo4_kernel<7> : abl == 16 ? conv3x3_halo4_kernel<16> : abl == 32 ? conv3x3_halo4_kernel<32> : conv3x3_halo4_kernel<0>;
#else
    auto k = conv3x3_halo4_kernel<0>;
#endif
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(pl.g.mtiles * ntiles), dim3(256), pl.lds, s, a, pl.g, ntiles);
    return hipGetLastError();
}

}  // namespace infur
