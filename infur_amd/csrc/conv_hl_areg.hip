// conv_hl_areg.hip -- INFUR_DTYPE_F16_HL: the 1x1 EXPANSIONS of a bottleneck (conv3: C2 -> 4 C2 channels + residual + ReLU, short
// reduction, many output channels) with the ACTIVATION FRAGMENT IN REGISTERS and everything else streamed.
//
// Replaces the Conv / Add / Relu nodes ONNX Runtime executes inside `session.run` (infur/src/predict_onnx.rs:138), like conv_hl.hip.
//
// Why (profiles/r05_f16hl_ablation_one_step.log): in the tiled kernel a conv3 tile moves as many bytes in
// its epilogue (residual in, output out: HBM) as its K loop ingests (L2 -> LDS), the two phases use different parts of the memory
// system and, with one 144-KB workgroup per CU, they run one after the other: layer3 conv3 at 1080p takes 70 us against 39 us of
// HBM time.  Here a workgroup owns 128 pixels for ALL output channels:
//   * each of its 8 waves (4 along the pixels x 2 along the channels) loads its 32 pixels x Cin activation fragment ONCE, straight
//     into the MFMA operand layout (12 VGPRs per 32 channels: two hi chunks and one lo chunk per lane), and keeps it;
//   * the weight matrix streams through LDS in 128-channel tiles, two 32-channel K steps per ring image (24 KB), by LDS-DMA;
//   * the residual tile of the NEXT 128 channels (32 pixels x 64 channels per wave, hi and lo planes) arrives by LDS-DMA in the
//     wave's private slot while the current tile multiplies; the epilogue adds it in the ACCUMULATOR layout, writes the new hi / lo
//     bytes over it (in place) and stores whole 128-byte / 64-byte pixel rows from there -- no f32 staging, no residual registers,
//     and the stores of tile t drain under the MFMAs of tile t + 1.
// The rows of a weight tile are staged in a permuted order (bits 2 and 3 of the row swapped: conv1x1_b2b.hip) so that a lane's
// accumulator registers 8 s .. 8 s + 7 are 8 CONSECUTIVE channels: one ds_read_b128 / ds_write_b128 per 8 values.
//
// Same arithmetic as every configuration of conv_hl_kernel: per 32-channel K step two f16 MFMAs and one bf8 cross-term MFMA on the
// same operand slots, K ascending, then * acc_scale, + bias, + (residual hi + lo), ReLU, split -- bit-identical
// (tests/test_gpu_hl.py::test_hl_tile_configurations_are_bit_identical), so the tuner picks it per layer shape like any other form.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "conv_hl_dev.h"

namespace infur {

#ifdef AH_TRACE
// Instrumentation build (make EXTRA=-DAH_TRACE build/conv_hl_areg.o; INFUR_AH_TRACE=1): workgroup 40, waves 0 and NW - 3, stamp
// s_memtime at every phase boundary; the second launch of the largest form prints the differences (LAB_NOTES.md, round 5).
__device__ unsigned long long g_ah_trace[2 * 512];
#define AH_T(tag) do { if (tr_on && tr_n < 510) { g_ah_trace[2 * tr_n] = (unsigned long long)(tag); g_ah_trace[2 * tr_n + 1] = __builtin_amdgcn_s_memtime(); tr_n++; } } while (0)
#else
#define AH_T(tag) do { } while (0)
#endif

namespace {

constexpr int AH_BM = 128, AH_BN = 128;
constexpr int AH_SUB = AH_BN * 96;      // one 32-channel K step of a weight tile: 128 rows x 64 B (hi) + 128 rows x 32 B (lo)
constexpr int AH_SUB_LO = AH_BN * 64;   // offset of the lo rows inside it
constexpr int AH_IMG = 2 * AH_SUB;      // ring image: two K steps
constexpr int AH_SLOTS = 128 * 128 * 3; // the waves' residual / output slots together: 128 pixels x 128 channels x 3 bytes
constexpr int AH_MAX_COUT = 2048;       // bias table
__host__ __device__ constexpr int ah_lds_bytes(int nimg) { return nimg * AH_IMG + AH_SLOTS + AH_MAX_COUT * 4; }

__host__ __device__ constexpr int ah_pi(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }
__host__ __device__ constexpr int ah_swz8(int row) { return (row >> 1) & 7; }  // rows of >= 128 bytes: XOR on the low three chunk bits

// wait until at most n of this wave's vector-memory LOADS are outstanding (n wave-uniform).  Every load of the main loop is an
// LDS-DMA issued through hl_dma16, so the count is ours; rounding n down is safe (a stricter wait).  Stores are counted by the
// hardware as well and may retire out of order with respect to loads: one still in flight makes a wait longer, never too short.
__device__ __forceinline__ void ah_wait_loads(const int n) {
    if (n >= 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    else if (n >= 18) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
    else if (n >= 15) asm volatile("s_waitcnt vmcnt(15) lgkmcnt(0)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (n >= 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if (n >= 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// KS = K steps of 32 channels (Cin = 32 KS, even); NIMG = ring images (NIMG - 1 of them, two K steps each, in flight);
// NW = waves: 8 (4 along the pixels x 2 along the channels, 32 pixels x 64 channels each, two waves per SIMD: Cin <= 256, the
// fragment is <= 96 VGPRs) or 4 (32 pixels x all 128 channels of a tile each, ONE wave per SIMD with the whole register file:
// Cin = 512 -- layer4's expansions -- whose fragment is 192 VGPRs)
template <int KS, int NIMG, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) conv_hl_areg_kernel(const ConvArgs a, const int mtiles) {
    static_assert(NW == 8 || NW == 4, "waves");
    static_assert(KS % 2 == 0 && KS >= 2 && KS <= (NW == 8 ? 8 : 16), "an image holds two K steps; the fragment is 12 KS VGPRs");
    static_assert(NIMG >= 2 && NIMG <= 4, "ring");
    constexpr int NSTEP = KS / 2;
    constexpr int TN = NW == 8 ? 2 : 4;          // 32-channel blocks per wave
    constexpr int WCH = TN * 32;                 // channels per wave
    constexpr int RSLOT = 32 * WCH * 3, RSLOT_LO = 32 * WCH * 2;  // a wave's slot: 32 rows x (2 WCH bytes hi + WCH bytes lo)
    constexpr int HP = 32 * WCH * 2 / 1024, LP = 32 * WCH / 1024, NP = HP + LP;  // 1-KB pieces of the slot: hi, lo
    constexpr int HROWS = 1024 / (WCH * 2), LROWS = 1024 / WCH;                   // rows per piece
    constexpr int HCH = WCH * 2 / 16, LCH = WCH / 16;                             // 16-byte chunks per row
    constexpr int PPI = 8 / NW;                  // hi pieces per wave and K step = lo pieces per wave and image
    constexpr int OFF_R = NIMG * AH_IMG, OFF_BIAS = OFF_R + AH_SLOTS;
    static_assert(NW * RSLOT == AH_SLOTS, "slots");
    hl_set_fp16_ovfl();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = NW == 8 ? wave >> 1 : wave, wn = NW == 8 ? wave & 1 : 0;
    const int r = lane & 31, h = lane >> 5;
    const int M = a.OH * a.OW, K = a.Cin;
    int tile;
    {  // XCD-aware order: block b runs on XCD b % 8; every XCD gets a contiguous run of pixel tiles
        const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
        const int q = mtiles >> 3, rr = mtiles & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    }
    const int m0 = tile * AH_BM;
    const int ntiles = a.Cout / AH_BN;
#ifdef AH_TRACE
    const bool tr_on = blockIdx.x == 40 && (tid == 0 || tid == (NW - 3) * 64);
    int tr_n = tid == 0 ? 0 : 256;
    AH_T(1);
#endif

    // ---- the wave's activation fragment, loaded once: pixel m0 + 32 wm + r, lane half h = channels 16 h .. 16 h + 15 of every K step
    //      (hi chunks 2 h and 2 h + 1, lo chunk h: conv_hl.hip's fragment reads) ----
    uint4 ahi[KS][2], alo[KS];
    {
        const auto ih = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (unsigned)((size_t)a.H * a.W * K * 2), 0x00020000);
        const auto il = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in_lo), 0, (unsigned)((size_t)a.H * a.W * K), 0x00020000);
        const int m = m0 + wm * 32 + r;
        const int oy = m / a.OW, ox = m - oy * a.OW;
        const unsigned pix = (unsigned)(oy * a.stride * a.W + ox * a.stride);
        const unsigned bh = m < M ? pix * (unsigned)(K * 2) + (unsigned)(h * 32) : HL_OOB;
        const unsigned bl = m < M ? pix * (unsigned)K + (unsigned)(h * 16) : HL_OOB;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            ahi[ks][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ih, bh, (unsigned)(ks * 64), 0));
            ahi[ks][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ih, bh, (unsigned)(ks * 64 + 16), 0));
            alo[ks] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(il, bl, (unsigned)(ks * 32), 0));
        }
    }
    // bias table
    {
        float* bt = reinterpret_cast<float*>(smem + OFF_BIAS);
        for (int i = tid; i < a.Cout; i += NW * 64) bt[i] = a.bias ? a.bias[i] : 0.f;
    }

    auto mk = [](const void* p, unsigned bytes) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        u32x4r d;
        d.x = __builtin_amdgcn_readfirstlane((unsigned)v);
        d.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        d.z = __builtin_amdgcn_readfirstlane(bytes);
        d.w = 0x00020000u;
        return d;
    };
    const bool has_res = a.res != nullptr;
    const unsigned out_elems = (unsigned)((size_t)M * a.Cout);  // rows >= M fall outside num_records: loads give zeros, stores are dropped
    const u32x4r bh_v = mk(a.wt, (unsigned)((size_t)a.Cout * K * 2)), bl_v = mk(a.wt_lo, (unsigned)((size_t)a.Cout * K));
    const u32x4r rh_v = mk(has_res ? a.res : a.out, out_elems * 2u), rl_v = mk(has_res ? a.res_lo : a.out_lo, out_elems);
    const auto oh_r = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_elems * 2u, 0x00020000);
    const auto ol_r = __builtin_amdgcn_make_buffer_rsrc(a.out_lo, 0, out_elems, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;

    // ---- weight stream: a ring image (two K steps) is 16 hi pieces (16 rows x 64 B) and 8 lo pieces (32 rows x 32 B); wave v
    //      issues hi pieces v PPI .. v PPI + PPI - 1 of either K step and PPI lo pieces; image row R holds output channel pi(R) ----
    unsigned bh_voff[PPI], bl_voff[PPI], bl_dst[PPI], bl_soff[PPI];
#pragma unroll
    for (int i = 0; i < PPI; i++) {
        const int row = 16 * (wave * PPI + i) + (lane >> 2);
        bh_voff[i] = (unsigned)ah_pi(row) * 64u + (unsigned)(((lane & 3) ^ hl_swz64(row)) * 16);  // (K-block-major weights: conv_hl.hip)
        // lo piece i of this wave: NW = 8: piece wave & 3 of K step wave >> 2; NW = 4: piece `wave` of K step i
        const int lp = NW == 8 ? (wave & 3) : wave, ls = NW == 8 ? (wave >> 2) : i;
        const int rowl = 32 * lp + (lane >> 1);
        bl_voff[i] = (unsigned)ah_pi(rowl) * 32u + (unsigned)(((lane & 1) ^ hl_swz32(rowl)) * 16);
        bl_dst[i] = (unsigned)(ls * AH_SUB + AH_SUB_LO + lp * 1024);
        bl_soff[i] = (unsigned)(ls * a.Cout * 32);
    }
    const int Q = ntiles * NSTEP;  // linear (channel tile, image) counter
    // every workgroup walks the channel tiles cyclically from its own first tile (conv1x1_areg.hip: in lockstep all CUs would pull
    // the same 24 KB from the same few L2 channels at once); the K order inside a tile is unchanged
    const int nt_first = tile % ntiles;
    auto nt_of = [&](int w) { const int t = nt_first + w; return t >= ntiles ? t - ntiles : t; };
    int ld_q = 0, ld_w = 0, ld_st = 0;  // image being loaded: its counter, its channel tile (walk index) and step
    unsigned ld_slot = 0;
    constexpr int DPI = 3 * PPI;  // DMA instructions per wave and image
    auto dma_image = [&]() {
        const int nt = nt_of(ld_w);
        const unsigned img = lds0 + ld_slot;
        const unsigned soh = (unsigned)(2 * ld_st) * (unsigned)(a.Cout * 64) + (unsigned)(nt * AH_BN * 64);  // K step 2 ld_st, rows of tile nt
        const unsigned sol = (unsigned)(2 * ld_st) * (unsigned)(a.Cout * 32) + (unsigned)(nt * AH_BN * 32);
#pragma unroll
        for (int i = 0; i < PPI; i++) {
            const unsigned dst = (unsigned)((wave * PPI + i) * 1024);
            hl_dma16(bh_v, __builtin_amdgcn_readfirstlane(img + dst), bh_voff[i], __builtin_amdgcn_readfirstlane(soh));
            hl_dma16(bh_v, __builtin_amdgcn_readfirstlane(img + AH_SUB + dst), bh_voff[i], __builtin_amdgcn_readfirstlane(soh + (unsigned)(a.Cout * 64)));
            hl_dma16(bl_v, __builtin_amdgcn_readfirstlane(img + bl_dst[i]), bl_voff[i], __builtin_amdgcn_readfirstlane(sol + bl_soff[i]));
        }
        ld_q++;
        ld_st++;
        if (ld_st == NSTEP) {
            ld_st = 0;
            ld_w++;
        }
        ld_slot = ld_slot + AH_IMG == NIMG * AH_IMG ? 0u : ld_slot + AH_IMG;
    };

    // ---- the wave's residual / output slot: 32 hi rows of 2 WCH bytes, 32 lo rows of WCH bytes, the 16-byte chunk index XOR-ed
    //      with ah_swz8(row) (hl_swz64(row) for 64-byte rows).  A DMA piece is 1 KB of whole rows; the same per-lane offsets
    //      address the output stores.  Pieces 0 .. HP - 1 are hi, HP .. NP - 1 lo ----
    const unsigned slot = (unsigned)(OFF_R + wave * RSLOT);
    auto lo_swz = [](int row) { return LCH == 4 ? hl_swz64(row) : ah_swz8(row); };
    unsigned pv[NP];
#pragma unroll
    for (int p = 0; p < HP; p++) {
        const int row = HROWS * p + lane / HCH;
        const unsigned d = (unsigned)((lane % HCH) ^ ah_swz8(row));
        pv[p] = ((unsigned)(m0 + wm * 32 + row) * (unsigned)a.Cout + (unsigned)(wn * WCH) + d * 8u) * 2u;
    }
#pragma unroll
    for (int p = 0; p < LP; p++) {
        const int row = LROWS * p + lane / LCH;
        const unsigned d = (unsigned)((lane % LCH) ^ lo_swz(row));
        pv[HP + p] = (unsigned)(m0 + wm * 32 + row) * (unsigned)a.Cout + (unsigned)(wn * WCH) + d * 16u;
    }
    auto dma_residual = [&](unsigned nb) {
#pragma unroll
        for (int p = 0; p < HP; p++) hl_dma16(rh_v, __builtin_amdgcn_readfirstlane(lds0 + slot + (unsigned)(p * 1024)), pv[p], __builtin_amdgcn_readfirstlane(nb * 2u));
#pragma unroll
        for (int p = 0; p < LP; p++)
            hl_dma16(rl_v, __builtin_amdgcn_readfirstlane(lds0 + slot + (unsigned)(RSLOT_LO + p * 1024)), pv[HP + p], __builtin_amdgcn_readfirstlane(nb));
    };

    // prologue: the first residual tile and the first NIMG images; everything landed before the first step
    if (has_res) dma_residual((unsigned)(nt_of(0) * AH_BN));
    for (int i = 0; i < NIMG && ld_q < Q; i++) dma_image();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    AH_T(2);
    // load bookkeeping (wave-uniform): `issued` counts the DMA instructions since here; mark[j] = its value right after the pieces
    // of image q + 1 + j went out, mark_r after the pieces of the residual tile the next epilogue needs
    int issued = 0, mark_r = 0;
    int mark[NIMG - 1];
#pragma unroll
    for (int j = 0; j < NIMG - 1; j++) mark[j] = 0;

    // fragment addresses inside an image's K step (conv_hl.hip), the wave's WCH rows at wn * WCH
    const int b_hi0 = (wn * WCH + r) * 64 + (((2 * h) ^ hl_swz64(r)) * 16);
    const int b_hi1 = (wn * WCH + r) * 64 + (((2 * h + 1) ^ hl_swz64(r)) * 16);
    const int b_lo = AH_SUB_LO + (wn * WCH + r) * 32 + ((h ^ hl_swz32(r)) * 16);
    const float acc_scale = a.acc_scale;
    unsigned cur = 0;  // byte offset of the image being multiplied

    for (int w = 0; w < ntiles; w++) {
        const unsigned nb = (unsigned)(nt_of(w) * AH_BN);
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[j][e] = 0.0f;
#pragma unroll
        for (int st = 0; st < NSTEP; st++) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int ks = 2 * st + s;
                const char* I = smem + cur + s * AH_SUB;
                uint4 fb0[TN], fb1[TN], fbl[TN];
#pragma unroll
                for (int j = 0; j < TN; j++) fb0[j] = *reinterpret_cast<const uint4*>(I + b_hi0 + j * 32 * 64);
#pragma unroll
                for (int j = 0; j < TN; j++) fb1[j] = *reinterpret_cast<const uint4*>(I + b_hi1 + j * 32 * 64);
#pragma unroll
                for (int j = 0; j < TN; j++) fbl[j] = *reinterpret_cast<const uint4*>(I + b_lo + j * 32 * 32);
                i32x8 a8;
                a8[0] = hl_top4(ahi[ks][0].x, ahi[ks][0].y);
                a8[1] = hl_top4(ahi[ks][0].z, ahi[ks][0].w);
                a8[2] = hl_top4(ahi[ks][1].x, ahi[ks][1].y);
                a8[3] = hl_top4(ahi[ks][1].z, ahi[ks][1].w);
                a8[4] = (int)alo[ks].x; a8[5] = (int)alo[ks].y; a8[6] = (int)alo[ks].z; a8[7] = (int)alo[ks].w;
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb0[j]), __builtin_bit_cast(f16x8, ahi[ks][0]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb1[j]), __builtin_bit_cast(f16x8, ahi[ks][1]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    i32x8 b8;
                    b8[0] = (int)fbl[j].x; b8[1] = (int)fbl[j].y; b8[2] = (int)fbl[j].z; b8[3] = (int)fbl[j].w;
                    b8[4] = hl_top4(fb0[j].x, fb0[j].y);
                    b8[5] = hl_top4(fb0[j].z, fb0[j].w);
                    b8[6] = hl_top4(fb1[j].x, fb1[j].y);
                    b8[7] = hl_top4(fb1[j].z, fb1[j].w);
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[j], 1, 1, 0, 127 - kHlLoShift, 0, 127);
                }
            }
            // end of image q: this wave's pieces of image q + 1 must have landed before the barrier; after it image q is free
            AH_T(10 + st);
            ah_wait_loads(issued - mark[0]);
            AH_T(30 + st);
            __builtin_amdgcn_s_barrier();
            AH_T(50 + st);
            if (ld_q < Q) {
                dma_image();
                issued += DPI;
            }
#pragma unroll
            for (int j = 0; j + 1 < NIMG - 1; j++) mark[j] = mark[j + 1];
            mark[NIMG - 2] = issued;
            cur = cur + AH_IMG == NIMG * AH_IMG ? 0u : cur + AH_IMG;
        }

        // ---- epilogue of channel tile nt_of(w): the wave's 32 pixels x WCH channels.  Accumulator registers 8 s .. 8 s + 7 of block j
        //      are channels 32 j + 16 s + 8 h .. + 7 of pixel r (the row permutation above) ----
        AH_T(70);
        if (has_res) ah_wait_loads(issued - mark_r);
        AH_T(71);
        char* S = smem + slot;
        const float* bt = reinterpret_cast<const float*>(smem + OFF_BIAS) + nb + wn * WCH + 8 * h;
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const float4 b0 = *reinterpret_cast<const float4*>(bt + 32 * j + 16 * s), b1 = *reinterpret_cast<const float4*>(bt + 32 * j + 16 * s + 4);
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                char* ph = S + r * (WCH * 2) + (((4 * j + 2 * s + h) ^ ah_swz8(r)) * 16);
                char* pl = S + RSLOT_LO + r * WCH + (((2 * j + s) ^ lo_swz(r)) * 16) + h * 8;
                float x[8];
#pragma unroll
                for (int t = 0; t < 8; t++) x[t] = __fadd_rn(__fmul_rn(acc[j][8 * s + t], acc_scale), bv[t]);
                if (has_res) {
                    const f16x8 rh = *reinterpret_cast<const f16x8*>(ph);
                    const u32x2 rl = *reinterpret_cast<const u32x2*>(pl);
                    float lo[8];
                    hl_lo8_to_f32(rl.x, lo);
                    hl_lo8_to_f32(rl.y, lo + 4);
#pragma unroll
                    for (int t = 0; t < 8; t++) x[t] += (float)rh[t] + lo[t];
                }
                f16x8 hv;
                u32x2 lv;
                hl_split8(x, hv, lv, a.relu ? 0.f : -kHlHiMax);  // (the ReLU is the split's lower clamp)
                *reinterpret_cast<f16x8*>(ph) = hv;
                *reinterpret_cast<u32x2*>(pl) = lv;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        AH_T(72);
        // whole rows out of the slot: piece p = 1 KB lane-linear (the DMA's layout)
        uint4 y[NP];
#pragma unroll
        for (int p = 0; p < HP; p++) y[p] = *reinterpret_cast<const uint4*>(S + p * 1024 + lane * 16);
#pragma unroll
        for (int p = 0; p < LP; p++) y[HP + p] = *reinterpret_cast<const uint4*>(S + RSLOT_LO + p * 1024 + lane * 16);
#pragma unroll
        for (int p = 0; p < HP; p++) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, y[p]), oh_r, pv[p], nb * 2u, 0);
#pragma unroll
        for (int p = 0; p < LP; p++) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, y[HP + p]), ol_r, pv[HP + p], nb, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (has_res && w + 1 < ntiles) {  // the slot is free again: the next tile's residual
            dma_residual((unsigned)(nt_of(w + 1) * AH_BN));
            issued += NP;
            mark_r = issued;
        }
        AH_T(73);
    }
}

template <int KS, int NIMG, int NW>
hipError_t launch_ah(const ConvArgs& a, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + AH_BM - 1) / AH_BM;
    auto k = conv_hl_areg_kernel<KS, NIMG, NW>;
    constexpr int lds = ah_lds_bytes(NIMG);
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(mtiles), dim3(NW * 64), lds, s, a, mtiles);
#ifdef AH_TRACE
    static int dumps = 0;
    if ((KS == 8 || KS == 16) && getenv("INFUR_AH_TRACE") && dumps < 2) {
        (void)hipStreamSynchronize(s);
        static unsigned long long hbuf[2 * 512];
        (void)hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_ah_trace), sizeof hbuf);
        dumps++;
        if (dumps == 2)
            for (int half = 0; half < 2; half++) {
                fprintf(stderr, "AHTRACE KS %d wave %d:", KS, half ? NW - 3 : 0);
                unsigned long long t0 = hbuf[2 * (half * 256) + 1], prev = t0;
                for (int i = half * 256; i < half * 256 + 255 && hbuf[2 * i]; i++) {
                    fprintf(stderr, " %llu:+%llu", hbuf[2 * i], hbuf[2 * i + 1] - prev);
                    prev = hbuf[2 * i + 1];
                }
                fprintf(stderr, " total %llu\n", prev - t0);
            }
    }
#endif
    return hipGetLastError();
}

}  // namespace

bool conv_hl_areg_valid(const ConvArgs& a, int out_f32) {
    return !out_f32 && a.KH == 1 && a.KW == 1 && a.pad == 0 && !a.in2 && a.batch <= 1 && a.in_lo && a.wt_lo && a.out_lo &&
           (a.res != nullptr) == (a.res_lo != nullptr) && (a.Cin == 64 || a.Cin == 128 || a.Cin == 256 || a.Cin == 512) && a.Cout >= 256 &&
           a.Cout % AH_BN == 0 && a.Cout <= AH_MAX_COUT && !a.acc_scale_b && (size_t)a.H * a.W * a.Cin * 2 < 0x80000000ull &&
           (size_t)a.OH * a.OW * a.Cout * 2 < 0x80000000ull && (size_t)a.Cout * a.Cin * 2 < 0x80000000ull;
}

hipError_t launch_conv_hl_areg(const ConvArgs& a, hipStream_t s) {
    if (!conv_hl_areg_valid(a, 0)) return hipErrorInvalidValue;
    switch (a.Cin) {
        case 64: return launch_ah<2, 4, 8>(a, s);
        case 128: return launch_ah<4, 4, 8>(a, s);
        case 256: return launch_ah<8, 4, 8>(a, s);
        case 512: return launch_ah<16, 4, 4>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace infur
