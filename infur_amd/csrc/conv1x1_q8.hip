// conv1x1_q8.hip -- 1x1 convolutions of QUANTISED models (u8 activations x s8 weights on v_mfma_i32_32x32x32_i8, QLinearConv /
// QLinearAdd epilogue, u8 output) with the activation tile in REGISTERS and NO LDS staging of the output.
//
// Why: a quantised 1x1 convolution has a K loop of one to eight 128-byte steps -- it is all prologue and epilogue.  The tiled
// implicit GEMM (conv_igemm.hip) re-reads the activation tile once per N tile and the weight tile once per M tile (layer3 conv3 at
// 1080p: 195 MB of L1 fills for 75 MB of compulsory bytes) and its epilogue is VALU-bound (14 operations per output with the
// residual sum: qepilogue.h).  A first A-resident port of conv1x1_areg.hip (8 waves, output through per-wave LDS slices: 134 KB of
// LDS, one workgroup per CU) lost to the tiled kernel: nothing overlapped its loads, requantisation and stores.  This form keeps
// the A-resident walk and removes what limited the occupancy:
//   * a workgroup = 4 waves x 32 pixels; each wave loads its 32 x Cin bytes ONCE into the MFMA operand layout (u8 -> s8 by one XOR)
//     and walks all N tiles of 128 channels, the weights streaming through a ring of three 16 KB LDS images by LDS-DMA;
//   * the rows of a weight fragment are read from LDS in a PERMUTED order (MFMA row 8g + 4h + e <- channel 16h + 4g + e), so that in
//     the 32x32 C/D layout a lane's 16 accumulators are 16 CONSECUTIVE output channels of its pixel: requantised and packed they are
//     one 16-byte store, the residual one 16-byte load -- no transposition through LDS, no per-wave slices;
//   * LDS per workgroup = the 48 KB ring + the bias / multiplier tables of the workgroup's OWN output channels (8 bytes each: 52 KB
//     up to 512 channels): three workgroups per CU overlap each other's loads, VALU work and stores;
//   * configuration 18 shares the N tiles of an M tile out over `nsplit` workgroups (conv1x1_q8_nsplit): M = 32400 alone is 254
//     workgroups for 256 CUs;
//   * the accumulators start at the folded bias; the requantisation runs on pairs (qepilogue.h: 9 instead of 14 issues per output
//     with the residual sum, 4 instead of 7 without).
// Integer accumulation has no order and the epilogue arithmetic is qepilogue.h's: bit-identical to the tiled forms
// (tests/test_gpu_quant.py runs every test with this configuration forced, too).
#include <atomic>
#include <cstdlib>

#include "kernels.h"
#include "qepilogue.h"

namespace infur {

typedef int i32x16c __attribute__((ext_vector_type(16)));
typedef int i32x4c __attribute__((ext_vector_type(4)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_c;

namespace {

constexpr unsigned OOBC = 0x80000000u;
constexpr int Q8_BM = 128, Q8_BN = 128;
constexpr int Q8_B_IMG = Q8_BN * 128;  // one weight image: 128 rows x 128 bytes of k
constexpr int Q8_NIMG = 3;
constexpr int Q8_RING = Q8_NIMG * Q8_B_IMG;
constexpr int Q8_MAXC = 2048;  // output channels of one workgroup (its share of Cout) whose bias / multiplier tables may sit behind the ring
// dynamic LDS of a launch: the ring + 8 bytes per output channel of a workgroup's share -- 52 KB up to 512 channels, i.e. THREE
// workgroups per CU (with the tables of all 2048 channels it was 64 KB and two, whatever the register budget said)
constexpr int q8_lds(int share_channels) { return Q8_RING + 8 * share_channels; }

// chunk swizzle of a weight row: the 16 lanes of one ds_read_b128 group read rows {0..7, 16..23} or {8..15, 24..31} (the permuted
// fragment order below), which this function spreads over all 16 (row parity, chunk) bank groups
__host__ __device__ constexpr int q8_swz(int row) { return ((row >> 1) & 3) | (((row >> 4) & 1) << 2); }
// MFMA row r of a 32-row block <- weight row (output channel) q8_pi(r)
__host__ __device__ constexpr int q8_pi(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }

__device__ __forceinline__ void q8_dma16(const u32x4c rsrc, const unsigned lds, const unsigned voff, const unsigned soff) {
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

// KS = K steps of 128 bytes (Cin = 128 * KS channels)
// (three workgroups = three waves per SIMD up to Cin = 256: 170 VGPRs, and 52 KB of LDS each while a workgroup's share of Cout is <= 512;
//  the 64- and 128-register fragments of Cin = 512 / 1024 leave room for two)
template <int KS>
__global__ void __launch_bounds__(256, KS <= 2 ? 3 : 2) conv1x1_q8_kernel(const ConvArgs a, const int mtiles, const int nsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.OH * a.OW;
    const int Kb = a.Cin;  // bytes of a row of A / B
    int tile, nshare;
    {  // XCD-aware order: block b runs on XCD b % 8; every XCD gets a contiguous run of (M tile, N share) pairs, shares fastest:
       // the workgroups that read the same activation tile sit on one XCD and share its L2
        const int groups = mtiles * nsplit;
        const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
        const int q = groups >> 3, r = groups & 7;
        const int g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        tile = g / nsplit;
        nshare = g - tile * nsplit;
    }
    const int m = tile * Q8_BM + wave * 32 + (lane & 31);  // this lane's pixel
    const int hh = lane >> 5;
    // N-split form (configuration 18): the workgroup walks only its share of the N tiles -- nsplit times the workgroups for the
    // shapes whose M alone does not fill the chip (1080p layer3 / layer4: 254 M tiles for 256 CUs x 2-3 resident workgroups)
    const int ntiles = a.Cout / Q8_BN / nsplit;
    const int nt_base = nshare * ntiles;

    // ---- the wave's activation fragments, loaded once: all Cin bytes of pixel m, as s8 = u8 - 128 ----
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (unsigned)((size_t)a.H * a.W * a.Cin), 0x00020000);
    i32x4c areg[KS * 4];
    {
        const int oy = m / a.OW, ox = m - oy * a.OW;  // (1x1, stride s: pixel (oy, ox) reads input pixel (oy * s, ox * s))
        const unsigned base = m < M ? (unsigned)(oy * a.stride * a.W + ox * a.stride) * (unsigned)Kb + (unsigned)hh * 16u : OOBC;
#pragma unroll
        for (int sl = 0; sl < KS * 4; sl++)
            areg[sl] = __builtin_bit_cast(i32x4c, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, base, (unsigned)(sl * 32), 0)) ^ (int)0x80808080;
    }

    // ---- weight stream: LDS-DMA, 8 whole rows per wave instruction (4 instructions per wave and K step), chunk index
    //      XOR-swizzled on the source side ----
    u32x4c wt_v;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(a.wt);
        wt_v.x = __builtin_amdgcn_readfirstlane((unsigned)v);
        wt_v.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        wt_v.z = __builtin_amdgcn_readfirstlane((unsigned)((size_t)a.Cout * Kb));
        wt_v.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_void_c*)smem;
    unsigned b_voff[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        b_voff[i] = (unsigned)row * (unsigned)Kb + (unsigned)(((lane & 7) ^ q8_swz(row)) * 16);
    }
    const int Q = ntiles * KS;  // linear (N tile, K step) counter
    // every workgroup starts its cyclic walk over the N tiles at its own tile (conv1x1_areg.hip: no L2 hot spot)
    const int nt_first = tile % ntiles;
    auto nt_of = [&](int w) { const int t = nt_first + w; return nt_base + (t >= ntiles ? t - ntiles : t); };
    auto dma_step = [&](int q) {
        const int wq = q / KS, ks = q - wq * KS;
        const unsigned img = lds0 + (unsigned)((q % Q8_NIMG) * Q8_B_IMG);
        const unsigned soff = (unsigned)(nt_of(wq) * Q8_BN) * (unsigned)Kb + (unsigned)(ks * 128);
#pragma unroll
        for (int i = 0; i < 4; i++)
            q8_dma16(wt_v, __builtin_amdgcn_readfirstlane(img + (unsigned)((wave * 4 + i) * 1024)), b_voff[i], __builtin_amdgcn_readfirstlane(soff));
    };
    // output and residual through buffer descriptors: a pixel beyond M gets an offset outside num_records (loads return zero,
    // stores are dropped)
    const unsigned out_bytes = (unsigned)((size_t)M * a.Cout);
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_bytes, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res), 0, a.res ? out_bytes : 0u, 0x00020000);
    const bool has_res = a.res != nullptr;
    const unsigned prow_off = m < M ? (unsigned)m * (unsigned)a.Cout + (unsigned)hh * 16u : OOBC;  // + channel offset of the block
    // tile 0's residual is issued here, inside the prologue's one latency window (ablation of layer3 conv3 at 1080p: 13 us with
    // everything but the skeleton removed, + 13.7 requantisation, + 5 stores, + 3.5 MFMA, + 2.3 residual loads = the 38.6 measured --
    // the phases of a workgroup's short life simply add up)
    u32x4c rr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rr[j] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, prow_off, (unsigned)(nt_of(0) * Q8_BN) + 32u * j, 0);
    for (int q = 0; q < Q8_NIMG && q < Q; q++) dma_step(q);
    // folded bias and requantisation multiplier of every output channel -> LDS: the epilogues read them as broadcasts, so that no
    // global load (and no compiler-placed vmcnt wait, which would also drain the DMA pieces and stores in flight) sits between
    // the accumulators and the stores
    // (the workgroup's own channels only; the pointers are biased so that the epilogue indexes them with the absolute channel)
    int* tab_b = reinterpret_cast<int*>(smem + Q8_RING) - nt_base * Q8_BN;
    float* tab_m = reinterpret_cast<float*>(smem + Q8_RING + ntiles * Q8_BN * 4) - nt_base * Q8_BN;
    for (int i = nt_base * (Q8_BN / 4) + tid; i < (nt_base + ntiles) * (Q8_BN / 4); i += 256) {
        reinterpret_cast<int4*>(tab_b)[i] = reinterpret_cast<const int4*>(a.q_bias)[i];
        reinterpret_cast<float4*>(tab_m)[i] = reinterpret_cast<const float4*>(a.q_mult)[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment rows in the permuted order: this lane supplies channel 32 j + pi(lane & 31) of a 32-row block
    const int prow = q8_pi(lane & 31);
    const int b_lds = prow * 128;
    const int b_swz = q8_swz(prow);  // (rows 32 apart share the swizzle)
    const QEpi qe = {(float)a.q_yzp, -(float)a.q_yzp, 255.f - (float)a.q_yzp, a.q_ra, a.q_rb, (float)a.q_bzp, (float)a.q_czp};

    for (int wnt = 0; wnt < ntiles; wnt++) {
        const int nt = nt_of(wnt);
        const unsigned n0 = (unsigned)(nt * Q8_BN);
        if (wnt > 0) {  // the residual of this N tile: issued before the K loop, consumed after it
#pragma unroll
            for (int j = 0; j < 4; j++) rr[j] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, prow_off, n0 + 32u * j, 0);
        }
        // the accumulators start at the folded bias of their channel (an integer sum has no order)
        i32x16c acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) {
                const qi4 b4 = *reinterpret_cast<const qi4*>(tab_b + n0 + 32 * j + 16 * hh + 4 * t4);
#pragma unroll
                for (int t = 0; t < 4; t++) acc[j][4 * t4 + t] = b4[t];
            }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int q = wnt * KS + ks;
            const char* Bb = smem + (q % Q8_NIMG) * Q8_B_IMG + b_lds;
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                i32x4c fb[4];
#pragma unroll
                for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const i32x4c*>(Bb + j * 32 * 128 + (((2 * sl + hh) ^ b_swz) * 16));
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], areg[ks * 4 + sl], acc[j], 0, 0, 0);
            }
            // Before the barrier that ends K step q this wave's pieces of step q + 1 must have landed.  Loads retire in order, so
            // it is enough that at most the loads ISSUED AFTER those pieces are still outstanding: the four pieces of step q + 2
            // (if there is one) and, in the first two steps of an N tile, the tile's four residual loads (issued after the pieces of
            // step q0 + 2, before those of q0 + 3; tile 0's went out in the prologue, but the pieces its steps wait for landed there
            // too).  Stores in between may retire early or late: counting only loads, a store still in flight makes the wait longer,
            // never unsafe (conv1x1_areg.hip has the argument).
            const int allowed = (q + 2 < Q ? 4 : 0) + (ks <= 1 ? 4 : 0);  // (ks is unrolled: a compile-time term; the other is wave-uniform)
            if (allowed == 8)
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (allowed == 4)
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // the image just consumed takes K step q + 3; after the LAST step of an N tile the issue waits until the epilogue has
            // consumed the residual (the compiler's own wait for those loads does not know the DMA pieces and would wait for them too)
            if (ks + 1 < KS && q + Q8_NIMG < Q) dma_step(q + Q8_NIMG);
        }

        // ---- epilogue of this N tile: QLinearConv requantisation (+ the block's QLinearAdd); a lane's 16 accumulators of
        //      block j are channels n0 + 32 j + 16 hh .. + 15 of its pixel ----
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = (int)n0 + 32 * j + 16 * hh;
            u32x4c pk;
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) {
                const qf4 m4 = *reinterpret_cast<const qf4*>(tab_m + n + 4 * t4);
                const qi4 a4 = {acc[j][4 * t4], acc[j][4 * t4 + 1], acc[j][4 * t4 + 2], acc[j][4 * t4 + 3]};
                pk[t4] = has_res ? q_word<true>(a4, m4, rr[j][t4], qe) : q_word<false>(a4, m4, 0u, qe);
            }
            __builtin_amdgcn_raw_buffer_store_b128(pk, out_rsrc, prow_off, n0 + 32u * j, 0);
        }
        {  // the deferred issue of the tile's last K step
            const int q = wnt * KS + KS - 1;
            if (q + Q8_NIMG < Q) dma_step(q + Q8_NIMG);
        }
    }
}

template <int KS>
hipError_t launch_q8(const ConvArgs& a, int nsplit, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + Q8_BM - 1) / Q8_BM;
    auto k = conv1x1_q8_kernel<KS>;
    static std::atomic<bool> attr_done[64];  // > 64 KB of dynamic LDS needs the attribute once per kernel and device
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, q8_lds(Q8_MAXC));
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(mtiles * nsplit), dim3(256), q8_lds(a.Cout / nsplit), s, a, mtiles, nsplit);
    return hipGetLastError();
}

}  // namespace

bool conv1x1_q8_valid(const ConvArgs& a, int mode, int out_f32) {
    return mode == 4 && !out_f32 && a.KH == 1 && a.KW == 1 && a.pad == 0 && !a.in2 && a.batch <= 1 && a.q_mult && a.q_bias &&
           (a.Cin == 128 || a.Cin == 256 || a.Cin == 512 || a.Cin == 1024) && a.Cout >= 128 && a.Cout <= Q8_MAXC && (a.Cout & (Q8_BN - 1)) == 0 &&
           (size_t)a.H * a.W * a.Cin < 0x80000000ull && (size_t)a.OH * a.OW * a.Cout < 0x80000000ull && (size_t)a.Cout * a.Cin < 0x80000000ull;
}

// the N split of configuration 18: the smallest divisor of the N tile count that brings the launch to about two workgroups per CU
// (0: the plain form already has them, or Cout has a single N tile).  Measured at 1080p, M = 32400 = 254 M tiles, split 1 / 2 / 4 / 8:
// layer4 conv3 73.6 / 57.4 / 63.1 / 67.3 us, layer4 downsample 99.5 / 77.5 / 84.0 / 94.2, layer3 downsample 36.6 / 30.3 / 33.8 / 40.8 --
// one round of workgroups that each walk several N tiles beats more, shorter-lived ones (a workgroup's life starts with one full
// memory latency for its activation tile, and a second round of workgroups pays it again with most of the chip idle).
int conv1x1_q8_nsplit(const ConvArgs& a) {
    const int mtiles = (a.OH * a.OW + Q8_BM - 1) / Q8_BM, ntiles = a.Cout / Q8_BN;
    if (mtiles >= 480 || ntiles < 2) return 0;
    static const int forced = getenv("INFUR_Q8_NSPLIT") ? atoi(getenv("INFUR_Q8_NSPLIT")) : 0;  // (experiment hook)
    if (forced > 1 && ntiles % forced == 0) return forced;
    for (int d = 2; d <= ntiles; d++)
        if (ntiles % d == 0 && (mtiles * d >= 480 || d == ntiles)) return d;
    return 0;
}

hipError_t launch_conv1x1_q8(const ConvArgs& a, int nsplit, hipStream_t s) {
    if (nsplit < 1 || (a.Cout / Q8_BN) % nsplit) return hipErrorInvalidValue;
    switch (a.Cin) {
        case 128: return launch_q8<1>(a, nsplit, s);
        case 256: return launch_q8<2>(a, nsplit, s);
        case 512: return launch_q8<4>(a, nsplit, s);
        case 1024: return launch_q8<8>(a, nsplit, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace infur
