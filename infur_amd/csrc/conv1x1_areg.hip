// conv1x1_areg.hip -- 1x1 convolutions with a SHORT reduction (Cin <= 256) and many output channels, f16 operands:
// the activation tile stays in REGISTERS while the workgroup walks all N tiles.
//
// Why: in the f16-rate modes the bottleneck expansions of FCN-ResNet (conv3: 256 -> 1024 channels + residual, 22 of them
// in a ResNet-101) are not MFMA-bound but bound by the L2 -> L1 fill rate: the tiled implicit GEMM (conv_igemm.hip)
// re-reads the activation tile once per N tile (8x) and the weight tile once per M tile; at 4K that is 1.33 GB of L1
// fills per launch for 0.6 GB of compulsory traffic (LAB_NOTES.md 3.3).  Here a workgroup owns 256 pixels: each of its 8
// waves (4 along M x 2 along N) loads its 64 x Cin activation fragment ONCE, straight into the MFMA operand layout (<= 128
// VGPRs), and then the workgroup streams the weight matrix through LDS in
// 128-channel tiles by LDS-DMA (a ring of four 16 KB images: three K steps always in flight), accumulating and writing one
// 256 x 128 output tile after the other.  Activation re-reads disappear, weight re-reads halve (256 instead of 128
// pixels per workgroup) and the activation fragments are never read from LDS at all.
//
// Same arithmetic as every configuration of conv_igemm_kernel (k ascending, v_mfma_f32_32x32x16_f16 with the weight
// fragment as the row operand, f32 accumulation, + bias + residual, ReLU, f16 store): bit-identical results
// (tests/test_gpu_conv_configs.py), so the autotuner may pick it per layer shape like any other configuration.
#include <atomic>
#include <cstdlib>

#include "kernels.h"

namespace infur {

typedef float f32x16a __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8a __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4a __attribute__((ext_vector_type(4)));
typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
typedef unsigned u32x2a __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_a;

namespace {

constexpr unsigned OOBA = 0x80000000u;
constexpr int AR_BN = 128, AR_ROWB = 2 * 128 + 16;  // staged epilogue row: 64 f32 + pad
constexpr int AR_B_IMG = AR_BN * 128;                             // one weight image: 128 rows x 128 bytes (64 k)
constexpr int AR_NIMG = 4;                                        // ring of weight images: K step q + 4 is in flight while q computes
constexpr int AR_LDS = AR_NIMG * AR_B_IMG + 8 * 32 * AR_ROWB;     // weight images + per-wave epilogue slices

__host__ __device__ constexpr int ar_swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void ar_dma16(const u32x4a rsrc, const unsigned lds, const unsigned voff, const unsigned soff) {
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

// KS = K steps of 64 channels (Cin = 64 * KS).  A wave owns TMI x TNJ blocks of 32 x 32 of the 256 x 128 tile:
//   <KS, 2, 2>: 4 waves along M x 2 along N, 64 x 64 each (Cin <= 256: the activation fragment is <= 128 VGPRs)
//   <8, 1, 4>:  8 waves along M, 32 pixels x all 128 channels each (round 3: Cin = 512, the expansions of layer4 -- the
//               fragment of 32 pixels x 512 channels is 128 VGPRs; every weight fragment then feeds one MFMA instead of two)
//   <KS, 1, 4, 4>: FOUR waves x 32 pixels = 128 pixels per workgroup (round 4): at 1080p the stride-8 map has M = 32,400 pixels = 127
//               workgroups of 256 -- half the chip idle; 254 workgroups of 128 put one on every CU (the form 3 of conv1x1_b2b.hip)
template <int KS, int TMI = 2, int TNJ = 2, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) conv1x1_areg_kernel(const ConvArgs a, const int mtiles) {
    static_assert(TMI * TNJ == 4 && (TMI == 1 || TMI == 2), "a wave covers 4 blocks of 32 x 32");
    static_assert(NW == 8 || (NW == 4 && TNJ == 4), "8 waves cover 256 x 128, 4 waves 128 x 128");
    constexpr int BM = NW * TMI * TNJ * 32 * 32 / AR_BN;  // pixels per workgroup
    constexpr int PPW = 16 / NW;                          // weight pieces (1 KB) per wave and K step
    constexpr int WN = 4 / TNJ;  // waves along N
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int M = a.OH * a.OW;
    const int Kb = a.Cin * 2;  // bytes of a row of A / B
    int tile;
    {  // XCD-aware order: block b runs on XCD b % 8; every XCD gets a contiguous run of M tiles
        const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
        const int q = mtiles >> 3, r = mtiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = tile * BM;
    const int ntiles = (a.Cout + AR_BN - 1) / AR_BN;

    // ---- the wave's activation fragments, loaded once: rows wm * 32 TMI + i * 32 + (lane & 31), all Cin channels ----
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (unsigned)((size_t)a.H * a.W * a.Cin * 2), 0x00020000);
    h16x8a areg[TMI][KS * 4];
#pragma unroll
    for (int i = 0; i < TMI; i++) {
        const int m = m0 + wm * (32 * TMI) + i * 32 + (lane & 31);
        // (1x1, stride s: pixel (oy, ox) reads input pixel (oy * s, ox * s))
        const int oy = m / a.OW, ox = m - oy * a.OW;
        const unsigned base = m < M ? (unsigned)(oy * a.stride * a.W + ox * a.stride) * (unsigned)Kb + (unsigned)(lane >> 5) * 16u : OOBA;
#pragma unroll
        for (int sl = 0; sl < KS * 4; sl++)
            areg[i][sl] = __builtin_bit_cast(h16x8a, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, base, (unsigned)(sl * 32), 0));
    }

    // ---- weight stream: LDS-DMA, 8 whole rows per wave instruction, chunk index XOR-swizzled on the source side ----
    u32x4a wt_v;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(a.wt);
        wt_v.x = __builtin_amdgcn_readfirstlane((unsigned)v);
        wt_v.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        wt_v.z = __builtin_amdgcn_readfirstlane((unsigned)((size_t)a.Cout * Kb));
        wt_v.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_void_a*)smem;
    unsigned b_voff[PPW];  // per piece: row (within the N tile) * Kb + swizzled chunk * 16
    int b_row[PPW];
#pragma unroll
    for (int i = 0; i < PPW; i++) {
        const int row = 8 * (wave * PPW + i) + (lane >> 3);
        b_row[i] = row;
        b_voff[i] = (unsigned)row * (unsigned)Kb + (unsigned)(((lane & 7) ^ ar_swz(row)) * 16);
    }
    const int Q = ntiles * KS;  // linear (N tile, K step) counter
    // Every workgroup walks the N tiles cyclically from its own starting tile: if all of them streamed weight tile 0,
    // then 1, ... in lockstep, 256 CUs would pull the same 16 KB from the same few L2 channels at the same time
    // (measured: 2x SLOWER than the tiled kernel); staggered, the whole weight matrix is in use at any moment.
    const int nt_first = tile % ntiles;
    auto nt_of = [&](int w) { const int t = nt_first + w; return t >= ntiles ? t - ntiles : t; };
    auto dma_step = [&](int q) {
        const int wq = q / KS, ks = q - wq * KS;
        const int nt = nt_of(wq);
        const unsigned img = lds0 + (unsigned)((q & (AR_NIMG - 1)) * AR_B_IMG);
        const unsigned soff = (unsigned)(nt * AR_BN) * (unsigned)Kb + (unsigned)(ks * 128);
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            const bool ok = nt * AR_BN + b_row[i] < a.Cout;
            ar_dma16(wt_v, __builtin_amdgcn_readfirstlane(img + (unsigned)((wave * PPW + i) * 1024)), ok ? b_voff[i] : OOBA, __builtin_amdgcn_readfirstlane(soff));  // (OOB + soff stays beyond num_records: zeros)
        }
    };
    for (int q = 0; q < AR_NIMG && q < Q; q++) dma_step(q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int b_lds = (wn * (32 * TNJ) + (lane & 31)) * 128;
    const int b_swz = ar_swz(lane & 31);  // fragment rows are 32 apart: the swizzle does not change
    char* stage = smem + AR_NIMG * AR_B_IMG + wave * 32 * AR_ROWB;
    const int e_row = lane >> 4, e_col = lane & 15;  // epilogue: 16 lanes x 4 channels per pixel row, 4 rows per instruction
    // Output and residual go through buffer descriptors: a 32-bit lane offset plus a scalar row-group offset instead of
    // 64-bit addresses (the registers are needed for the activation fragments); rows beyond M fall outside num_records,
    // so their loads return zero and their stores are dropped by the hardware.
    const unsigned out_bytes = (unsigned)((size_t)M * a.Cout * 2);
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_bytes, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res), 0, a.res ? out_bytes : 0u, 0x00020000);
    const bool has_res = a.res != nullptr;
    const bool has_bias = a.bias != nullptr;
    const unsigned it_bytes = (unsigned)(4 * a.Cout * 2);  // four pixel rows further

    for (int wnt = 0; wnt < ntiles; wnt++) {
        const int nt = nt_of(wnt);
        f32x16a acc[TMI][TNJ];
#pragma unroll
        for (int i = 0; i < TMI; i++)
#pragma unroll
            for (int j = 0; j < TNJ; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int q = wnt * KS + ks;
            const char* Bb = smem + (q & (AR_NIMG - 1)) * AR_B_IMG + b_lds;
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                h16x8a fb[TNJ];
#pragma unroll
                for (int j = 0; j < TNJ; j++)
                    fb[j] = *reinterpret_cast<const h16x8a*>(Bb + j * 32 * 128 + (((2 * sl + (lane >> 5)) ^ b_swz) * 16));
#pragma unroll
                for (int i = 0; i < TMI; i++)
#pragma unroll
                    for (int j = 0; j < TNJ; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], areg[i][ks * 4 + sl], acc[i][j], 0, 0, 0);
            }
            // Before the barrier that ends K step q the wave's pieces of step q + 1 must have landed.  They were issued
            // three steps ago; younger than them are only the pieces of steps q + 2 and q + 3 (two instructions each) and
            // the stores of epilogues in between.  vmcnt counts loads and stores together and stores may retire ahead of
            // older loads, so only the DMA pieces are counted: a store still in flight makes the wait longer, never unsafe.
            const int younger = (q + 2 < Q ? 1 : 0) + (q + 3 < Q ? 1 : 0);
            if (younger == 2)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory");
            else if (younger == 1)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (q + AR_NIMG < Q) dma_step(q + AR_NIMG);  // the image just consumed takes K step q + 4
        }

        // ---- epilogue of this N tile: + bias, + residual, ReLU, f16 store; through the wave's own LDS slice so that 16
        //      lanes write the 128 contiguous bytes of a pixel's 64 channels (same scheme as conv_igemm_kernel) ----
#pragma unroll
        for (int hf = 0; hf < TNJ / 2; hf++) {  // 64 output channels at a time through the wave's LDS slice
        const int n = nt * AR_BN + wn * (32 * TNJ) + hf * 64 + e_col * 4;
        const bool n_ok = n < a.Cout;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bias && n_ok) bv = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
        for (int i = 0; i < TMI; i++) {
            // (Fetching the first block's residual before the K loop of the N tile, so that it lands behind the MFMAs, was
            //  measured too: no change -- 0.134-0.136 ms on layer3 conv3 at 4K either way; the kernel is at 0.9 of copy speed.)
            // All eight residual loads of this 32-row block are in flight before the accumulators are staged: with one
            // workgroup per CU nothing else hides their latency (two at a time made the whole kernel latency-bound,
            // 1.8 TB/s of stores).
            const unsigned eoff = n_ok ? ((unsigned)(m0 + wm * (32 * TMI) + i * 32 + e_row) * (unsigned)a.Cout + (unsigned)n) * 2u : OOBA;
            u32x2a rr[8];
#pragma unroll
            for (int it = 0; it < 8; it++) rr[it] = __builtin_amdgcn_raw_buffer_load_b64(res_rsrc, eoff, it * it_bytes, 0);
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 v = make_float4(acc[i][2 * hf + jj][4 * g + 0], acc[i][2 * hf + jj][4 * g + 1], acc[i][2 * hf + jj][4 * g + 2],
                                                 acc[i][2 * hf + jj][4 * g + 3]);
                    *reinterpret_cast<float4*>(stage + (lane & 31) * AR_ROWB + (jj * 32 + 8 * g + 4 * (lane >> 5)) * 4) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; it++) {
                float4 v = *reinterpret_cast<const float4*>(stage + (it * 4 + e_row) * AR_ROWB + e_col * 16);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (has_res) {
                    const h16x4a rv = __builtin_bit_cast(h16x4a, rr[it]);
                    v.x += (float)rv[0]; v.y += (float)rv[1]; v.z += (float)rv[2]; v.w += (float)rv[3];
                }
                if (a.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                const h16x4a hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2a, hv), out_rsrc, eoff, it * it_bytes, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        }
    }
}

template <int KS, int TMI = 2, int TNJ = 2, int NW = 8>
hipError_t launch_ks(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = NW * TMI * TNJ * 32 * 32 / AR_BN;
    const int M = a.OH * a.OW;
    const int mtiles = (M + BM - 1) / BM;
    auto k = conv1x1_areg_kernel<KS, TMI, TNJ, NW>;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, AR_LDS);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(mtiles), dim3(NW * 64), AR_LDS, s, a, mtiles);
    return hipGetLastError();
}

}  // namespace

bool conv1x1_areg_valid(const ConvArgs& a, int mode, int out_f32) {
    return mode == 1 && !out_f32 && a.KH == 1 && a.KW == 1 && a.pad == 0 && !a.in2 && a.batch <= 1 &&
           (a.Cin == 64 || a.Cin == 128 || a.Cin == 256 || a.Cin == 512) && a.Cout >= 256 && (a.Cout & (AR_BN - 1)) == 0 &&
           // (whole N tiles only: every conv3 of a ResNet; the ragged-N guards in the kernel are untested)
           (size_t)a.H * a.W * a.Cin * 2 < 0x80000000ull && (size_t)a.OH * a.OW * a.Cout * 2 < 0x80000000ull && (size_t)a.Cout * a.Cin * 2 < 0x80000000ull;
}

hipError_t launch_conv1x1_areg(const ConvArgs& a, hipStream_t s) {
    // the 128-pixel form where 256-pixel workgroups would leave more than a third of the 256 CUs without one (INFUR_AREG_BM=128 / 256:
    // measurement hook)
    static const int bm_env = getenv("INFUR_AREG_BM") ? atoi(getenv("INFUR_AREG_BM")) : 0;
    const bool small = bm_env == 128 || (bm_env != 256 && (a.OH * a.OW + 255) / 256 <= 170);
    if (small) {
        switch (a.Cin) {
            case 64: return launch_ks<1, 1, 4, 4>(a, s);
            case 128: return launch_ks<2, 1, 4, 4>(a, s);
            case 256: return launch_ks<4, 1, 4, 4>(a, s);
            case 512: return launch_ks<8, 1, 4, 4>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (a.Cin) {
        case 64: return launch_ks<1>(a, s);
        case 128: return launch_ks<2>(a, s);
        case 256: return launch_ks<4>(a, s);
        case 512: return launch_ks<8, 1, 4>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace infur
