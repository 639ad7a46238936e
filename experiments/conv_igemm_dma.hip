// conv_igemm_dma.hip -- implicit-GEMM convolution on the f32 matrix core with LDS-DMA staging.
//
// Same GEMM view, tiling, MFMA operand trick and epilogue as conv_igemm.hip, but the operand
// tiles travel HBM/L2 -> LDS directly (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no
// ds_write pass.  Ablation on MI355X showed the VGPR round trip (load returns + ds_write_b128
// operand reads) competing with the MFMA's accumulator traffic was the largest loss of the
// register-staged kernel (139 -> 152 TFLOP/s with staging removed).
//
// LDS image: an LDS-DMA writes wave-uniform base + lane*16, so rows cannot be padded.  Rows are
// BK floats (CH = BK/4 chunks of 16 B); chunk c of row r is stored at position
// p = c ^ ((r / RPB) % CH), RPB = rows per 256-B bank row.  The permutation is applied on the
// per-lane SOURCE address of the DMA and again on the ds_read_b128 address, which makes every
// 16-lane read group hit 16 distinct 16-B slots (conflict-free) while each 8/4-lane group of
// the DMA still fetches one whole contiguous row slice from memory.
//
// Ring of NBUF = 3 LDS buffers, one raw s_barrier per K step:
//   step k computes from buf[k % 3]; DMA(k+1) has landed (each wave waits for its own pieces
//   with a counted vmcnt, then the barrier), DMA(k+2) is in flight, and right after the
//   barrier of step k the now-free buf[k % 3] is re-armed with DMA(k+3): two K steps of
//   latency cover.  Out-of-range taps / ragged tiles use the buffer descriptor's bounds check
//   (offset >= num_records -> zeros written to LDS).
#include <type_traits>

#include "kernels.h"

namespace infur {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr unsigned OOB_OFF = 0x80000000u;
constexpr int NBUF = 3;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16 bytes per lane, HBM/L2 -> LDS at (wave-uniform byte address) lds + lane*16; an offset >=
// num_records writes zeros.  Issued from inline asm on purpose: hipcc (ROCm 7.2) cannot tell the
// ring slots apart and drains every LDS-DMA it knows about with `s_waitcnt vmcnt(0)` before the
// next ds_read, which serialises the pipeline; the asm form is invisible to that pass and is
// ordered by our own counted vmcnt + barrier (cdna_hip_programming.md section 5.7).  M0 (the DMA's
// LDS base) is compiler-reserved: saved and restored inside the same statement.
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds, unsigned voff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds), "s"(rsrc)
        : "memory");
}

// raw buffer descriptor: base, stride 0, num_records bytes, DATA_FORMAT=32 (as make_buffer_rsrc)
__device__ __forceinline__ u32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#if defined(__HIP_DEVICE_COMPILE__)
// MFMA side of a workgroup: fragment reads, 64 (or 32) MFMAs per K step, one barrier per K
// step, fused epilogue.  SELF_ISSUE: this wave also re-arms the ring (via `issue`) and waits for
// its own DMA pieces before each barrier; otherwise producer waves do that.
template <int BM, int BN, int WM, int WN, int BK, bool SELF_ISSUE, typename IssueFn>
__device__ __forceinline__ void mfma_loop(const ConvArgs& a, char* smem, int lane, int wm, int wn, int m0, int n0,
                                          int M, int ksteps, IssueFn& issue) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int CH = BK / 4;
    constexpr int ROWB = BK * 4;
    constexpr int RPI = 64 / CH;
    constexpr int RPB = 256 / ROWB;
    constexpr int NS = BK / 8;
    constexpr int IPW = (BM / RPI + BN / RPI) / NW;
    constexpr int BUFB = (BM + BN) * ROWB;

    // fragment reads: lane reads row (lane & 31) of each 32-row tile, chunk 2*kk + (lane >> 5),
    // at its swizzled position
    const int a_row = wm * TM * 32 + (lane & 31);
    const int b_row = wn * TN * 32 + (lane & 31);
    const int a_swz = (a_row / RPB) % CH, b_swz = (b_row / RPB) % CH;  // same for every 32-row tile
    const int a_base = a_row * ROWB, b_base = BM * ROWB + b_row * ROWB;
    auto read_frags = [&](int slot, int kk, float4 (&fa)[TM], float4 (&fb)[TN]) {
        const char* base = smem + slot * BUFB;
        const int c = kk * 2 + (lane >> 5);
        const char* Ab = base + a_base + ((c ^ a_swz) << 4);
        const char* Bb = base + b_base + ((c ^ b_swz) << 4);
#pragma unroll
        for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * ROWB);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    float4 fa[TM], fb[TN], fa_n[TM], fb_n[TN];

    // One K step.  ISSUE: re-arm this step's buffer with DMA(ks+3) after the barrier.
    // WAIT: vmcnt value that guarantees this wave's DMA(ks+1) pieces have landed.
    // NEXT: there is a next K step (barrier + prefetch of its first fragments).
    auto k_step = [&](int slot, auto ISSUE, auto WAIT, auto NEXT) {
        const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
#pragma unroll
        for (int kk = 0; kk < NS; kk++) {
            if (kk < NS - 1)
                read_frags(slot, kk + 1, fa_n, fb_n);
            else if (NEXT)
                read_frags(nslot, 0, fa_n, fb_n);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    // D rows = output channels, D cols = pixels (operands swapped on purpose)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].z, fa[i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].w, fa[i].w, acc[i][j], 0, 0, 0);
                }
            if (kk == NS - 2 && NEXT) {
                // (my DMA(ks+1) pieces landed;) all my reads of this buffer completed
                if (SELF_ISSUE) wait_vmcnt<decltype(WAIT)::value>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (SELF_ISSUE && ISSUE) issue(slot);
            }
#pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = fa_n[i];
#pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = fb_n[j];
        }
    };
    constexpr auto Y = std::true_type{};
    constexpr auto N = std::false_type{};
    constexpr auto W1 = std::integral_constant<int, IPW>{};
    constexpr auto W0 = std::integral_constant<int, 0>{};

    read_frags(0, 0, fa, fb);
    int ks = 0, slot = 0;
    for (; ks + 3 < ksteps; ks++) {  // steady state
        k_step(slot, Y, W1, Y);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
    }
    if (ks + 2 < ksteps) {  // DMA(ks+2) still in flight, nothing left to issue
        k_step(slot, N, W1, Y);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        ks++;
    }
    if (ks + 1 < ksteps) {  // only DMA(ks+1) outstanding
        k_step(slot, N, W0, Y);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        ks++;
    }
    k_step(slot, N, W0, N);

    // epilogue (see conv_igemm.hip): a lane owns one pixel and 4 consecutive output channels
    // per register group -> 16-byte bias / residual loads and NHWC stores
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
                if (vec_ok) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                    if (a.res) {
                        const float4 rv = *reinterpret_cast<const float4*>(a.res + o);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    if (a.relu) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    *reinterpret_cast<float4*>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (n + t >= a.Cout) break;
                        float x = v[t] + a.bias[n + t];
                        if (a.res) x += a.res[o + t];
                        if (a.relu) x = fmaxf(x, 0.f);
                        a.out[o + t] = x;
                    }
                }
            }
        }
    }
}
#endif  // __HIP_DEVICE_COMPILE__

// NP = number of PRODUCER waves appended to the WM*WN MFMA waves.  NP == 0: every wave issues
// its share of the DMAs.  NP == 2: wave specialisation -- the MFMA waves never issue a VMEM
// instruction (an LDS-DMA costs its issuing wave 60-185 cycles, more than the 64-cycle shadow
// of an f32 MFMA, i.e. a bubble in that wave's MFMA stream); two extra waves do all address
// arithmetic and DMA issue and meet the consumers at the one barrier per K step.
template <int BM, int BN, int WM, int WN, int BK, int NP>
__global__ void __launch_bounds__((WM * WN + NP) * 64)
    conv_igemm_f32_dma_kernel(const ConvArgs a, const int mtiles, const int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)  // gfx950 builtins below have no meaning in the host pass (only a stub is needed there)
    constexpr int NW = WM * WN;          // waves
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int CH = BK / 4;           // 16-B chunks per row
    constexpr int ROWB = BK * 4;         // bytes per row
    constexpr int RPI = 64 / CH;         // rows per DMA instruction (1 KiB)
    constexpr int RPB = 256 / ROWB;      // rows per 256-B bank row
    constexpr int NS = BK / 8;           // 8-wide k slices per K step
    constexpr int NI = NP ? NP : NW;     // waves that issue DMAs
    constexpr int A_IPW = BM / RPI / NI; // DMA instructions per issuing wave per K step
    constexpr int B_IPW = BN / RPI / NI;
    constexpr int IPW = A_IPW + B_IPW;
    constexpr int BUFB = (BM + BN) * ROWB;
    static_assert((BM / RPI) % NI == 0 && (BN / RPI) % NI == 0, "every issuing wave must issue the same number of DMAs");
    static_assert(2 * IPW < 64, "vmcnt is a 6-bit counter");
    static_assert(NS >= 2, "need at least two slices per K step");

    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [NBUF][A tile | B tile]

    // XCD-aware tile order (see conv_igemm.hip)
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = NP > 0 && wave >= NW;   // wave-uniform role
    const int iw = NP > 0 ? wave - NW : wave;     // index among the issuing waves
    const int wm = wave / WN, wn = wave % WN;     // MFMA waves only

    const int M = a.OH * a.OW;
    const int Ktot = a.KH * a.KW * a.Cin;
    const int cchunks = a.Cin / BK;
    const int ksteps = a.KH * a.KW * cchunks;

    const u32x4 in_rsrc = make_rsrc(a.in, (unsigned)((size_t)a.H * a.W * a.Cin * 4));
    const u32x4 wt_rsrc = make_rsrc(a.wt, (unsigned)((size_t)a.Cout * Ktot * 4));
    // LDS byte address of the ring (dynamic LDS starts at the group segment's static size)
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;

    if (NP == 0 || producer) {
        // ---- DMA source coordinates: instruction i of this issuing wave covers tile rows
        //      (iw + i*NI)*RPI .. +RPI-1; this lane fetches row lane/CH, LDS position lane%CH ----
        int a_iy0[A_IPW], a_ix0[A_IPW];
        unsigned a_coff[A_IPW];
#pragma unroll
        for (int i = 0; i < A_IPW; i++) {
            const int row = (iw + i * NI) * RPI + lane / CH;
            const int c = (lane % CH) ^ ((row / RPB) % CH);
            const int m = m0 + row;
            const int oy = m / a.OW, ox = m - oy * a.OW;
            a_iy0[i] = m < M ? oy * a.stride - a.pad : -0x100000;  // fails every bounds test
            a_ix0[i] = ox * a.stride - a.pad;
            a_coff[i] = (unsigned)c * 16u;
        }
        unsigned b_off[B_IPW];
#pragma unroll
        for (int i = 0; i < B_IPW; i++) {
            const int row = (iw + i * NI) * RPI + lane / CH;
            const int c = (lane % CH) ^ ((row / RPB) % CH);
            const int n = n0 + row;
            b_off[i] = n < a.Cout ? (unsigned)n * (unsigned)Ktot * 4u + (unsigned)c * 16u : OOB_OFF;
        }
        int ky = 0, kx = 0, cc = 0, ks_issue = 0;  // coordinates of the K step being ISSUED

        // issue this wave's DMAs of the next unissued K step into ring slot `slot`
        auto issue = [&](int slot) {
            const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(slot * BUFB + iw * 1024));
            const int dy = ky * a.dil, dx = kx * a.dil;
            const unsigned cbyte = (unsigned)(cc * BK) * 4u;
#pragma unroll
            for (int i = 0; i < A_IPW; i++) {
                const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off = (unsigned)(iy * a.W + ix) * (unsigned)(a.Cin * 4) + cbyte + a_coff[i];
                dma16(in_rsrc, base + i * NI * 1024, ok ? off : OOB_OFF);
            }
            const unsigned koff = (unsigned)ks_issue * (BK * 4u);
#pragma unroll
            for (int i = 0; i < B_IPW; i++)
                dma16(wt_rsrc, base + BM * ROWB + i * NI * 1024, b_off[i] == OOB_OFF ? OOB_OFF : b_off[i] + koff);
            ks_issue += 1;
            cc += 1;
            const int w1 = cc == cchunks;
            cc = w1 ? 0 : cc;
            kx += w1;
            const int w2 = kx == a.KW;
            kx = w2 ? 0 : kx;
            ky += w2;
        };

        if (NP > 0) {
            // ================= producer wave: fill the ring, then one re-arm per K step =================
            issue(0);
            if (ksteps > 1) issue(1);
            if (ksteps > 2) issue(2);
            if (ksteps > 2)
                wait_vmcnt<2 * IPW>();
            else if (ksteps > 1)
                wait_vmcnt<IPW>();
            else
                wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // DMA(0) visible to the consumers
            int slot = 0;
            for (int ks = 0; ks + 1 < ksteps; ks++) {
                if (ks + 2 < ksteps)  // my DMA(ks+1) pieces landed, DMA(ks+2) may stay in flight
                    wait_vmcnt<IPW>();
                else
                    wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();  // consumers are done reading buf[slot]
                if (ks + 3 < ksteps) issue(slot);
                slot = slot + 1 == NBUF ? 0 : slot + 1;
            }
            return;
        }

        // ================= NP == 0: every wave is producer and consumer =================
        // (consumer code below re-arms through this lambda)
        issue(0);
        if (ksteps > 1) issue(1);
        if (ksteps > 2) issue(2);
        if (ksteps > 2)
            wait_vmcnt<2 * IPW>();
        else if (ksteps > 1)
            wait_vmcnt<IPW>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        mfma_loop<BM, BN, WM, WN, BK, true>(a, smem, lane, wm, wn, m0, n0, M, ksteps, issue);
        return;
    }
    // ================= consumer (MFMA) wave of the specialised kernel =================
    __builtin_amdgcn_s_barrier();  // pairs with the producers' "DMA(0) visible" barrier
    auto no_issue = [](int) {};
    mfma_loop<BM, BN, WM, WN, BK, false>(a, smem, lane, wm, wn, m0, n0, M, ksteps, no_issue);
#endif  // __HIP_DEVICE_COMPILE__
}


template <int BM, int BN, int WM, int WN, int BK, int NP>
static hipError_t launch_dma_cfg(const ConvArgs& a, hipStream_t s) {
    const int M = a.OH * a.OW;
    const int mtiles = (M + BM - 1) / BM;
    const int ntiles = (a.Cout + BN - 1) / BN;
    const size_t lds = (size_t)NBUF * (BM + BN) * BK * sizeof(float);
    auto k = conv_igemm_f32_dma_kernel<BM, BN, WM, WN, BK, NP>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(mtiles * ntiles), dim3((WM * WN + NP) * 64), lds, s, a, mtiles, ntiles);
    return hipGetLastError();
}

// mode: 16 / 32 = every wave issues (BK 16 / 32); 132 = BK 32 with two producer waves
hipError_t launch_conv_igemm_f32_dma(const ConvArgs& a, int mode, hipStream_t s) {
    if (a.Cin % 32 != 0) return hipErrorInvalidValue;
    if ((size_t)a.H * a.W * a.Cin * 4 >= 0x80000000ull || (size_t)a.Cout * a.KH * a.KW * a.Cin * 4 >= 0x80000000ull)
        return hipErrorInvalidValue;
    if (mode == 132) {
        if (a.Cout >= 128) return launch_dma_cfg<128, 128, 2, 2, 32, 2>(a, s);
        if (a.Cout > 32) return launch_dma_cfg<128, 64, 2, 2, 32, 2>(a, s);
        return launch_dma_cfg<256, 32, 4, 1, 32, 2>(a, s);
    }
    if (a.Cout >= 128) return mode == 16 ? launch_dma_cfg<128, 128, 2, 2, 16, 0>(a, s) : launch_dma_cfg<128, 128, 2, 2, 32, 0>(a, s);
    if (a.Cout > 32) return mode == 16 ? launch_dma_cfg<128, 64, 2, 2, 16, 0>(a, s) : launch_dma_cfg<128, 64, 2, 2, 32, 0>(a, s);
    return launch_dma_cfg<256, 32, 4, 1, 32, 0>(a, s);
}

}  // namespace infur
