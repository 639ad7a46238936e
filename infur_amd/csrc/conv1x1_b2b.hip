// conv1x1_b2b.hip -- two 1x1 convolutions back to back in ONE kernel, f16 operands:
//     y   = ReLU(W3 * t2 + b3 + x)        a bottleneck's conv3 + residual      (C2 -> 4*C2 channels, written once)
//     t1' = ReLU(W1' * y + b1')           the NEXT bottleneck's conv1          (4*C2 -> C2 channels)
// for the same pixels, so y -- the widest tensor of the stage (265 MB per launch at 4K in FCN-ResNet101's layer3) -- is
// written once as the next block's residual and never read back by conv1'.  Replaces two of the Conv/Add/Relu node
// groups ONNX Runtime executes inside `session.run` (infur/src/predict_onnx.rs:138).
//
// Why: in the f16-rate modes these 1x1 GEMMs are HBM-bound: conv3 reads t2 + x and writes y, conv1' reads
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
#include <type_traits>

#include "kernels.h"

namespace infur {

typedef float f32x16b __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8b __attribute__((ext_vector_type(8)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_b;

namespace {

constexpr int BB_BM = 256;   // pixels per workgroup: 8 waves x 32
constexpr int BB_NT = 32;    // channels of y per step
constexpr int BB_NSLOT = 3;  // ring depth: step t computes while t + 1 and t + 2 are in flight

template <int KS>
struct B2bLds {
    static constexpr int C2 = 64 * KS, C4 = 4 * C2, N2 = C2;
    static constexpr int W3_SLICE = BB_NT * C2 * 2;  // 32 rows x C2 f16: KS sub-images [32 rows][128 B]
    static constexpr int W1_SLICE = N2 * BB_NT * 2;  // N2 rows x 64 B
    static constexpr int WSLOT = W3_SLICE + W1_SLICE;
    static constexpr int RSLOT = 8 * 2048;           // 8 waves x [32 px][64 B]: residual in, y out (in place)
    static constexpr int OFF_R = BB_NSLOT * WSLOT;
    static constexpr int OFF_B3 = OFF_R + BB_NSLOT * RSLOT;
    static constexpr int OFF_B1 = OFF_B3 + C4 * 4;
    static constexpr int TOTAL = OFF_B1 + N2 * 4;
    static constexpr int FINAL = 8 * 32 * N2 * 2;    // staging of t1' (over the idle rings)
    static_assert(FINAL <= OFF_B3, "the t1' staging must not reach the bias tables");
    static_assert(TOTAL <= 160 * 1024, "LDS");
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

// KS = C2 / 64 (2: layer2, 4: layer3 of a ResNet-50/101)
template <int KS>
__global__ void __launch_bounds__(512, 2) conv1x1_b2b_kernel(const B2bArgs a, const int mtiles) {
    using L = B2bLds<KS>;
    constexpr int C2 = L::C2, C4 = L::C4, N2 = L::N2, NJ = N2 / 32, NSTEP = C4 / BB_NT;
    constexpr int NPW = KS / 2;        // DMA pieces (1 KB) per wave and step for each weight slice
    constexpr int PP = 2 * NPW + 2;    // ... plus the wave's two residual pieces
    static_assert(KS == 2 || KS == 4, "C2 = 128 or 256");
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
    const int m0w = tile * BB_BM + wave * 32;  // first pixel of this wave

    // ---- the wave's activation fragment, loaded once: pixel m0w + r, all C2 channels (rows >= M lie beyond num_records: zeros) ----
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (unsigned)((size_t)M * C2 * 2), 0x00020000);
    h16x8b areg[KS * 4];
    {
        const unsigned abase = (unsigned)(m0w + r) * (unsigned)(C2 * 2) + (unsigned)hh * 16u;
#pragma unroll
        for (int sl = 0; sl < KS * 4; sl++)
            areg[sl] = __builtin_bit_cast(h16x8b, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, abase, (unsigned)(sl * 32), 0));
    }
    // ---- bias tables into LDS (read back as broadcasts in the epilogues) ----
    for (int i = tid; i < C4 / 4; i += 512) reinterpret_cast<float4*>(smem + L::OFF_B3)[i] = reinterpret_cast<const float4*>(a.b3)[i];
    for (int i = tid; i < N2 / 4; i += 512) reinterpret_cast<float4*>(smem + L::OFF_B1)[i] = reinterpret_cast<const float4*>(a.b1)[i];

    // ---- the three DMA streams ----
    const u32x4b w3_v = bb_rsrc(a.w3, (unsigned)(C4 * C2 * 2));
    const u32x4b w1_v = bb_rsrc(a.w1, (unsigned)(N2 * C4 * 2));
    const u32x4b rs_v = bb_rsrc(a.res, (unsigned)((size_t)M * C4 * 2));
    const unsigned lds0 = (unsigned)(size_t)(lds_void_b*)smem;
    unsigned w3_voff[NPW], w1_voff[NPW], rs_voff[2];
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int p = wave * NPW + i;
        {  // W3 piece p: K sub-image p >> 2, rows 8 (p & 3) .. + 7, 8 chunks of 16 B each (chunk index XOR-swizzled on the source side)
            const int row = 8 * (p & 3) + (lane >> 3);
            w3_voff[i] = (unsigned)(b2b_pi(row) * C2 * 2 + (p >> 2) * 128 + (((lane & 7) ^ ((row >> 1) & 7)) * 16));
        }
        {  // W1' piece p: rows 16 p .. + 15 of the N2 x 64 B slice, 4 chunks each
            const int row = 16 * p + (lane >> 2);
            const int src = (row & ~31) | b2b_pi(row & 31);
            w1_voff[i] = (unsigned)src * (unsigned)(C4 * 2) + (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {  // residual piece i: pixels 16 i .. + 15 of the wave, the step's 64 bytes each
        const int row = 16 * i + (lane >> 2);
        rs_voff[i] = (unsigned)(m0w + row) * (unsigned)(C4 * 2) + (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
    }
    auto dma_step = [&](const int t, const int slot) {
        const unsigned wbase = lds0 + (unsigned)(slot * L::WSLOT);
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int p = wave * NPW + i;
            bb_dma16(w3_v, __builtin_amdgcn_readfirstlane(wbase + (unsigned)(p * 1024)), w3_voff[i], __builtin_amdgcn_readfirstlane((unsigned)(t * BB_NT * C2 * 2)));
        }
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int p = wave * NPW + i;
            bb_dma16(w1_v, __builtin_amdgcn_readfirstlane(wbase + (unsigned)(L::W3_SLICE + p * 1024)), w1_voff[i], __builtin_amdgcn_readfirstlane((unsigned)(t * BB_NT * 2)));
        }
        const unsigned rbase = lds0 + (unsigned)(L::OFF_R + slot * L::RSLOT + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; i++)
            bb_dma16(rs_v, __builtin_amdgcn_readfirstlane(rbase + (unsigned)(i * 1024)), rs_voff[i], __builtin_amdgcn_readfirstlane((unsigned)(t * BB_NT * 2)));
    };
    dma_step(0, 0);
    if (NSTEP > 1) dma_step(1, 1);

    // lane-constant LDS offsets
    int w3_lane[4];  // weight fragment of GEMM 1: row r of a [32][128 B] sub-image, chunk 2 q + hh
#pragma unroll
    for (int q = 0; q < 4; q++) w3_lane[q] = r * 128 + (((2 * q + hh) ^ ((r >> 1) & 7)) * 16);
    int d_lane[2];   // row r of a [rows][64 B] image, chunk 2 s + hh: weight fragment of GEMM 2, residual / y in the C/D layout
#pragma unroll
    for (int s = 0; s < 2; s++) d_lane[s] = r * 64 + (((2 * s + hh) ^ ((r >> 2) & 3)) * 16);
    const int t_row = lane >> 2, t_chunk = lane & 3;  // y store: 4 lanes x 16 B per pixel, 16 pixels per instruction

    const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)((size_t)M * C4 * 2), 0x00020000);
    const unsigned y_voff = (unsigned)(m0w + t_row) * (unsigned)(C4 * 2) + (unsigned)(t_chunk * 16);

    f32x16b acc2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc2[j][e] = 0.0f;
    // Everything the compiler knows to be in flight (the activation fragment, the bias tables) is waited for HERE, with the
    // builtin, so that its own wait-count bookkeeping starts the loop empty: otherwise it guards the first use of every
    // fragment register inside the loop with s_waitcnt vmcnt(18) ... vmcnt(3) -- counts that know nothing of the DMA pieces
    // issued by inline asm and would drain the two steps of prefetch on every step.
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) expcnt(7) lgkmcnt(0): also puts the bias tables in LDS before the first barrier

    auto step = [&](auto SC, const int t) __attribute__((always_inline)) {
        constexpr int S = decltype(SC)::value;
        // this wave's pieces of step t have landed: younger than them are only the PP pieces of step t + 1 (and stores,
        // which can only make the wait longer: loads retire in order among themselves)
        if (t + 1 < NSTEP)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's pieces of step t are in LDS, and every wave is done with step t - 1
        if (t + 2 < NSTEP) dma_step(t + 2, (S + 2) % BB_NSLOT);  // ... whose slot takes step t + 2

        // ---- GEMM 1: acc1[channel pi-row][pixel] over all C2 input channels ----
        const char* w3s = smem + S * L::WSLOT;
        f32x16b acc1;
#pragma unroll
        for (int e = 0; e < 16; e++) acc1[e] = 0.0f;
#pragma unroll
        for (int sl = 0; sl < KS * 4; sl++) {
            const h16x8b fb = *reinterpret_cast<const h16x8b*>(w3s + (sl >> 2) * 4096 + w3_lane[sl & 3]);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, areg[sl], acc1, 0, 0, 0);
        }
        // ---- epilogue 1: + bias, + residual, ReLU, f16 (same order as the unfused conv3) ----
        char* rs = smem + L::OFF_R + S * L::RSLOT + wave * 2048;
        h16x8b yop[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const h16x8b rv = *reinterpret_cast<const h16x8b*>(rs + d_lane[s]);
            const float* bp = reinterpret_cast<const float*>(smem + L::OFF_B3) + t * BB_NT + 16 * s + 8 * hh;
            const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
            const float bias[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float v = acc1[8 * s + e];
                v += bias[e];
                v += (float)rv[e];
                v = fmaxf(v, 0.f);
                yop[s][e] = (_Float16)v;
            }
            *reinterpret_cast<h16x8b*>(rs + d_lane[s]) = yop[s];  // in place: a lane reads and writes only its own two chunks
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int row = 16 * it + t_row;
            const u32x4b yv = *reinterpret_cast<const u32x4b*>(rs + row * 64 + ((t_chunk ^ ((row >> 2) & 3)) * 16));
            __builtin_amdgcn_raw_buffer_store_b128(yv, y_rsrc, y_voff + (unsigned)(it * 16 * C4 * 2), (unsigned)(t * BB_NT * 2), 0);
        }
        // ---- GEMM 2: this step's 32 channels of y are two k-slices of conv1' ----
        const char* w1s = smem + S * L::WSLOT + L::W3_SLICE;
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const h16x8b fb = *reinterpret_cast<const h16x8b*>(w1s + j * 2048 + d_lane[s]);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, yop[s], acc2[j], 0, 0, 0);
            }
    };

    for (int t = 0; t < NSTEP; t += BB_NSLOT) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < NSTEP) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < NSTEP) step(std::integral_constant<int, 2>{}, t + 2);
    }

    // ---- epilogue 2: t1' = ReLU(acc2 + b1') as f16, through the wave's own staging slice (over the idle rings) so that a
    //      pixel's N2 * 2 contiguous bytes leave in 16-byte lanes ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* fs = smem + wave * (32 * N2 * 2);
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
                float v = acc2[j][8 * s + e] + bias[e];
                v = fmaxf(v, 0.f);
                hv[e] = (_Float16)v;
            }
            const int c = 4 * j + 2 * s + hh;
            *reinterpret_cast<h16x8b*>(fs + r * (N2 * 2) + ((c ^ (r & 7)) * 16)) = hv;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = N2 / 8, RPI = 64 / CPR;  // 16-byte chunks per pixel row, pixel rows per store instruction
    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out2, 0, (unsigned)((size_t)M * N2 * 2), 0x00020000);
#pragma unroll
    for (int it = 0; it < 32 / RPI; it++) {
        const int row = it * RPI + lane / CPR, c = lane % CPR;
        const u32x4b v = *reinterpret_cast<const u32x4b*>(fs + row * (N2 * 2) + ((c ^ (row & 7)) * 16));
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rsrc, (unsigned)(m0w + row) * (unsigned)(N2 * 2) + (unsigned)(c * 16), 0, 0);
    }
}

template <int KS>
hipError_t launch_ks(const B2bArgs& a, hipStream_t s) {
    const int mtiles = (a.M + BB_BM - 1) / BB_BM;
    auto k = conv1x1_b2b_kernel<KS>;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!known || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, B2bLds<KS>::TOTAL);
        if (e != hipSuccess) return e;
        if (known) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(mtiles), dim3(512), B2bLds<KS>::TOTAL, s, a, mtiles);
    return hipGetLastError();
}

}  // namespace

bool conv1x1_b2b_valid(const B2bArgs& a) {
    return (a.C2 == 128 || a.C2 == 256) && a.M > 0 && a.relu1 && a.relu2 &&  // (both convs of a bottleneck end in ReLU) a.in && a.w3 && a.b3 && a.res && a.y && a.w1 && a.b1 && a.out2 &&
           // 32-bit buffer offsets; rows past M must stay below 2^32 as well
           ((size_t)a.M + BB_BM) * (size_t)a.C2 * 4 * 2 < 0x80000000ull;
}

hipError_t launch_conv1x1_b2b(const B2bArgs& a, hipStream_t s) {
    if (!conv1x1_b2b_valid(a)) return hipErrorInvalidValue;
    return a.C2 == 256 ? launch_ks<4>(a, s) : launch_ks<2>(a, s);
}

}  // namespace infur
