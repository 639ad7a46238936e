// ingest_rate.hip -- how many bytes per clock a CU takes in through each path (L2-resident source, 256 workgroups of 8 waves,
// one per CU):   0  buffer_load_dwordx4 ... lds   (LDS-DMA: the path conv_hl / the dmai forms / conv3x3_halo stage operands through)
//                1  global_load_dwordx4 -> VGPR    (consumed by an xor chain)
//                2  global_load_dwordx4 -> VGPR -> ds_write_b128   (not run: hipcc hoists the loop-invariant store; the K-step kernels below cover it)
//                3  buffer_load_dword ... lds      (4-byte DMA, for the per-instruction cost)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/ingest_rate.hip -o gpurun_out/ingest_rate && gpurun_out/ingest_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(const u32x4 rsrc, unsigned lds, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4(const u32x4 rsrc, unsigned lds, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds), "s"(rsrc), "s"(soff) : "memory");
}

// every wave moves `iters` x PIECES KB; the source window (bytes) is walked cyclically so that it stays in L2
template <int MODE, int PIECES>
__global__ void __launch_bounds__(512) ingest(const char* src, unsigned window, int iters, unsigned* sink, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long v = reinterpret_cast<unsigned long long>(src);
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    rs.z = window;
    rs.w = 0x00020000u;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    unsigned off = ((blockIdx.x * 8 + wave) * 4096u) % window;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < PIECES; p++) {
            const unsigned o = (off + p * 1024u) % window;
            if (MODE == 0) dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (wave * PIECES + p) * 1024u), lane * 16u, __builtin_amdgcn_readfirstlane(o));
            if (MODE == 3) dma4(rs, __builtin_amdgcn_readfirstlane(lds0 + (wave * PIECES + p) * 256u), lane * 4u, __builtin_amdgcn_readfirstlane(o));
            if (MODE == 1 || MODE == 2) {
                const u32x4 x = *reinterpret_cast<const u32x4*>(src + o + lane * 16);
                if (MODE == 1) acc ^= x;
                if (MODE == 2) *reinterpret_cast<u32x4*>(smem + (wave * PIECES + p) * 1024 + lane * 16) = x;
            }
        }
        if (MODE == 0 || MODE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");  // one iteration in flight
        off = (off + PIECES * 1024u * 8u * 32u) % window;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 2) acc.x ^= *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    if (acc.x == 0x12345678u && acc.y == 1u) sink[0] = acc.z ^ acc.w;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// The ACTIVATION operand's pattern: workgroup b owns 256 rows of `stride` bytes; per K step every wave issues PIECES DMA
// instructions, each covering 1024 / SEG rows x SEG contiguous bytes (SEG = 32: the lo plane of conv_hl, 64: its hi plane, 128: an f16 /
// i8 operand); the step advances SEG bytes along the rows.  `steps` K steps, then the next 256 rows.
template <int SEG, int PIECES>
__global__ void __launch_bounds__(512) ingest_rows(const char* src, unsigned bytes, unsigned stride, int steps, int iters, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long v = reinterpret_cast<unsigned long long>(src);
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    rs.z = bytes;
    rs.w = 0x00020000u;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    constexpr int LPR = SEG / 16, RPP = 64 / LPR;  // lanes per row, rows per piece
    unsigned voff[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; p++) voff[p] = (unsigned)((wave * PIECES + p) * RPP + lane / LPR) * stride + (unsigned)(lane % LPR) * 16u;
    const unsigned rows_per_wg = 8 * PIECES * RPP;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned blk = blockIdx.x;
    for (int it = 0; it < iters; it++) {
        const unsigned base = (unsigned)(((unsigned long long)blk * rows_per_wg * stride) % (bytes - rows_per_wg * stride));
        for (int k = 0; k < steps; k++) {
#pragma unroll
            for (int p = 0; p < PIECES; p++)
                dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + ((k & 1) * 8 * PIECES + wave * PIECES + p) * 1024u), voff[p], __builtin_amdgcn_readfirstlane(base + (unsigned)k * SEG));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        }
        blk += gridDim.x;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int SEG, int PIECES>
void run_rows(const char* src, unsigned bytes, unsigned stride, unsigned long long* cyc) {
    const int wgs = 256, steps = stride / SEG, iters = 4;
    auto k = ingest_rows<SEG, PIECES>;
    const int lds = 120 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, src, bytes, stride, steps, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto c : h) mx = c > mx ? c : mx;
    const double bytes_per_wg = (double)iters * steps * PIECES * 1024.0 * 8;
    printf("rows of %4u B, %3d-byte segments, %2d pieces/wave/step (%3d KB per step per CU): %7.3f ms  %6.2f TB/s chip  %6.1f B per tick per CU\n", stride, SEG, PIECES,
           PIECES * 8, ms, bytes_per_wg * wgs / ms / 1e9, bytes_per_wg / (double)mx);
}


// Does LDS-DMA overlap with the matrix pipe and with LDS reads of the same CU?  Per iteration every wave issues PIECES DMA instructions
// (L2-resident source), then NM MFMAs (32x32x16 f16, 32 cycles each) on registers, optionally fed by ds_read_b128 of the region the
// PREVIOUS iteration's DMA filled, then waits for its DMA and (optionally) a workgroup barrier -- the shape of a K step.
typedef float f32x16m __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
template <int PIECES, int NM, bool DMA, bool LDSRD, bool BAR>
__global__ void __launch_bounds__(512, 2) overlap(const char* src, unsigned window, int iters, float* sink, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long v = reinterpret_cast<unsigned long long>(src);
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    rs.z = window;
    rs.w = 0x00020000u;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    unsigned off = ((blockIdx.x * 8 + wave) * 4096u) % window;
    f32x16m acc[4];
    for (int j = 0; j < 4; j++)
        for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
    f16x8m fa, fb;
    for (int e = 0; e < 8; e++) { fa[e] = (_Float16)(lane * 0.001f + e); fb[e] = (_Float16)(e * 0.5f); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const unsigned half = (it & 1) * 8 * PIECES * 1024u;
        if (DMA) {
#pragma unroll
            for (int p = 0; p < PIECES; p++)
                dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + half + (wave * PIECES + p) * 1024u), lane * 16u, __builtin_amdgcn_readfirstlane((off + p * 1024u) % window));
        }
#pragma unroll
        for (int m = 0; m < NM; m++) {
            if (LDSRD && (m & 1) == 0) {  // one fragment read per two MFMAs (conv_hl: 18 reads per 24 MFMAs)
                const uint4 x = *reinterpret_cast<const uint4*>(smem + ((half ^ (8 * PIECES * 1024u)) + ((wave * PIECES * 1024u + m * 512u + lane * 16u) % (8 * PIECES * 1024u))));
                fb = __builtin_bit_cast(f16x8m, x);
            }
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[m & 3], 0, 0, 0);
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
        off = (off + PIECES * 1024u * 8u * 32u) % window;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int j = 0; j < 4; j++) r += acc[j][lane & 15];
    if (r == 12345.678f) sink[0] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int PIECES, int NM, bool DMA, bool LDSRD, bool BAR>
void run_overlap(const char* src, float* sink, unsigned long long* cyc) {
    const int wgs = 256, iters = 1000;
    auto k = overlap<PIECES, NM, DMA, LDSRD, BAR>;
    const int lds = 120 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, src, 1u << 20, iters, sink, cyc);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto c : h) mx = c > mx ? c : mx;
    printf("step: %2d KB DMA per CU %s, %2d MFMAs per wave (%4d MFMA cycles per SIMD)%s%s: %6.0f ticks per step\n", DMA ? PIECES * 8 : 0, DMA ? "" : "(off)", NM, NM * 32 * 2,
           LDSRD ? ", fragment reads" : "", BAR ? ", barrier" : "", (double)mx / iters);
}

// LDS read rate: 8 waves, ds_read_b128 (lane-linear: conflict-free) / ds_read_b64 in a loop, optionally with LDS-DMA writing another region
template <int NR, bool DMA, int WIDTH>
__global__ void __launch_bounds__(512, 2) lds_rate(const char* src, unsigned window, int iters, unsigned* sink, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long v = reinterpret_cast<unsigned long long>(src);
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    rs.z = window;
    rs.w = 0x00020000u;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    unsigned off = ((blockIdx.x * 8 + wave) * 4096u) % window;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (DMA) {
#pragma unroll
            for (int p = 0; p < 6; p++)
                dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + 65536u + (wave * 6 + p) * 1024u), lane * 16u, __builtin_amdgcn_readfirstlane((off + p * 1024u) % window));
        }
        unsigned base = (unsigned)(wave * 8192 + (it & 7) * 64);
        asm volatile("" : "+v"(base));  // (opaque: the reads stay in the loop)
#pragma unroll
        for (int m = 0; m < NR; m++) {
            if (WIDTH == 16) {
                const u32x4 x = *reinterpret_cast<const u32x4*>(smem + ((base + m * 1024 + lane * 16) & 65535));
                acc ^= x;
            } else {
                const unsigned long long x = *reinterpret_cast<const unsigned long long*>(smem + ((base + m * 512 + lane * 8) & 65535));
                acc.x ^= (unsigned)x; acc.y ^= (unsigned)(x >> 32);
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off = (off + 6 * 1024u * 8u * 32u) % window;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc.x == 0x12345678u && acc.y == 1u) sink[0] = acc.z ^ acc.w;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int NR, bool DMA, int WIDTH>
void run_lds(const char* src, unsigned* sink, unsigned long long* cyc) {
    const int wgs = 256, iters = 1000;
    auto k = lds_rate<NR, DMA, WIDTH>;
    const int lds = 120 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, src, 1u << 20, iters, sink, cyc);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto c : h) mx = c > mx ? c : mx;
    const double bytes = (double)NR * 64 * WIDTH * 8;
    printf("lds: %2d ds_read_b%d per wave per iteration (%3.0f KB per CU)%s: %6.0f ticks per iteration = %5.1f B per tick per CU of reads\n", NR, WIDTH * 8, bytes / 1024, DMA ? " + 48 KB of LDS-DMA" : "", (double)mx / iters,
           bytes / ((double)mx / iters));
}

// A K step with ONE operand through LDS-DMA + ds_read and the OTHER straight from L2 into registers (global_load_dwordx4, issued one
// step ahead): PIECES DMA instructions, NLDS fragment reads, NGL global loads and NM MFMAs per wave and step, a barrier per step.
template <int PIECES, int NLDS, int NGL, int NM>
__global__ void __launch_bounds__(512, 2) kstep_mixed(const char* src, unsigned window, int iters, float* sink, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long v = reinterpret_cast<unsigned long long>(src);
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)v);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    rs.z = window;
    rs.w = 0x00020000u;
    const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, window, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    unsigned off = ((blockIdx.x * 8 + wave) * 4096u) % window;
    f32x16m acc[4];
    for (int j = 0; j < 4; j++)
        for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
    f16x8m fa;
    for (int e = 0; e < 8; e++) fa[e] = (_Float16)(lane * 0.001f + e);
    u32x4 g[NGL > 0 ? NGL : 1], gn[NGL > 0 ? NGL : 1];
    for (int i = 0; i < NGL; i++) g[i] = __builtin_amdgcn_raw_buffer_load_b128(grs, (unsigned)(lane * 16), (off + i * 1024u) % window, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const unsigned half = (it & 1) * 8 * (PIECES > 0 ? PIECES : 1) * 1024u;
#pragma unroll
        for (int p = 0; p < PIECES; p++)
            dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + half + (wave * PIECES + p) * 1024u), lane * 16u, __builtin_amdgcn_readfirstlane((off + p * 1024u) % window));
#pragma unroll
        for (int i = 0; i < NGL; i++) gn[i] = __builtin_amdgcn_raw_buffer_load_b128(grs, (unsigned)(lane * 16), (off + (8 + i) * 1024u) % window, 0);
        unsigned base = (half ^ (8 * (PIECES > 0 ? PIECES : 1) * 1024u)) + (unsigned)(wave * 1024);
        asm volatile("" : "+v"(base));
#pragma unroll
        for (int m = 0; m < NM; m++) {
            f16x8m fb;
            if (NLDS > 0 && m % (NM / (NLDS > 0 ? NLDS : 1)) == 0 && m / (NM / (NLDS > 0 ? NLDS : 1)) < NLDS) {
                const uint4 x = *reinterpret_cast<const uint4*>(smem + ((base + (m * 512u) + lane * 16u) % (8 * (PIECES > 0 ? PIECES : 1) * 1024u)));
                fb = __builtin_bit_cast(f16x8m, x);
            } else {
                fb = __builtin_bit_cast(f16x8m, g[NGL > 0 ? m % NGL : 0]);
            }
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[m & 3], 0, 0, 0);
        }
        if (PIECES > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NGL) : "memory");  // (the global loads of the next step are younger)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < NGL; i++) g[i] = gn[i];
        off = (off + 16 * 1024u * 8u) % window;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int j = 0; j < 4; j++) r += acc[j][lane & 15];
    if (r == 12345.678f) sink[0] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int PIECES, int NLDS, int NGL, int NM>
void run_mixed(const char* src, float* sink, unsigned long long* cyc) {
    const int wgs = 256, iters = 1000;
    auto k = kstep_mixed<PIECES, NLDS, NGL, NM>;
    const int lds = 120 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, src, 1u << 20, iters, sink, cyc);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto c : h) mx = c > mx ? c : mx;
    printf("mixed step: %2d KB by LDS-DMA, %2d ds_read_b128 and %2d global_load_b128 per wave, %2d MFMAs per wave (%4d cycles per SIMD): %6.0f ticks per step\n", PIECES * 8, NLDS, NGL, NM,
           NM * 64, (double)mx / iters);
}

template <int MODE, int PIECES>
void run(const char* name, const char* src, unsigned window, unsigned* sink, unsigned long long* cyc, int wgs) {
    const int iters = 2000;
    auto k = ingest<MODE, PIECES>;
    const int lds = 120 * 1024;  // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, src, window, iters, sink, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto c : h) mx = c > mx ? c : mx;
    const double bytes_per_wg = (double)iters * PIECES * (MODE == 3 ? 256.0 : 1024.0) * 8;
    printf("%-44s pieces/wave/iter %d  window %4u KB: %7.3f ms  %6.2f TB/s chip  %6.1f B per s_memtime tick per CU (%llu ticks)\n", name, PIECES, window >> 10, ms,
           bytes_per_wg * wgs / ms / 1e9, bytes_per_wg / (double)mx, mx);
}

int main() {
    char* src;
    unsigned* sink;
    unsigned long long* cyc;
    const size_t cap = 64u << 20;
    hipMalloc(&src, cap);
    hipMemset(src, 1, cap);
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, 1024 * 8);
    const bool rows_only = getenv("ROWS_ONLY") != nullptr;
    for (unsigned window : {1u << 20, 32u << 20}) {
        if (rows_only) break;
        run<0, 3>("LDS-DMA b128", src, window, sink, cyc, 256);
        run<0, 6>("LDS-DMA b128", src, window, sink, cyc, 256);
        run<0, 12>("LDS-DMA b128", src, window, sink, cyc, 256);
        run<3, 12>("LDS-DMA b32", src, window, sink, cyc, 256);
        run<1, 6>("global_load b128 -> VGPR", src, window, sink, cyc, 256);
        run<1, 12>("global_load b128 -> VGPR", src, window, sink, cyc, 256);
    }
    // fewer CUs active: is the limit per CU or chip-wide?
    if (!rows_only) {
    run<0, 6>("LDS-DMA b128, 64 workgroups", src, 1u << 20, sink, cyc, 64);
    run<1, 12>("global_load b128 -> VGPR, 64 workgroups", src, 1u << 20, sink, cyc, 64);
    {
        float* fs;
        hipMalloc(&fs, 64);
        run_overlap<6, 32, true, false, false>(src, fs, cyc);
        run_overlap<6, 32, false, false, false>(src, fs, cyc);
        run_overlap<6, 0, true, false, false>(src, fs, cyc);
        run_overlap<6, 32, true, true, false>(src, fs, cyc);
        run_overlap<6, 32, false, true, false>(src, fs, cyc);
        run_overlap<6, 32, true, true, true>(src, fs, cyc);
        run_overlap<6, 32, false, true, true>(src, fs, cyc);
        run_overlap<6, 0, true, false, true>(src, fs, cyc);
        run_overlap<3, 16, true, true, true>(src, fs, cyc);
        run_overlap<3, 16, false, true, true>(src, fs, cyc);
        run_overlap<3, 0, true, false, true>(src, fs, cyc);
    }
    {
        float* fs2;
        hipMalloc(&fs2, 64);
        run_mixed<6, 16, 0, 32>(src, fs2, cyc);   // conv_hl today: both operands through LDS (48 KB DMA, 16-18 fragment reads)
        run_mixed<3, 12, 6, 32>(src, fs2, cyc);   // A through LDS (24 KB DMA, 12 reads), B from L2 into registers (6 loads)
        run_mixed<3, 8, 8, 32>(src, fs2, cyc);
        run_mixed<0, 0, 16, 32>(src, fs2, cyc);   // everything from L2 into registers
        run_mixed<3, 12, 0, 32>(src, fs2, cyc);   // (the LDS half alone)
    }
    run_lds<16, false, 16>(src, sink, cyc);
    run_lds<32, false, 16>(src, sink, cyc);
    run_lds<32, true, 16>(src, sink, cyc);
    run_lds<32, false, 8>(src, sink, cyc);
    }
    // the activation operand's pattern (rows of a [pixels][C] tensor, one K step = one segment of every row), 256 MB source
    char* big;
    const unsigned big_bytes = 256u << 20;
    hipMalloc(&big, big_bytes);
    hipMemset(big, 1, big_bytes);
    if (getenv("ROWS_ONLY")) {
        // round 6: the same row-segment pattern with the rows L2-RESIDENT (every workgroup walks the same few MB) -- is it the
        // pattern (partial lines per request) or HBM that holds the activation operand at 11-12 B/clk/CU?
        for (unsigned window : {2u << 20, 8u << 20, 64u << 20, 256u << 20}) {
            printf("-- window %u MB\n", window >> 20);
            run_rows<32, 1>(big, window, 2048, cyc);
            run_rows<64, 2>(big, window, 4096, cyc);
            run_rows<128, 4>(big, window, 4096, cyc);
            run_rows<128, 8>(big, window, 4096, cyc);
            run_rows<64, 2>(big, window, 4096 + 128, cyc);   // pitch-padded rows
            run_rows<128, 4>(big, window, 4096 + 128, cyc);
        }
        return 0;
    }
    for (unsigned stride : {1024u, 2048u, 4096u}) {
        run_rows<32, 1>(big, big_bytes, stride / 2, cyc);   // lo plane of 256 rows: 8 KB per step
        run_rows<64, 2>(big, big_bytes, stride, cyc);       // hi plane of 256 rows: 16 KB per step
        run_rows<64, 4>(big, big_bytes, stride, cyc);       // ... 512 rows
        run_rows<128, 4>(big, big_bytes, stride, cyc);      // 128-byte segments: 256 rows, 32 KB per step
        run_rows<128, 8>(big, big_bytes, stride, cyc);
    }
    return 0;
}
