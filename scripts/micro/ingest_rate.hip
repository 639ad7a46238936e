// ingest_rate.hip -- how many bytes per clock a CU takes in through each path (L2-resident source, 256 workgroups of 8 waves,
// one per CU):   0  buffer_load_dwordx4 ... lds   (LDS-DMA: the path conv_hl / the dmai forms / conv3x3_halo stage operands through)
//                1  global_load_dwordx4 -> VGPR    (consumed by an xor chain)
//                2  global_load_dwordx4 -> VGPR -> ds_write_b128
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
    for (unsigned window : {1u << 20, 32u << 20}) {
        run<0, 3>("LDS-DMA b128", src, window, sink, cyc, 256);
        run<0, 6>("LDS-DMA b128", src, window, sink, cyc, 256);
        run<0, 12>("LDS-DMA b128", src, window, sink, cyc, 256);
        run<3, 12>("LDS-DMA b32", src, window, sink, cyc, 256);
        run<1, 6>("global_load b128 -> VGPR", src, window, sink, cyc, 256);
        run<1, 12>("global_load b128 -> VGPR", src, window, sink, cyc, 256);
        run<2, 6>("global_load b128 -> VGPR -> ds_write_b128", src, window, sink, cyc, 256);
        run<2, 12>("global_load b128 -> VGPR -> ds_write_b128", src, window, sink, cyc, 256);
    }
    // fewer CUs active: is the limit per CU or chip-wide?
    run<0, 6>("LDS-DMA b128, 64 workgroups", src, 1u << 20, sink, cyc, 64);
    run<1, 12>("global_load b128 -> VGPR, 64 workgroups", src, 1u << 20, sink, cyc, 64);
    // the activation operand's pattern (rows of a [pixels][C] tensor, one K step = one segment of every row), 256 MB source
    char* big;
    const unsigned big_bytes = 256u << 20;
    hipMalloc(&big, big_bytes);
    hipMemset(big, 1, big_bytes);
    for (unsigned stride : {1024u, 2048u, 4096u}) {
        run_rows<32, 1>(big, big_bytes, stride / 2, cyc);   // lo plane of 256 rows: 8 KB per step
        run_rows<64, 2>(big, big_bytes, stride, cyc);       // hi plane of 256 rows: 16 KB per step
        run_rows<64, 4>(big, big_bytes, stride, cyc);       // ... 512 rows
        run_rows<128, 4>(big, big_bytes, stride, cyc);      // 128-byte segments: 256 rows, 32 KB per step
        run_rows<128, 8>(big, big_bytes, stride, cyc);
    }
    return 0;
}
