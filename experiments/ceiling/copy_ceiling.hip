// copy_ceiling.hip -- what a hand-written streaming kernel sustains on this MI355X: the ceiling the HBM-bound kernels of the
// path (Winograd transforms, 1x1 convs with a residual, the fused conv3 -> conv1 pair) are graded against.
// (torch.Tensor.copy_ -- 5.1-5.2 TB/s -- understates it; MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy.)
//
//   hipcc --offload-arch=gfx950 -O3 experiments/ceiling/copy_ceiling.hip -o /tmp/copy_ceiling && /tmp/copy_ceiling
//
// Forms, all 16 bytes per lane, grid-stride, 256-thread blocks, 8 blocks per CU:
//   copy      y[i] = x[i]                      plain loads / stores; swept over 2..32 blocks per CU, 4 / 8 loads in flight per
//                                              lane, grid-stride vs one contiguous chunk per block: the best is reported
//   copy_nt   the same with non-temporal loads and stores (streaming data nothing re-reads)
//   read      sum of x (one float4 accumulator per lane, one atomicAdd-free write per block)
//   write     y[i] = const
//   rmw       y[i] += 1
// Reported: (bytes read + bytes written) / time, median of 20 launches after 3 warm-ups, for working sets from 64 MB (fits the
// 256 MB Infinity Cache) to 4 GB.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                          \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// U loads of 16 bytes in flight per lane before the first store.  CHUNK: a block streams one contiguous range (DRAM-page
// friendly) instead of the grid-stride interleave.
template <int U, bool NT, bool CHUNK>
__global__ void __launch_bounds__(256) copy_kernel(const f4* __restrict__ x, f4* __restrict__ y, size_t n) {
    size_t i, end, stride;
    if constexpr (CHUNK) {
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        i = (size_t)blockIdx.x * per + threadIdx.x;
        end = (size_t)(blockIdx.x + 1) * per < n ? (size_t)(blockIdx.x + 1) * per : n;
        stride = 256;
    } else {
        i = (size_t)blockIdx.x * 256 + threadIdx.x;
        end = n;
        stride = (size_t)gridDim.x * 256;
    }
    for (; i + (U - 1) * stride < end; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(x + i + k * stride) : x[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if constexpr (NT)
                __builtin_nontemporal_store(v[k], y + i + k * stride);
            else
                y[i + k * stride] = v[k];
        }
    }
    for (; i < end; i += stride) y[i] = x[i];
}

__global__ void __launch_bounds__(256) read_kernel(const f4* __restrict__ x, float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += x[i];
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[blockIdx.x] = s;  // (keeps the loads alive; practically never true)
}

__global__ void __launch_bounds__(256) write_kernel(f4* __restrict__ y, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = v;
}

__global__ void __launch_bounds__(256) rmw_kernel(f4* __restrict__ y, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = y[i] + 1.0f;
}

template <typename F>
static double median_us(F&& launch) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CHK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < 20; i++) {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    CHK(hipEventDestroy(e0));
    CHK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 8;
    printf("# %s, %d CUs, %d blocks of 256 threads, 16 B per lane\n", p.name, p.multiProcessorCount, blocks);
    printf("| MB | copy TB/s | copy nt TB/s | read TB/s | write TB/s | rmw TB/s |\n|---|---|---|---|---|---|\n");
    float* d_out;
    CHK(hipMalloc(&d_out, blocks * sizeof(float)));
    for (size_t mb : {64, 128, 256, 512, 1024, 4096}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        f4 *x, *y;
        CHK(hipMalloc(&x, bytes));
        CHK(hipMalloc(&y, bytes));
        CHK(hipMemset(x, 1, bytes));
        CHK(hipMemset(y, 0, bytes));
        // the copy is swept over its launch shape; the best one is the ceiling (and is printed)
        double c0 = 1e30, c1 = 1e30;
        char best[64] = "";
        for (int bpc : {2, 4, 8, 16, 32}) {
            const int nb = p.multiProcessorCount * bpc;
            auto take = [&](double us, const char* form, bool nt) {
                double& c = nt ? c1 : c0;
                if (us < c) {
                    c = us;
                    if (!nt) snprintf(best, sizeof best, "%s, %d blocks/CU", form, bpc);
                }
            };
            take(median_us([&] { hipLaunchKernelGGL((copy_kernel<4, false, false>), dim3(nb), dim3(256), 0, 0, x, y, n); }), "stride x4", false);
            take(median_us([&] { hipLaunchKernelGGL((copy_kernel<8, false, false>), dim3(nb), dim3(256), 0, 0, x, y, n); }), "stride x8", false);
            take(median_us([&] { hipLaunchKernelGGL((copy_kernel<8, false, true>), dim3(nb), dim3(256), 0, 0, x, y, n); }), "chunk x8", false);
            take(median_us([&] { hipLaunchKernelGGL((copy_kernel<8, true, false>), dim3(nb), dim3(256), 0, 0, x, y, n); }), "stride x8", true);
            take(median_us([&] { hipLaunchKernelGGL((copy_kernel<8, true, true>), dim3(nb), dim3(256), 0, 0, x, y, n); }), "chunk x8", true);
        }
        const double r = median_us([&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, x, d_out, n); });
        const double w = median_us([&] { hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, 0, y, n); });
        const double m = median_us([&] { hipLaunchKernelGGL(rmw_kernel, dim3(blocks), dim3(256), 0, 0, y, n); });
        auto tb = [&](double factor, double us) { return factor * (double)bytes / us / 1e6; };
        printf("| %zu | %.2f (%s) | %.2f | %.2f | %.2f | %.2f |\n", mb, tb(2, c0), best, tb(2, c1), tb(1, r), tb(1, w), tb(2, m));
        CHK(hipFree(x));
        CHK(hipFree(y));
    }
    return 0;
}
