// Is the scalar offset of a raw buffer access part of the bounds check on gfx950?  (It decides whether "row beyond M" may be
// expressed as voffset-in-range + soffset: conv1x1_areg.hip's epilogue.)   hipcc --offload-arch=gfx950 soffset_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* buf, unsigned records_bytes, unsigned soff, unsigned* loaded) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, records_bytes, 0x00020000);
    const unsigned voff = threadIdx.x * 4;  // in range for every lane
    loaded[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b32(0xdeadbeefu, rsrc, voff, soff, 0);
}
int main() {
    unsigned *d, *l;
    const int n = 1024;
    hipMalloc(&d, n * 4);
    hipMalloc(&l, 64 * 4);
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; i++) h[i] = 0x1000 + i;
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    // descriptor covers the first 256 bytes (64 dwords); soffset = 512 bytes moves every lane's access beyond it
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 256u, 512u, l);
    hipDeviceSynchronize();
    std::vector<unsigned> lo(64);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(lo.data(), l, 64 * 4, hipMemcpyDeviceToHost);
    int stray = 0;
    for (int i = 64; i < n; i++) stray += h[i] == 0xdeadbeefu;
    printf("records 256 B, voffset 0..252, soffset 512: loads returned %#x %#x (0 = dropped), %d dwords beyond num_records overwritten\n", lo[0], lo[63], stray);
    printf("=> soffset %s part of the range check\n", stray ? "is NOT" : "IS");
    return 0;
}
