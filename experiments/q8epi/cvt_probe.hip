// probe: does v_cvt_pk_u8_f32 saturate (so the med3 in front of it is redundant for integer-valued inputs), and are the packed-f32
// multiply / add bitwise the scalar IEEE operations?   hipcc --offload-arch=gfx950 -O2 cvt_probe.hip -o cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, int n, const float* a, const float* b, float* pm, float* pa) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0xAABBCC00u);
    f32x2 x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]}, m, s;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(x), "v"(y));
    pm[2 * i] = m[0]; pm[2 * i + 1] = m[1]; pa[2 * i] = s[0]; pa[2 * i + 1] = s[1];
}
int main() {
    const int n = 4096;
    float h[n]; unsigned o[n];
    float vals[] = {-1e30f, -300.f, -1.f, -0.f, 0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 254.f, 254.5f, 255.f, 255.5f, 256.f, 300.f, 1e9f, 1e30f, INFINITY, -INFINITY};
    int nv = sizeof(vals) / 4;
    for (int i = 0; i < n; i++) h[i] = i < nv ? vals[i] : (float)(i - 2000);
    float *ha = new float[2 * n], *hb = new float[2 * n], *hm = new float[2 * n], *hs = new float[2 * n];
    unsigned seed = 12345;
    for (int i = 0; i < 2 * n; i++) {
        seed = seed * 1664525u + 1013904223u; unsigned u = seed; seed = seed * 1664525u + 1013904223u; unsigned v = seed;
        // random bit patterns of moderate exponent + some subnormal-producing pairs
        u = (u & 0x807fffffu) | ((100u + (u >> 23) % 56u) << 23); v = (v & 0x807fffffu) | ((100u + (v >> 23) % 56u) << 23);
        if (i % 64 == 0) { u = (u & 0x807fffffu) | (20u << 23); v = (v & 0x807fffffu) | (30u << 23); }
        memcpy(&ha[i], &u, 4); memcpy(&hb[i], &v, 4);
    }
    float *d, *da, *db, *dm, *ds; unsigned* dout;
    hipMalloc(&d, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&da, 8 * n); hipMalloc(&db, 8 * n); hipMalloc(&dm, 8 * n); hipMalloc(&ds, 8 * n);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice); hipMemcpy(da, ha, 8 * n, hipMemcpyHostToDevice); hipMemcpy(db, hb, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, dout, n, da, db, dm, ds);
    hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hm, dm, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 8 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < nv; i++) printf("cvt_pk_u8_f32(%g) = %u (upper bytes %06x)\n", h[i], o[i] & 255, o[i] >> 8);
    int bad = 0;
    for (int i = nv; i < n; i++) { float v = h[i]; unsigned want = v < 0 ? 0 : (v > 255 ? 255 : (unsigned)v); if ((o[i] & 255) != want) bad++; }
    printf("integer-valued inputs -2000..2095: %d mismatches against saturate(v)\n", bad);
    int bm = 0, ba = 0, sub = 0;
    for (int i = 0; i < 2 * n; i++) {
        volatile float m = ha[i] * hb[i], s = ha[i] + hb[i]; float mm = m, ss = s;
        if (std::fpclassify(mm) == FP_SUBNORMAL) sub++;
        if (memcmp(&mm, &hm[i], 4)) { if (bm < 5) printf("mul differs: %a * %a = %a (cpu) %a (gpu)\n", ha[i], hb[i], mm, hm[i]); bm++; }
        if (memcmp(&ss, &hs[i], 4)) ba++;
    }
    printf("v_pk_mul_f32: %d of %d differ from the scalar IEEE product (%d subnormal results); v_pk_add_f32: %d differ\n", bm, 2 * n, sub, ba);
    return 0;
}
