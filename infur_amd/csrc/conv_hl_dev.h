// conv_hl_dev.h -- device helpers shared by the kernels of INFUR_DTYPE_F16_HL that multiply three-byte tensors (conv_hl.hip: tiled
// implicit GEMM; conv_hl_areg.hip: activation fragment in registers): the LDS-DMA instruction, the chunk swizzles of 64-byte (hi) and
// 32-byte (lo) LDS rows, the top-byte extraction of the cross-term operand.
#pragma once
#include "hl_format.h"
#include "kernels.h"

namespace infur {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr unsigned HL_OOB = 0x80000000u;  // out of range for every tensor accepted (< 2 GiB): the DMA lands zeros
constexpr int HL_KC = 32;                 // channels per K step

// LDS-DMA, 16 bytes per lane to (M0) + lane * 16 (conv_igemm_kernel.h: dma16 -- inline asm so that OUR vmcnt orders it)
__device__ __forceinline__ void hl_dma16(const u32x4r rsrc, const unsigned lds, const unsigned voff, const unsigned soff) {
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

__host__ __device__ constexpr int hl_swz64(int row) { return (row >> 2) & 3; }
__host__ __device__ constexpr int hl_swz32(int row) { return (row >> 3) & 1; }
// top bytes of the four f16 in (d0, d1): [d0.b1, d0.b3, d1.b1, d1.b3]
__device__ __forceinline__ int hl_top4(const unsigned d0, const unsigned d1) { return (int)__builtin_amdgcn_perm(d1, d0, 0x07050301u); }

}  // namespace infur
