// dvgr_planes.h -- device helpers shared by the kernels that stream pre-split bf16 hi / lo planes through an LDS ring
// (dvgr_gemm.hip: gemm_planes_kernel and its variants; dvgr_gatf.hip: the fused GAT-view kernel).
//   LDS plane tile, lane-linear per DMA instruction (global_load_lds writes base + lane * 16), 32-deep K-tile:
//     KC tile [rows][4 chunks of 8 bf16]:      chunk slot = chunk ^ ((-(row >> 2)) & 3)      (b128 reads conflict free)
//     IC tile [32 k][16 chunks of 8 bf16]:     chunk slot = chunk ^ swz(k), swz(k) = ((k & 3) | ((k >> 3) & 1) << 2) << 1
//                                              (ds_read_b64_tr_b16 blocks of 4 k-rows x 16 columns conflict free)
//   The swizzle is applied to the SOURCE address of the DMA and undone by the fragment read.
//   Lane (r, g) of an MFMA operand holds k = 8g .. 8g+7 of row / column r.
#pragma once
#include "dvgr_gemm.h"

namespace dvgr {

constexpr int BK = 32;                      // reduction depth of one LDS stage
constexpr int RT = 128;                     // tile extent (M and N) of the 128 x 128 engines
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int plane_swz_ic(int k) { return (((k & 3) | (((k >> 3) & 1) << 2)) << 1); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int LAYOUT>
__device__ __forceinline__ bf16x8 planes_frag(const uint16_t* tile, int idx16, int lane) {
  const int r = lane & 15, g = lane >> 4;
  if (LAYOUT == KC) {
    const int row = idx16 + r;
    const int slot = g ^ ((-(row >> 2)) & 3);
    return *reinterpret_cast<const bf16x8*>(tile + row * BK + slot * 8);
  } else {
    const int q = r >> 2, p = r & 3;
    const int k = 8 * g + q;
    const int c = (idx16 >> 3) + (p >> 1);
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    const uint16_t* a0 = tile + k * RT + ((c ^ plane_swz_ic(k)) << 3) + 4 * (p & 1);
    const uint16_t* a1 = a0 + 4 * RT;            // rows k + 4: same swizzle value
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 t = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
    return __builtin_bit_cast(bf16x8, t);
  }
}

}  // namespace dvgr
