// dvgr_gatf.h -- what the two fused GAT-view kernels share (dvgr_gatf.hip: projection + attention forward; dvgr_gatb.hip: the
// dz contraction + attention backward): the tile configuration, the loader waves' LDS-DMA ring and the computing waves' K loop
// on a tile of MTILES x 16 rows (whole samples) x F = 64 NTC columns (one head).  Structure of gemm_planes_kernel (dvgr_gemm.hip).
#pragma once
#include "dvgr_planes.h"
#include "dvgr_stages.h"

namespace dvgr {

template <int TERMS, int MTILES, int NTC>
struct GatfCfg {
  static constexpr int kPlanes = TERMS == 3 ? 2 : 1;
  static constexpr int RP = 16 * MTILES;                  // rows of the A tile
  static constexpr int F = 64 * NTC;                      // columns of the tile = width of one head
  static constexpr int FS = F + 16;                       // shZ row stride (floats): 16 banks mod 64, the fp32 B fragment is conflict free
  static constexpr int kTileA = RP * BK;                  // bf16 elements of one A plane tile
  static constexpr int kTileB = F * BK;
  static constexpr int kSlot = kPlanes * (kTileA + kTileB);
  static constexpr int NS = (4 * kSlot * 2 <= 150 * 1024) ? 4 : 3;   // ring slots
  static constexpr int IA = MTILES;                       // DMA instructions per plane and K-tile: A tile (64 chunks of 16 bytes each)
  static constexpr int IB = F / 16;                       //   B tile
  static constexpr int ITOT = IA + IB;
  static constexpr int IHI = (ITOT + 3) / 4;              // instructions of loader wave w: IHI for w < ITOT % 4 (all of them when
  static constexpr int ILO = ITOT / 4;                    //   ITOT % 4 == 0), ILO otherwise
  static constexpr int MT0 = (MTILES + 1) / 2;            // row tiles of the row group 0 waves, MTILES - MT0 of the others
  static constexpr int kRingBytes = NS * kSlot * 2;
};

// ---- loader waves -------------------------------------------------------------------------------------------------
// xa / x_lo: hi plane of the A matrix (rows m0 .. of it, clamped to BN - 1) and its lo-plane offset; wb / w_lo: the same of
// the B matrix, whose rows brow0 .. brow0 + F - 1 form the tile (forward: head h of Wcat; backward: the head's columns of P1^T)
// NPF / pf: optionally, NPF further vector-memory instructions of this wave (plain loads into registers: an epilogue's inputs)
// issued by pf() right behind the ring's prologue, i.e. YOUNGER than the first NS - 1 K-tiles: the counted waits for those
// tiles leave them in flight (loads, stores and LDS-DMA share one in-order counter), every later wait covers them.
struct GatfNoPrefetch { __device__ __forceinline__ void operator()() const {} };
template <class C, int TERMS, int CNT, int NPF = 0, class PF = GatfNoPrefetch>
__device__ __forceinline__ void gatf_loader(const uint16_t* xa, int64_t x_lo, const uint16_t* wb, int64_t w_lo, uint16_t* smem,
                                            int brow0, int m0, int BN, int D, int nk, int wave, int lane, PF pf = PF()) {
  // instruction k of this wave is instruction i = wave + 4 k of the K-tile: i < IA -> rows 16 i .. 16 i + 15 of the A tile,
  // else rows 16 (i - IA) .. of the B tile.  Per-lane element offsets are computed once (the K loop moves a uniform base).
  uint32_t off[CNT];
  bool is_a[CNT];
  int dst[CNT];                                            // element offset of the instruction's 1 KB piece inside the slot (hi plane)
#pragma unroll
  for (int k = 0; k < CNT; ++k) {
    const int i = wave + 4 * k;
    is_a[k] = i < C::IA;
    const int j = is_a[k] ? i : i - C::IA;
    const int S = j * 64 + lane;
    const int row = S >> 2, slot = S & 3;
    const int chunk = slot ^ ((-(row >> 2)) & 3);
    if (is_a[k]) {
      int gi = m0 + row;
      gi = gi < BN ? gi : BN - 1;                          // rows past the matrix: any valid row (never stored)
      off[k] = (uint32_t)gi * (uint32_t)D + (uint32_t)(chunk * 8);
      dst[k] = j * 512;
    } else {
      off[k] = (uint32_t)(brow0 + row) * (uint32_t)D + (uint32_t)(chunk * 8);
      dst[k] = C::kPlanes * C::kTileA + j * 512;
    }
  }
  auto issue = [&](int kt, uint16_t* slot_base) {
    const int k0 = kt * BK;
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
      const uint16_t* base = (is_a[k] ? xa : wb) + k0;
      const int64_t lo = is_a[k] ? x_lo : w_lo;
      const int ptile = is_a[k] ? C::kTileA : C::kTileB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[k]),
                                       (__attribute__((address_space(3))) void*)(slot_base + dst[k]), 16, 0, 0);
      if (TERMS == 3)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + lo + off[k]),
                                         (__attribute__((address_space(3))) void*)(slot_base + dst[k] + ptile), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < C::NS - 1; ++s)
    if (s < nk) issue(s, smem + s * C::kSlot);
  if (NPF > 0) {                                           // (program order IS the order the waits below count in)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pf();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int kPer = C::kPlanes * CNT;                   // DMA instructions of this wave per K-tile
  static_assert(2 * kPer + NPF < 64, "vmcnt is a 6-bit counter");
  for (int kt = 0; kt < nk; ++kt) {
    const int younger = (nk - 1 - kt) < C::NS - 2 ? (nk - 1 - kt) : C::NS - 2;
    // (the NPF loads sit between K-tile NS - 2 and K-tile NS - 1 in the queue: younger than the tile waited for while kt <= NS - 2;
    // a K loop shorter than NS + 1 tiles simply waits for them as well)
    if (NPF > 0 && kt <= C::NS - 2 && younger >= 2 && nk > C::NS) wait_vmcnt<2 * kPer + NPF>();
    else if (younger >= 2) wait_vmcnt<2 * kPer>();
    else if (younger == 1) wait_vmcnt<kPer>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                          // publishes tile kt, frees slot (kt - 1) % NS
    if (kt + C::NS - 1 < nk) issue(kt + C::NS - 1, smem + ((kt + C::NS - 1) % C::NS) * C::kSlot);
  }
}

// ---- computing waves ----------------------------------------------------------------------------------------------
// first(): optionally, work of the computing wave issued right behind the FIRST K-tile's barrier (an epilogue's prefetch loads:
// the ring's prologue has landed by then, and nothing in this loop waits for vector memory)
template <class C, int TERMS, int MTW, int NTC, class FIRST = GatfNoPrefetch>
__device__ __forceinline__ void gatf_compute(const uint16_t* smem, int nk, int row0, int col0, int lane, f32x4 (&acc)[4][NTC],
                                             FIRST first = FIRST()) {
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ah[MTW], al[MTW], bh[NTC], bl[NTC];
  auto read_frags = [&](int kt) {
    const uint16_t* curA = smem + (kt % C::NS) * C::kSlot;
    const uint16_t* curB = curA + C::kPlanes * C::kTileA;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      ah[i] = planes_frag<KC>(curA, row0 + i * 16, lane);
      if (TERMS == 3) al[i] = planes_frag<KC>(curA + C::kTileA, row0 + i * 16, lane);
    }
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      bh[j] = planes_frag<KC>(curB, col0 + j * 16, lane);
      if (TERMS == 3) bl[j] = planes_frag<KC>(curB + C::kTileB, col0 + j * 16, lane);
    }
  };
  auto mma_tile = [&]() {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        if (TERMS == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();
    if (kt == 0) first();
    read_frags(kt);
    mma_tile();
  }
}

// ---- host side: the tile shape of a clip count (both kernels) ----------------------------------------------------
struct GatfShape { int mtiles16, nt, spt; };
inline bool gatf_shape(int B, int N, GatfShape* s) {
  if (N < 1 || N > 64) return false;
  if (N <= 16) { s->mtiles16 = 4; s->nt = 1; s->spt = 64 / N; }
  else if (N <= 20) { s->mtiles16 = 5; s->nt = 2; s->spt = 80 / N; }
  else if (N <= 32) { s->mtiles16 = 6; s->nt = 2; s->spt = 96 / N; }
  else if (N <= 48) { s->mtiles16 = 6; s->nt = 3; s->spt = 96 / N; }
  else { s->mtiles16 = 8; s->nt = 4; s->spt = 128 / N; }
  if (s->spt > B) s->spt = B;
  return s->spt >= 1;
}

}  // namespace dvgr
