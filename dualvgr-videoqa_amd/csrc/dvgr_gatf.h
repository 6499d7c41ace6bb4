// dvgr_gatf.h -- what the two fused GAT-view kernels share (dvgr_gatf.hip: projection + attention forward; dvgr_gatb.hip: the
// dz contraction + attention backward): the tile configuration, the loader waves' LDS-DMA ring and the computing waves' K loop
// on a tile of MTILES x 16 rows (whole samples) x F = 64 NTC columns (one head).  Structure of gemm_planes_kernel (dvgr_gemm.hip).
#pragma once
#include "dvgr_planes.h"
#include "dvgr_stages.h"

namespace dvgr {

// LO8: the lo plane of the B operand (the weights: 192 of a K-tile's 272 rows) arrives as 8 bits per element instead of 16 --
// q = round(lo / s_row) + 128, one fp32 scale per row, prepared with the parameter planes (dvgr_pack_lo8) -- and becomes the
// bf16 MFMA operand in the computing waves' registers (v_cvt_f32_ubyte*, one FMA, v_cvt_pk_bf16_f32 per pair).  The K loop runs at
// the rate its operand bytes arrive (every extra byte costs its share of time), the conversion's VALU work
// is free beside the MFMAs (experiments_r05.tsv, probe_lo_plane_conversion_valu_c2): 17.6 % fewer bytes per K-tile.  The lo
// plane carries the 9th .. 16th significant bit of a weight; 8 bits against the row's largest |lo| keep it to 2^-17 of the
// row's largest weight -- the grade of the bf16 lo plane itself.
// KB: the depth of a K-tile, 32 or 64.  A row piece of a 32-deep tile is 64 bytes -- half a 128-byte line -- and every row piece
// is one request to the L2: at C2 a launch's K loop asks for 155 G row pieces per second, 58 % of the L2's 269 G requests / s
// (128 channels x 2.1 GHz), while its BYTES are 29 % of the L2's rate; fewer bytes at the same piece count (LO8) buy nothing.
// 64-deep tiles move whole lines: half the requests for the same bytes, at the price of a 2-slot ring (68 KB per slot).
// EARLY (-DDVGR_GATF_EARLY=1, the 2-slot ring only; measured slower, kept as an experiment): a slot is handed back to the loaders
// as soon as every computing wave holds the K-tile's fragments in REGISTERS -- a second barrier per K-tile, in front of the MFMAs
// instead of behind them -- so that the registers are the ring's third stage and the DMA of K-tile kt + 2 is in flight while kt
// is multiplied.  Phase clocks at C2 (profiles/r05_kloop_decomposition.txt): the DMA alone 14.4 -> 11.8 us (its latency is hidden:
// 70 GB/s per CU, the L2's gather rate), the computing waves alone 12.0 -> 13.8 (all reads, then all MFMAs), together 18.2 against
// 16.5 -- LDS-DMA writes and fragment reads slow each other whenever both are in flight, and the early form keeps them so.
#ifndef DVGR_GATF_EARLY
#define DVGR_GATF_EARLY 0
#endif
// PF_LATE (-DDVGR_GATF_PF_LATE=0: the first form): the loader waves' epilogue loads (gatf_loader: NPF / pf) go out right behind the
// LAST K-tile's DMA instead of behind the ring's prologue -- younger than every K-tile still waited for, they never block the
// ring (on a 2-slot ring every wait is for everything older), and their bytes are under the last K-tile, not under the whole
// loop.  Phase clocks of the backward kernel, alternated twice on one device: 46.4 / 46.0 -> 45.5 / 45.6 us per workgroup.
#ifndef DVGR_GATF_PF_LATE
#define DVGR_GATF_PF_LATE 1
#endif
#ifndef DVGR_GATF_LOADER_PRIO
#define DVGR_GATF_LOADER_PRIO 0
#endif
#ifndef DVGR_GATF_MMA_PRIO
#define DVGR_GATF_MMA_PRIO 0
#endif
template <int TERMS, int MTILES, int NTC, bool LO8 = false, int KB_ = 32>
struct GatfCfg {
  static constexpr int KB = KB_;
  static_assert(KB == 32 || KB == 64, "K-tile depth");
  static_assert(!(LO8 && KB == 64), "the 8-bit plane is laid out for 32-deep tiles");
  static constexpr bool kLo8 = LO8 && TERMS == 3;
  static constexpr int kPlanes = TERMS == 3 ? 2 : 1;
  static constexpr int RP = 16 * MTILES;                  // rows of the A tile
  static constexpr int F = 64 * NTC;                      // columns of the tile = width of one head
  static constexpr int FS = F + 16;                       // shZ row stride (floats): 16 banks mod 64, the fp32 B fragment is conflict free
  static constexpr int CH = KB / 8;                       // 16-byte chunks per row of a plane tile
  static constexpr int RPI = 64 / CH;                     // rows per DMA instruction (1 KB)
  static constexpr int kTileA = RP * KB;                  // bf16 elements of one A plane tile
  static constexpr int kTileB = F * KB;
  static constexpr int kTileBlo = TERMS == 3 ? (kLo8 ? kTileB / 2 : kTileB) : 0;    // (in bf16-sized elements)
  static constexpr int kSlot = kPlanes * kTileA + kTileB + kTileBlo;
  static constexpr int NS = (4 * kSlot * 2 <= 150 * 1024) ? 4 : (3 * kSlot * 2 <= 150 * 1024) ? 3 : 2;   // ring slots
  static constexpr int IA = RP / RPI;                     // DMA instructions per plane and K-tile: A tile (64 chunks of 16 bytes each)
  static constexpr int IB = F / RPI;                      //   B tile (bf16 plane); the 8-bit lo plane: IB / 2 (32 rows each)
  static constexpr int IBlo = TERMS == 3 ? (kLo8 ? IB / 2 : IB) : 0;
  static constexpr int DTOT = kPlanes * IA + IB + IBlo;   // DMA instructions per K-tile, dealt round-robin to the 4 loader waves
  static constexpr int IHI = (DTOT + 3) / 4;              // instructions of loader wave w: IHI for w < DTOT % 4 (all of them when
  static constexpr int ILO = DTOT / 4;                    //   DTOT % 4 == 0), ILO otherwise
  static constexpr int MT0 = (MTILES + 1) / 2;            // row tiles of the row group 0 waves, MTILES - MT0 of the others
  static constexpr int kRingBytes = NS * kSlot * 2;
  static constexpr bool kEarly = DVGR_GATF_EARLY != 0 && NS == 2 && KB == 64;
  static constexpr int RD = kEarly ? NS + 1 : NS;         // K-tiles the ring holds: its slots (+ the computing waves' registers)
  static_assert(!kLo8 || (IB % 2) == 0, "the 8-bit plane moves 32 rows per DMA instruction");
};

// LDS image of a plane tile: [rows][CH chunks of 8 bf16], DMA lane-linear; position `slot` of a row holds source chunk
// gatf_swz(row, slot) (an involution), so a fragment read of chunk c goes to position gatf_swz(row, c).  32-deep: the swizzle of
// dvgr_planes.h (b128 reads of 16 rows conflict free); 64-deep: rows are 128 bytes = 32 banks, chunk ^ ((row >> 1) & 7) puts the
// 16 rows of a fragment read on 16 different 4-bank groups.
template <int KB>
__device__ __forceinline__ int gatf_swz(int row, int c) { return KB == 32 ? (c ^ ((-(row >> 2)) & 3)) : (c ^ ((row >> 1) & 7)); }
// lane (r, g) of an MFMA operand: k = 32 hh + 8 g .. + 7 of tile row idx16 + r
template <int KB>
__device__ __forceinline__ bf16x8 gatf_frag(const uint16_t* tile, int idx16, int lane, int hh) {
  const int row = idx16 + (lane & 15), c = 4 * hh + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(tile + row * KB + gatf_swz<KB>(row, c) * 8);
}

// What the loader streams: hi plane pointers of A (rows m0 .. of it, clamped to BN - 1) and B (rows brow0 .. brow0 + F - 1: forward
// head h of Wcat, backward the head's columns of P1^T), their bf16 lo planes at + x_lo / + w_lo elements, and -- LO8 -- the 8-bit
// lo plane of B (row-major bytes, same shape as B).
struct GatfOperands {
  const uint16_t* xa; int64_t x_lo;
  const uint16_t* wb; int64_t w_lo;
  const uint8_t* wq;
};

// ---- loader waves -------------------------------------------------------------------------------------------------
// The K-tile's DMA instructions as one flat list -- A hi row blocks, A lo, B hi, B lo (bf16: 16 rows of 64 bytes each; 8-bit: 32
// rows of 32 bytes, the two 16-byte halves of a row 16 positions apart so that the 8-byte fragment reads of 16 rows are 256
// contiguous bytes) -- of which wave w issues d = w + 4 k.
// NPF / pf: optionally, NPF further vector-memory instructions of this wave (plain loads into registers: an epilogue's inputs)
// issued by pf() right behind the ring's prologue, i.e. YOUNGER than the first NS - 1 K-tiles: the counted waits for those
// tiles leave them in flight (loads, stores and LDS-DMA share one in-order counter), every later wait covers them.
struct GatfNoPrefetch { __device__ __forceinline__ void operator()() const {} };
template <class C, int TERMS, int CNT, int NPF = 0, class PF = GatfNoPrefetch>
__device__ __forceinline__ void gatf_loader(const GatfOperands& op, uint16_t* smem, int brow0, int m0, int BN, int D, int nk,
                                            int wave, int lane, PF pf = PF()) {
#if defined(DVGR_CLK) && defined(DVGR_GATF_DBG_SAMETILE)    // (phase-clock experiments: every workgroup streams the SAME tiles -- all L2 hits)
  brow0 = 0; m0 = 0;
#endif
  const char* src[CNT];                                    // this lane's source address of instruction k at K-tile 0
  int step[CNT];                                           // bytes per K-tile: 64 (bf16 planes), 32 (8-bit plane)
  int dst[CNT];                                            // element offset of the instruction's 1 KB piece inside the slot
#pragma unroll
  for (int k = 0; k < CNT; ++k) {
    int d = wave + 4 * k;
    d = d < C::DTOT ? d : C::DTOT - 1;                     // (never issued: CNT is chosen per wave so that d < DTOT)
    const int S = lane;
    if (d < C::kPlanes * C::IA) {                          // A: plane d / IA, row block d % IA
      const int pl = d / C::IA, j = d - pl * C::IA;
      const int row = j * C::RPI + S / C::CH, slot = S % C::CH;
      const int chunk = gatf_swz<C::KB>(row, slot);
      int gi = m0 + row;
      gi = gi < BN ? gi : BN - 1;                          // rows past the matrix: any valid row (never stored)
      src[k] = reinterpret_cast<const char*>(op.xa + (pl ? op.x_lo : 0) + (int64_t)gi * D + chunk * 8);
      step[k] = 2 * C::KB;
      dst[k] = pl * C::kTileA + j * 512;
    } else if (d < C::kPlanes * C::IA + C::IB + (C::kLo8 ? 0 : C::IBlo)) {       // B, bf16 planes
      const int e = d - C::kPlanes * C::IA;
      const int pl = e / C::IB, j = e - pl * C::IB;
      const int row = j * C::RPI + S / C::CH, slot = S % C::CH;
      const int chunk = gatf_swz<C::KB>(row, slot);
      src[k] = reinterpret_cast<const char*>(op.wb + (pl ? op.w_lo : 0) + (int64_t)(brow0 + row) * D + chunk * 8);
      step[k] = 2 * C::KB;
      dst[k] = C::kPlanes * C::kTileA + pl * C::kTileB + j * 512;
    } else {                                               // B, 8-bit lo plane: 32 rows per instruction
      const int j8 = d - (C::kPlanes * C::IA + C::IB);
      const int grp = S >> 5, q = S & 31, half = q >> 4, rr = q & 15;
      const int row = 32 * j8 + 16 * grp + rr;
      src[k] = reinterpret_cast<const char*>(op.wq + (int64_t)(brow0 + row) * D + half * 16);
      step[k] = BK;
      dst[k] = C::kPlanes * C::kTileA + C::kTileB + j8 * 512;
    }
  }
  auto issue = [&](int kt, uint16_t* slot_base) {
#if defined(DVGR_CLK) && defined(DVGR_GATF_DBG_NODMA)       // (phase-clock experiments: the K loop without its DMA)
    return;
#endif
#pragma unroll
    for (int k = 0; k < CNT; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[k] + (int64_t)kt * step[k]),
                                       (__attribute__((address_space(3))) void*)(slot_base + dst[k]), 16, 0, 0);
  };
  if (DVGR_GATF_LOADER_PRIO) __builtin_amdgcn_s_setprio(DVGR_GATF_LOADER_PRIO);
#pragma unroll
  for (int s = 0; s < C::RD - 1; ++s)
    if (s < nk) issue(s, smem + (s % C::NS) * C::kSlot);
  constexpr bool kPfLate = DVGR_GATF_PF_LATE != 0 && NPF > 0;
  auto pf_now = [&]() {                                    // (program order IS the order the waits below count in)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pf();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  if (NPF > 0 && !kPfLate) pf_now();
  constexpr int kPer = CNT;                                // DMA instructions of this wave per K-tile
  constexpr int YM = C::RD - 2 < 2 ? C::RD - 2 : 2;        // K-tiles in flight behind the one waited for (at most)
  static_assert(YM * kPer + NPF < 64, "vmcnt is a 6-bit counter");
  bool pf_done = false;                                    // PF_LATE: the loads are in the queue, behind every K-tile
  for (int kt = 0; kt < nk; ++kt) {
    const int younger = (nk - 1 - kt) < YM ? (nk - 1 - kt) : YM;
    // (the NPF loads sit between K-tile RD - 2 and K-tile RD - 1 in the queue: younger than the tile waited for while kt <= RD - 2;
    // a K loop shorter than RD + 1 tiles simply waits for them as well)
    const bool pf_younger = NPF > 0 && !kPfLate && kt <= C::RD - 2 && younger == YM && nk > C::RD;
    if (kPfLate && pf_done) {
      if (younger >= 2) wait_vmcnt<2 * kPer + NPF>();
      else if (younger == 1) wait_vmcnt<kPer + NPF>();
      else wait_vmcnt<NPF>();
    } else if (pf_younger) wait_vmcnt<YM * kPer + NPF>();
    else if (younger >= 2) wait_vmcnt<2 * kPer>();
    else if (younger == 1) wait_vmcnt<kPer>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                          // publishes tile kt; !EARLY: frees slot (kt - 1) % NS
    if (C::kEarly) __builtin_amdgcn_s_barrier();           // EARLY: tile kt is in the computing waves' registers, its slot is free
    if (kt + C::RD - 1 < nk) {
      issue(kt + C::RD - 1, smem + ((kt + C::RD - 1) % C::NS) * C::kSlot);
      if (kPfLate && kt + C::RD - 1 == nk - 1) { pf_now(); pf_done = true; }
    }
  }
  if (kPfLate && !pf_done) pf_now();                       // (a K loop no longer than the ring's prologue)
  if (DVGR_GATF_LOADER_PRIO) __builtin_amdgcn_s_setprio(0);
}
// (the loader waves of a workgroup: wave w carries IHI instructions per K-tile when w < DTOT % 4 or DTOT % 4 == 0, else ILO)
template <class C, int TERMS, int NPF = 0, class PF = GatfNoPrefetch>
__device__ __forceinline__ void gatf_loader_wave(const GatfOperands& op, uint16_t* smem, int brow0, int m0, int BN, int D, int nk,
                                                 int w, int lane, PF pf = PF()) {
  if (C::IHI == C::ILO || w < (C::DTOT & 3)) gatf_loader<C, TERMS, C::IHI, NPF, PF>(op, smem, brow0, m0, BN, D, nk, w, lane, pf);
  else gatf_loader<C, TERMS, (C::ILO > 0 ? C::ILO : 1), NPF, PF>(op, smem, brow0, m0, BN, D, nk, w, lane, pf);
}

// ---- computing waves ----------------------------------------------------------------------------------------------
// first(): optionally, work of the computing wave issued right behind the FIRST K-tile's barrier (an epilogue's prefetch loads:
// the ring's prologue has landed by then, and nothing in this loop waits for vector memory)
// wscale (LO8): the per-row scales of the B operand's 8-bit lo plane, indexed by the tile's row (= output column).
template <class C, int TERMS, int MTW, int NTC, class FIRST = GatfNoPrefetch>
__device__ __forceinline__ void gatf_compute(const uint16_t* smem, int nk, int row0, int col0, int lane, f32x4 (&acc)[4][NTC],
                                             FIRST first = FIRST(), const float* wscale = nullptr) {
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NH = C::kEarly ? C::KB / 32 : 1;           // fragment sets held at once: EARLY every 32-deep half of the K-tile
  bf16x8 ah[NH][MTW], al[NH][MTW], bh[NH][NTC], bl[NH][NTC];
  float qs[NTC], qc[NTC];                                  // lo = q * s - 128 s for this lane's row of B fragment j
#pragma unroll
  for (int j = 0; j < NTC; ++j) {
    qs[j] = C::kLo8 ? wscale[col0 + 16 * j + (lane & 15)] : 0.f;
    qc[j] = -128.f * qs[j];
  }
#if defined(DVGR_CLK) && defined(DVGR_GATF_DBG_NOREADS)     // (phase-clock experiments: MFMAs on whatever the registers hold, no LDS reads)
#pragma unroll
  for (int s = 0; s < NH; ++s) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) { ah[s][i] = bf16x8{}; al[s][i] = bf16x8{}; asm volatile("" : "+v"(ah[s][i]), "+v"(al[s][i])); }
#pragma unroll
    for (int j = 0; j < NTC; ++j) { bh[s][j] = bf16x8{}; bl[s][j] = bf16x8{}; asm volatile("" : "+v"(bh[s][j]), "+v"(bl[s][j])); }
  }
#endif
  auto read_frags = [&](int kt, int hh, int set) {
#if defined(DVGR_CLK) && defined(DVGR_GATF_DBG_NOREADS)
    return;
#endif
    const uint16_t* curA = smem + (kt % C::NS) * C::kSlot;
    const uint16_t* curB = curA + C::kPlanes * C::kTileA;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      ah[set][i] = gatf_frag<C::KB>(curA, row0 + i * 16, lane, hh);
      if (TERMS == 3) al[set][i] = gatf_frag<C::KB>(curA + C::kTileA, row0 + i * 16, lane, hh);
    }
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      bh[set][j] = gatf_frag<C::KB>(curB, col0 + j * 16, lane, hh);
      if (TERMS == 3 && !C::kLo8) bl[set][j] = gatf_frag<C::KB>(curB + C::kTileB, col0 + j * 16, lane, hh);
      if (C::kLo8) {
        // 8 bytes = k 8 g .. 8 g + 7 of row col0 + 16 j + r: position ((row >> 4) * 32 + (g >> 1) * 16 + (row & 15)) * 16 + (g & 1) * 8
        const int r = lane & 15, g = lane >> 4;
        const uint8_t* t8 = reinterpret_cast<const uint8_t*>(curB + C::kTileB);
        const uint2 q = *reinterpret_cast<const uint2*>(t8 + ((((col0 >> 4) + j) * 32 + (g >> 1) * 16 + r) << 4) + ((g & 1) << 3));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 w;
        w[0] = pack_bf16(__builtin_fmaf((float)(q.x & 0xFFu), qs[j], qc[j]), __builtin_fmaf((float)((q.x >> 8) & 0xFFu), qs[j], qc[j]));
        w[1] = pack_bf16(__builtin_fmaf((float)((q.x >> 16) & 0xFFu), qs[j], qc[j]), __builtin_fmaf((float)(q.x >> 24), qs[j], qc[j]));
        w[2] = pack_bf16(__builtin_fmaf((float)(q.y & 0xFFu), qs[j], qc[j]), __builtin_fmaf((float)((q.y >> 8) & 0xFFu), qs[j], qc[j]));
        w[3] = pack_bf16(__builtin_fmaf((float)((q.y >> 16) & 0xFFu), qs[j], qc[j]), __builtin_fmaf((float)(q.y >> 24), qs[j], qc[j]));
        bl[set][j] = __builtin_bit_cast(bf16x8, w);
      }
    }
  };
  auto mma_tile = [&](int set) {
#if defined(DVGR_CLK) && defined(DVGR_GATF_DBG_NOMFMA)      // (phase-clock experiments: the fragment reads without their MFMAs)
#pragma unroll
    for (int i = 0; i < MTW; ++i) asm volatile("" ::"v"(ah[set][i]), "v"(al[set][i]));
#pragma unroll
    for (int j = 0; j < NTC; ++j) asm volatile("" ::"v"(bh[set][j]), "v"(bl[set][j]));
    return;
#endif
    if (DVGR_GATF_MMA_PRIO) __builtin_amdgcn_s_setprio(DVGR_GATF_MMA_PRIO);
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        if (TERMS == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[set][i], bh[set][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][i], bl[set][j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][i], bh[set][j], acc[i][j], 0, 0, 0);
      }
    if (DVGR_GATF_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
  };
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();
    if (kt == 0) first();
#if defined(DVGR_CLK) && defined(DVGR_GATF_DBG_NOMMA)       // (phase-clock experiments: the K loop without fragment reads and MFMAs)
    if (C::kEarly) __builtin_amdgcn_s_barrier();
    continue;
#endif
    if (C::kEarly) {
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) read_frags(kt, hh, hh);
      __builtin_amdgcn_sched_barrier(0);                   // (the scheduler would sink the second half's reads between the MFMAs)
#if DVGR_GATF_EARLY == 2                                   // (A/B form: the first half's MFMAs in front of the hand-back)
      mma_tile(0);
      __builtin_amdgcn_sched_barrier(0);
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment of the K-tile is in registers:
      __builtin_amdgcn_s_barrier();                        // the slot goes back to the loaders
      __builtin_amdgcn_sched_barrier(0);
#if DVGR_GATF_EARLY != 2
      mma_tile(0);
#endif
#pragma unroll
      for (int hh = 1; hh < NH; ++hh) mma_tile(hh);
    } else {
#pragma unroll
      for (int hh = 0; hh < C::KB / 32; ++hh) {
        read_frags(kt, hh, 0);
        mma_tile(0);
      }
    }
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

// Half-height tiles (round 6, the 64-deep bf16x3 kernels only): where the 64-row tile (N <= 16) leaves most of the chip idle --
// C4's 32 QA pairs of 16 clips are 8 row tiles x 8 heads x 2 modalities = 128 workgroups, and every epilogue then runs on half
// the compute units -- a tile of 32 rows (32 / N samples) doubles the workgroups: per workgroup 56 KB of operand planes per K-tile
// instead of 64 (the weight panel is streamed by twice as many workgroups: +75 % bytes over the chip, on twice the compute
// units) and half the epilogue.  Taken when the full-height grid has < 200 workgroups and the half-height one fits one round.
// `s`: in = gatf_shape's choice, out = the half-height shape.
inline bool gatf_half_shape(int B, int N, GatfShape* s) {
  if (s->mtiles16 != 4 || s->nt != 1 || N > 16) return false;
  int spt2 = 32 / N;
  if (spt2 < 1) return false;
  if (spt2 > B) spt2 = B;
  const int wg_full = ((B + s->spt - 1) / s->spt) * kHeads * 2, wg_half = ((B + spt2 - 1) / spt2) * kHeads * 2;
  if (wg_full >= 200 || wg_half <= wg_full || wg_half > 256) return false;
  s->mtiles16 = 2;
  s->spt = spt2;
  return true;
}

}  // namespace dvgr
