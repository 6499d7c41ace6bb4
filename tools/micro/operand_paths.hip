// How fast can a compute unit take in the operands of the fused GAT-view K loop (C2: 80 x 192 x 768, bf16 hi / lo planes, one
// workgroup per CU, 256 workgroups) -- by WHICH path?  DESIGN.md section 3.1: the shipped loop moves all 68 KB of a 64-deep K-tile
// from L2 into LDS by LDS-DMA, and that path delivers 66-73 GB/s per CU (MI355X_MICROARCH.md "Indexed rows: gather into LDS"),
// i.e. 11.8 us per launch while the MFMAs need 7.2.  The B operand (the weights: 48 of the 68 KB) is needed by ONE wave when
// four waves own 80 x 48 each, so it could go L2 -> registers directly, from a copy of the weight planes laid out in MFMA
// fragment order (every wave instruction = 1 KB of contiguous memory).  This probe times, on the real shapes and the real XCD map:
//   mode 0  all 68 KB per K-tile by LDS-DMA, 2-slot ring, one barrier per K-tile, nobody reads (the shipped loop's DMA alone)
//   mode 1  A (20 KB) by LDS-DMA into a 4-slot ring + B (48 KB) by global_load_dwordx4 into registers, DEPTH 32-deep half tiles
//           in flight per wave; no LDS reads, no MFMAs (transport alone)
//   mode 2  mode 1 + the A fragment reads + the 3-term bf16 MFMAs: the whole candidate K loop
//   mode 3  mode 2 without the B loads (registers as they are): LDS + MFMA alone
//   mode 4  only the B register loads (no DMA at all)
// and prints microseconds per launch (hipEvents over 40 back-to-back launches) minus those of an empty launch of the same geometry.
//   hipcc --offload-arch=gfx950 -O3 operand_paths.hip -o operand_paths.bin ; ./operand_paths.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../dualvgr-videoqa_amd/csrc/dvgr_gatf.h"      // mode 7: the SHIPPED K loop itself (loader ring + computing waves), in this harness

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 768, N = 20, B = 64, BN = B * N, HEADS = 8, F = 192, RP = 80, KB = 64, NK = D / KB, MT = 5, NTC = 3;
constexpr int kTileA = RP * KB, kTileB = F * KB;

template <int NN>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NN) : "memory"); }
__device__ __forceinline__ int swz64(int row, int c) { return c ^ ((row >> 1) & 7); }
__device__ __forceinline__ void dma16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

struct Args {
  const uint16_t* xa; int64_t x_lo;       // activations: planes (BN, D) row-major, lo plane at + x_lo elements
  const uint16_t* wb; int64_t w_lo;       // weights: planes (2 modalities x 8 heads x 192, D) row-major
  const uint16_t* wf;                     // weights in fragment order: [mod][head][wave 4][half tile 24][j 3][plane 2][lane 64][8]
  float* sink; int rep;
};

// mode 0: the shipped scheme's DMA: 4 loader waves, 68 instructions per K-tile (A hi 5, A lo 5, B hi 12, B lo 12 ... in 1 KB pieces
// of 8 rows x 128 bytes), 2-slot ring
// ASTAGE 1: the A tile goes global -> registers -> ds_write_b128 (three register sets per loader wave) instead of by LDS-DMA
// ORD 1: the three MFMAs of one accumulator are not issued back to back (term-major over the three column tiles)
template <int MODE, int DEPTH, int THREADS = 512, int ASTAGE = 0, int ORD = 0>
__global__ __launch_bounds__(THREADS) void k_paths(const Args a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtiles = B / 4, total = mtiles * HEADS * 2, chunk = (int)gridDim.x >> 3;
  const int idx = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (idx >= total) return;
  const int mod = idx / (mtiles * HEADS), rem = idx - mod * mtiles * HEADS;
  const int hm = mtiles >> 1, blk = rem / (4 * hm), e = rem - blk * 4 * hm;
  const int h = 4 * (blk >> 1) + e / hm, mt = (blk & 1) * hm + e % hm;
  const int m0 = mt * RP;
  const uint16_t* xa = a.xa + (int64_t)mod * 2 * BN * D;            // (each modality: hi then lo plane)
  const uint16_t* wb = a.wb + ((int64_t)mod * HEADS + h) * F * D;
  const bool loader = MODE == 8 ? wave >= 8 : (wave >= 4 && wave < 8);
  const bool idle = MODE == 8 ? false : wave >= 8;                                      // (THREADS 768: four more waves that only meet the barriers, as epilogue-only waves would)
  f32x4 acc[MT][NTC];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 keep = {0u, 0u, 0u, 0u};

  if (a.rep == 0) return;
  if constexpr (MODE == 9) {                                        // the MFMAs alone: 4 waves (one per SIMD) x 1080, operands in registers
    if (wave < 4) {
      bf16x8 fa[MT], fb[NTC];
#pragma unroll
      for (int i = 0; i < MT; ++i) { u32x4 t = {(uint32_t)lane, 1u, 2u, (uint32_t)i}; asm volatile("" : "+v"(t)); fa[i] = __builtin_bit_cast(bf16x8, t); }
#pragma unroll
      for (int j = 0; j < NTC; ++j) { u32x4 t = {(uint32_t)lane, 3u, 4u, (uint32_t)j}; asm volatile("" : "+v"(t)); fb[j] = __builtin_bit_cast(bf16x8, t); }
      for (int kk = 0; kk < 2 * NK; ++kk) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  if constexpr (MODE == 7) {
    using C = dvgr::GatfCfg<3, 5, 3, false, 64>;
    static_assert(THREADS == 768, "the shipped loop runs 12 waves");
    dvgr::f32x4 acc7[4][3];
    const dvgr::GatfOperands op{xa, a.x_lo, a.wb + (int64_t)mod * HEADS * F * D, a.w_lo, nullptr};
    if (wave >= 8) dvgr::gatf_loader_wave<C, 3>(op, smem, h * F, m0, BN, D, NK, wave - 8, lane);
    else if ((wave >> 2) == 0) dvgr::gatf_compute<C, 3, C::MT0, 3>(smem, NK, 0, (wave & 3) * 48, lane, acc7);
    else dvgr::gatf_compute<C, 3, 5 - C::MT0, 3>(smem, NK, 16 * C::MT0, (wave & 3) * 48, lane, acc7);
    __builtin_amdgcn_s_barrier();
    float s7 = 0.f;
    if (wave < 8) {
#pragma unroll
      for (int i = 0; i < ((wave >> 2) == 0 ? C::MT0 : 5 - C::MT0); ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s7 += acc7[i][j][0] + acc7[i][j][1] + acc7[i][j][2] + acc7[i][j][3];
    }
    if (s7 == 1.2345e-30f) a.sink[blockIdx.x * THREADS + tid] = s7;
    return;
  }                                         // (the launch floor: an empty launch of the same geometry)
  {
    if (MODE == 9) {
    } else if (MODE == 0) {
      constexpr int kSlot = 2 * kTileA + 2 * kTileB;                // elements: 68 KB
      if (loader) {
        const int w = wave - 4;
        constexpr int DTOT = 20 + 24 + 24, CNT = (DTOT + 3) / 4;    // 1 KB pieces of 8 rows: A 10 per plane, B 24 per plane
        const char* src[CNT]; int dst[CNT]; bool on[CNT];
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
          int d = w + 4 * k;
          on[k] = d < DTOT;
          d = on[k] ? d : DTOT - 1;
          const int row8 = lane >> 3, slot = lane & 7;
          if (d < 20) {
            const int pl = d / 10, j = d - pl * 10, row = j * 8 + row8;
            src[k] = (const char*)(xa + (pl ? a.x_lo : 0) + (int64_t)(m0 + row) * D + swz64(row, slot) * 8);
            dst[k] = pl * kTileA + j * 512;
          } else {
            const int e2 = d - 20, pl = e2 / 24, j = e2 - pl * 24, row = j * 8 + row8;
            src[k] = (const char*)(wb + (pl ? a.w_lo : 0) + (int64_t)row * D + swz64(row, slot) * 8);
            dst[k] = 2 * kTileA + pl * kTileB + j * 512;
          }
        }
        auto issue = [&](int kt, uint16_t* base) {
#pragma unroll
          for (int k = 0; k < CNT; ++k)
            if (on[k]) dma16(src[k] + (int64_t)kt * 2 * KB, base + dst[k]);
        };
        issue(0, smem);
        for (int kt = 0; kt < NK; ++kt) {
          wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          if (kt + 1 < NK) issue(kt + 1, smem + ((kt + 1) & 1) * kSlot);
        }
      } else {
        for (int kt = 0; kt < NK; ++kt) __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_s_barrier();
    } else {
      constexpr int NS = 4, kSlot = 2 * kTileA;                     // A ring: 4 slots of 20 KB
      if (loader) {
        const int w = MODE == 8 ? wave - 8 : wave - 4;
        constexpr int DTOT = 20, CNT = 5;
        const char* src[CNT]; int dst[CNT];
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
          const int d = w + 4 * k, row8 = lane >> 3, slot = lane & 7;
          const int pl = d / 10, j = d - pl * 10, row = j * 8 + row8;
          src[k] = (const char*)(xa + (pl ? a.x_lo : 0) + (int64_t)(m0 + row) * D + swz64(row, slot) * 8);
          dst[k] = pl * kTileA + j * 512;
        }
        auto issue = [&](int kt, uint16_t* base) {
          if (MODE == 4) return;
#pragma unroll
          for (int k = 0; k < CNT; ++k) dma16(src[k] + (int64_t)kt * 2 * KB, base + dst[k]);
        };
        if (ASTAGE == 0) {
          issue(0, smem); issue(1, smem + kSlot); issue(2, smem + 2 * kSlot);
          for (int kt = 0; kt < NK; ++kt) {
            const int younger = (NK - 1 - kt) < 2 ? (NK - 1 - kt) : 2;
            if (younger == 2) wait_vmcnt<2 * CNT>(); else if (younger == 1) wait_vmcnt<CNT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                             // publishes tile kt, frees slot (kt - 1) % 4
            if (kt + 3 < NK) issue(kt + 3, smem + ((kt + 3) % NS) * kSlot);
          }
        } else {
          u32x4 ra[3][CNT];
          auto ld = [&](int kt, int set) {
#pragma unroll
            for (int k = 0; k < CNT; ++k) ra[set][k] = *reinterpret_cast<const u32x4*>(src[k] + kt * 2 * KB);
          };
          ld(0, 0); ld(1, 1); ld(2, 2);
#pragma unroll
          for (int kt = 0; kt < NK; ++kt) {
            const int younger = (NK - 1 - kt) < 2 ? (NK - 1 - kt) : 2;
            if (younger == 2) wait_vmcnt<2 * CNT>(); else if (younger == 1) wait_vmcnt<CNT>(); else wait_vmcnt<0>();
            uint16_t* base = smem + (kt % NS) * kSlot;
#pragma unroll
            for (int k = 0; k < CNT; ++k) *reinterpret_cast<u32x4*>(base + dst[k] + lane * 8) = ra[kt % 3][k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 3 < NK) ld(kt + 3, kt % 3);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                             // publishes tile kt
          }
        }
      } else if (idle) {
        for (int kt = 0; kt < NK; ++kt) __builtin_amdgcn_s_barrier();
      } else {
        // this wave's stream of B fragments: half tile kk (32 deep) = 6 pieces of 1 KB [j][plane], contiguous
        // (a uniform base in SGPRs + one 32-bit lane offset + an immediate: the saddr form of global_load -- with per-lane 64-bit
        // addresses the compiler keeps all 144 of them in registers and spills)
        const int cg = wave & 3, kh = MODE == 8 ? (wave >> 2) : 0;  // mode 8: waves cg and cg + 4 share the 80 x 48 tile, even / odd half tiles of K
        const char* wfb = reinterpret_cast<const char*>(a.wf) + ((((int64_t)mod * HEADS + h) * 4 + cg) * (2 * NK)) * (NTC * 2 * 1024);
        const uint32_t voff = lane * 16;
        u32x4 bq[DEPTH][NTC][2];
        auto fetch = [&](int kk, int st) {
          int soff = kk * (NTC * 2 * 1024);
          asm volatile("" : "+s"(soff));                            // (opaque: formed here, not hoisted out of the pass loop)
          const char* sb = wfb + soff;
#pragma unroll
          for (int j = 0; j < NTC; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
              bq[st][j][p] = *reinterpret_cast<const u32x4*>(sb + voff + (j * 2 + p) * 1024);
        };
        if (MODE != 3) {
#pragma unroll
          for (int s = 0; s < DEPTH; ++s) fetch(MODE == 8 ? 2 * s + kh : s, s);
        } else {
#pragma unroll
          for (int s = 0; s < DEPTH; ++s)
#pragma unroll
            for (int j = 0; j < NTC; ++j) { bq[s][j][0] = keep; bq[s][j][1] = keep; asm volatile("" : "+v"(bq[s][j][0]), "+v"(bq[s][j][1])); }
        }
        constexpr int KSTEP = MODE == 8 ? 2 : 1;                      // half tiles between two of this wave's
#pragma unroll
        for (int q = 0; q < 2 * NK / KSTEP; ++q) {
          const int kk = MODE == 8 ? 2 * q + kh : q;
          const int kt = kk >> 1, hh = kk & 1, st = q % DEPTH;
          if (MODE == 8 || hh == 0) __builtin_amdgcn_s_barrier();
          if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int j = 0; j < NTC; ++j) { keep ^= bq[st][j][0]; keep ^= bq[st][j][1]; }
          } else {
            const uint16_t* curA = smem + (kt % NS) * kSlot;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const int row = i * 16 + (lane & 15), c = 4 * hh + (lane >> 4);
              const bf16x8 ah = *reinterpret_cast<const bf16x8*>(curA + row * KB + swz64(row, c) * 8);
              const bf16x8 al = *reinterpret_cast<const bf16x8*>(curA + kTileA + row * KB + swz64(row, c) * 8);
              if (ORD == 0) {
#pragma unroll
                for (int j = 0; j < NTC; ++j) {
                  const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[st][j][0]), bl = __builtin_bit_cast(bf16x8, bq[st][j][1]);
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[i][j], 0, 0, 0);
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[i][j], 0, 0, 0);
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i][j], 0, 0, 0);
                }
              } else {
#pragma unroll
                for (int j = 0; j < NTC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, bq[st][j][0]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NTC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bq[st][j][1]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NTC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bq[st][j][0]), acc[i][j], 0, 0, 0);
              }
            }
          }
          if (MODE != 3 && q + DEPTH < 2 * NK / KSTEP) { __builtin_amdgcn_sched_barrier(0); fetch(kk + KSTEP * DEPTH, st); __builtin_amdgcn_sched_barrier(0); }
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  float s = __uint_as_float(keep[0] ^ keep[1] ^ keep[2] ^ keep[3]);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 1.2345e-30f) a.sink[blockIdx.x * THREADS + tid] = s;
}

template <int MODE, int DEPTH, int THREADS = 512, int ASTAGE = 0, int ORD = 0>
static float time_mode(const Args& base, int rep, size_t lds, hipStream_t st) {
  Args a = base; a.rep = rep;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_paths<MODE, DEPTH, THREADS, ASTAGE, ORD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t t0, t1;
  (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 5; ++i) k_paths<MODE, DEPTH, THREADS, ASTAGE, ORD><<<256, THREADS, lds, st>>>(a);
  (void)hipStreamSynchronize(st);
  const int L = 40;
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(t0, st);
    for (int i = 0; i < L; ++i) k_paths<MODE, DEPTH, THREADS, ASTAGE, ORD><<<256, THREADS, lds, st>>>(a);
    (void)hipEventRecord(t1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t0, t1);
    best = ms < best ? ms : best;
  }
  return best * 1e3f / L;
}

static float g_floor = 0.f;
template <int MODE, int DEPTH, int THREADS = 512, int ASTAGE = 0, int ORD = 0>
static void report(const char* what, const Args& a, size_t lds, hipStream_t st, double kb_per_pass) {
  if (g_floor == 0.f) g_floor = time_mode<MODE, DEPTH, THREADS, ASTAGE, ORD>(a, 0, lds, st);
  const float t1 = time_mode<MODE, DEPTH, THREADS, ASTAGE, ORD>(a, 1, lds, st);
  const double per = t1 - g_floor;
  printf("mode %d depth %d stageA %d ord %d  %-76s %6.2f us per launch - %4.2f floor = %6.2f us = %5.1f GB/s per CU, %5.2f TB/s chip\n", MODE, DEPTH, ASTAGE, ORD, what, t1, g_floor, per,
         kb_per_pass * 1024.0 / per * 1e-3, kb_per_pass * 1024.0 * 256 / per * 1e-6);
}

int main() {
  const size_t xe = (size_t)2 * 2 * BN * D, we = (size_t)2 * 2 * HEADS * F * D;
  uint16_t *x, *w, *wf; float* sink;
  CK(hipMalloc(&x, xe * 2)); CK(hipMalloc(&w, we * 2)); CK(hipMalloc(&wf, we * 2)); CK(hipMalloc(&sink, 256 * 768 * 4));
  std::vector<uint16_t> h(xe > we ? xe : we);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 20 & 0x3ff));     // bf16 values near 0.01
  CK(hipMemcpy(x, h.data(), xe * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), we * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(wf, h.data(), we * 2, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  // activations: [mod][plane][BN][D]; weights row-major: [mod][head][F][D] hi, then the lo planes behind all of them
  Args a{x, (int64_t)BN * D, w, (int64_t)2 * HEADS * F * D, wf, sink, 1};
  const double kt_all = 12 * 68.0, kt_a = 12 * 20.0, kt_b = 12 * 48.0;
  printf("C2 fused GAT-view K loop operands: 80 x 192 x 768, bf16 hi + lo planes, 256 workgroups (one per CU), 12 K-tiles of 64\n");
  report<7, 1, 768>("the SHIPPED K loop (dvgr_gatf.h: 8 computing + 4 loader waves, all by LDS-DMA, 2 slots of 68 KB)", a, 150 * 1024, st, kt_all);
  report<8, 2, 768>("split-K pairs: 8 computing waves (80 x 48, even / odd half tiles) + 4 loaders (A by LDS-DMA), depth 2", a, 128 * 1024, st, kt_all);
  report<8, 3, 768>("split-K pairs, depth 3", a, 128 * 1024, st, kt_all);
  report<8, 2, 768, 1>("split-K pairs, depth 2, A through registers", a, 128 * 1024, st, kt_all);
  report<8, 2, 768, 0, 1>("split-K pairs, depth 2, MFMAs term-major", a, 128 * 1024, st, kt_all);
  report<9, 1>("the MFMAs alone: 4 waves x 1080 v_mfma_f32_16x16x32_bf16, operands in registers, no memory, no barriers", a, 128 * 1024, st, kt_all);
  report<0, 1>("all 68 KB / K-tile by LDS-DMA, 2-slot ring, no readers (shipped loop's DMA)", a, 136 * 1024, st, kt_all);
  report<4, 4>("B only: 48 KB / K-tile L2 -> registers (fragment-order planes), no DMA", a, 80 * 1024 + 48 * 1024, st, kt_b);
  report<4, 6>("B only, 6 half tiles in flight", a, 128 * 1024, st, kt_b);
  report<1, 3>("A 20 KB by LDS-DMA (4 slots) + B 48 KB to registers, transport alone", a, 128 * 1024, st, kt_all);
  report<1, 4>("same, 4 half tiles of B in flight per wave", a, 128 * 1024, st, kt_all);
  report<1, 6>("same, 6 half tiles in flight", a, 128 * 1024, st, kt_all);
  report<3, 2>("A by LDS-DMA + A fragment reads + MFMAs, no B loads (LDS + MFMA alone)", a, 128 * 1024, st, kt_a);
  report<2, 2>("whole candidate K loop, 2 half tiles of B in flight", a, 128 * 1024, st, kt_all);
  report<2, 3>("whole candidate K loop, 3 half tiles in flight", a, 128 * 1024, st, kt_all);
  report<2, 4>("whole candidate K loop, 4 half tiles in flight", a, 128 * 1024, st, kt_all);
  report<2, 6>("whole candidate K loop, 6 half tiles in flight", a, 128 * 1024, st, kt_all);
  report<2, 2, 768>("whole candidate K loop, 2 half tiles in flight, 12-wave workgroup (168 registers)", a, 128 * 1024, st, kt_all);
  report<1, 2, 768>("transport alone, 2 half tiles in flight, 12-wave workgroup", a, 128 * 1024, st, kt_all);
  report<1, 3, 512, 1>("transport alone, A through registers + ds_write, B to registers", a, 128 * 1024, st, kt_all);
  report<3, 2, 512, 0, 1>("LDS + MFMA alone, MFMAs term-major", a, 128 * 1024, st, kt_a);
  report<3, 2, 512, 1, 1>("LDS + MFMA alone, MFMAs term-major, A through registers", a, 128 * 1024, st, kt_a);
  report<2, 2, 512, 0, 1>("whole loop, depth 2, A by LDS-DMA, MFMAs term-major", a, 128 * 1024, st, kt_all);
  report<2, 2, 512, 1, 0>("whole loop, depth 2, A through registers", a, 128 * 1024, st, kt_all);
  report<2, 2, 512, 1, 1>("whole loop, depth 2, A through registers, MFMAs term-major", a, 128 * 1024, st, kt_all);
  report<2, 3, 512, 1, 1>("whole loop, depth 3, A through registers, MFMAs term-major", a, 128 * 1024, st, kt_all);
  report<2, 4, 512, 1, 1>("whole loop, depth 4, A through registers, MFMAs term-major", a, 128 * 1024, st, kt_all);
  report<2, 2, 768, 1, 1>("whole loop, depth 2, A through registers, MFMAs term-major, 12-wave workgroup", a, 128 * 1024, st, kt_all);
  return 0;
}
