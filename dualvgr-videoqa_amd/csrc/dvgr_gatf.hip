// dvgr_gatf.hip -- the fused GAT-view kernel (SURVEY.md section 7.2, K2 `gat_pair`): projection, attention logits, softmax
// over the clips, punish-scaled aggregation, ELU and both dropout sites of ONE head of ONE modality for a tile of WHOLE
// samples, in one launch.  Reference: model/GraphNN.py:95-111 (PunishGraphAttentionLayer.forward, factorised logits as in
// SURVEY.md Appendix A.3) and :174-178 (punishGAT: x-site dropout -- the dropped copies arrive pre-materialised as bf16
// planes -- heads concatenated on dim 2, out-site dropout).
//
// Why one kernel: as two launches (gemm_planes_kernel<TAG_FWD_GAT> then k_gat_attn_fwd) the projection Wh (B N x 8F fp32,
// 15.7 MB at C2) is written by the contraction's generic epilogue, re-read by 1024 attention workgroups through L2 / HBM, and
// the two launches pay two ramps, two drains and a boundary.  The seam is TILE-LOCAL: head h of sample b needs only that
// sample's N rows x that head's F columns of Wh, so with an output tile of SPT whole samples x one head (16 MTILES rows x F
// columns) the attention is the contraction's epilogue and needs no grid-wide synchronisation.
//
// Work decomposition (C2: B = 64, N = 20, F = 192): SPT = 4 samples = 80 rows per tile -> 16 row tiles x 8 heads x 2
// modalities = 256 workgroups = one per compute unit.  12 waves per workgroup:
//   waves 8-11  loaders: LDS-DMA (global_load_lds_dwordx4) of the bf16 hi / lo planes of a 32-deep K-tile -- A = 16 MTILES
//               rows of the (dropped) layer input, B = the head's F rows of Wcat -- into an NS-slot ring, counted vmcnt, one
//               raw s_barrier per K-tile (the structure of gemm_planes_kernel, dvgr_gemm.hip);
//   waves 0-7   MFMA (v_mfma_f32_16x16x32_bf16, bf16x3 = hi.hi + hi.lo + lo.hi): wave w owns the F / 4 columns of column
//               group w & 3 and the row tiles of row group w >> 2 (ceil / floor halves of MTILES: waves w and w + 4 share a
//               SIMD, so every SIMD carries one larger and one smaller block).
// Epilogue, all 12 waves, in the ring's (now idle) LDS:
//   E1  accumulators + bias -> shZ [rows][F + 16] (fp32); the loaders fetch a_l, a_r, the punish gates s and clear shA
//   E2  Wh tile -> global (saved for the backward) as whole 16-byte row pieces; u_i = a_l . Wh_i, v_i = a_r . Wh_i (a wave per row)
//   E3  softmax_j LeakyReLU(u_i + v_j + c) (a wave per row), alpha-site dropout, times s_j -> shA  (punish AFTER the
//       logits, GraphNN.py:102-104: alpha . (s * Wh) = (alpha * s^T) . Wh)
//   E4  aggregation on the fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32): one (sample, 16-column strip) per wave at a time;
//       the wave overwrites its strip of shZ with the result (nobody else reads those columns of that sample)
//   E5  ELU, out-site dropout, fp32 + bf16 planes as 16-byte vectors into the list output (common heads -> com_*, specific
//       heads -> *_fusion)
// What it saves per launch pair at C2: the generic epilogue's transposition, 15.7 MB of Wh re-read, one launch boundary
// and the attention kernel's load phase.  Layer 0 keeps the two-launch form when the side stream is in use: its punish gate
// depends on the query chain (S0), which the separate Wh contraction hides.
#include <type_traits>

#include "dvgr_gatf.h"

#define DVGR_LAUNCH_CHECK() return (int)hipGetLastError()

namespace dvgr {

typedef float gatf_f32x4 __attribute__((ext_vector_type(4)));

// Phase clocks (debug builds only, `make clk`; tools/gatf_clocks.py): wave 0 of every workgroup stamps the 100 MHz wall clock
#ifdef DVGR_CLK
__device__ long long g_gatf_clk[1024 * 8];
#define GFCLK(k)                                                                                              \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_gatf_clk[blockIdx.x * 8 + (k)] = wall_clock64();            \
  } while (0)
#else
#define GFCLK(k) do { } while (0)
#endif

// grid: 8 * ceil(total / 8) workgroups, total = mtiles * 8 heads * 2 modalities.  Workgroups b and b + 8 share an XCD
// (round-robin placement: speed only), so XCD x takes the contiguous chunk x of the tile list, decoded below as 4 heads x half of
// the row tiles.
template <int TERMS, int MTILES, int NT, int NTC, bool LO8, int KB>
__global__ __launch_bounds__(768) void k_gat_fused_fwd(const GatFusedArgs a) {
  using C = GatfCfg<TERMS, MTILES, NTC, LO8, KB>;
  extern __shared__ __attribute__((aligned(16))) uint16_t gatf_smem[];
  uint16_t* smem = gatf_smem;
  const int N = a.at.N, D = a.at.D, B = a.at.B, BN = B * N;
  constexpr int F = C::F, FS = C::FS, RP = C::RP, F4 = F / 4, JT = F / 16;
  const int HF = kHeads * F;
  const int SPT = a.spt;                                   // samples per tile (SPT * N <= RP)
  const int mtiles = (B + SPT - 1) / SPT;
  const int total = mtiles * kHeads * 2;
  const int chunk = (int)gridDim.x >> 3;
  const int idx = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (idx >= total) return;                                // uniform per workgroup, before any barrier
  const int mod = idx / (mtiles * kHeads);
  const int rem = idx - mod * mtiles * kHeads;
  // An XCD's chunk of the tile list (workgroups b and b + 8 share an XCD) = 4 heads x HALF of the row tiles where the row-tile
  // count is even: per L2 four B panels (2.4 MB of planes at C2) and half a view's A panels (2.0 MB) = 4.3 MB, against 5.1 MB
  // with 2 heads x all row tiles (round 4) -- fabric bytes of the launch -16 %, step -0.6 % at C2, -0.4 % at C3
  // (experiments_r05.tsv, xcd_map_4heads_x_half_rows_*).  Speed only: any mapping is correct.
  int h, mt;
  if ((mtiles & 1) == 0) {
    const int hm = mtiles >> 1, blk = rem / (4 * hm), e = rem - blk * 4 * hm;
    h = 4 * (blk >> 1) + e / hm;
    mt = (blk & 1) * hm + e % hm;
  } else { h = rem / mtiles; mt = rem - h * mtiles; }
  const int g = h >> 2, hl = h & 3;
  const int b0 = mt * SPT;                                 // first sample of the tile
  const int m0 = b0 * N;                                   // first row
  const int spt_valid = (B - b0) < SPT ? (B - b0) : SPT;
  const int rows_valid = spt_valid * N;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave12 >= 8;
  const int nk = D / C::KB;

  f32x4 acc[4][NTC];                                       // (row group 0 uses MT0 of the 4 row slots, row group 1 MTILES - MT0)
  int row0 = 0;
  const int col0 = (wave12 & 3) * 16 * NTC;
  // what the epilogue needs from global memory is fetched BEFORE the K loop (it lands under it): the computing waves' bias
  // elements, and by the loader threads one element each of a_l, a_r and of the punish gates
  float bv[NTC], pre_al = 0.f, pre_ar = 0.f, pre_s = 0.f;
#pragma unroll
  for (int j = 0; j < NTC; ++j) bv[j] = 0.f;
  GFCLK(0);
  if (loader) {
    const int tl = tid - 512;                              // 0 .. 255
    const float* ga = a.at.gat_a[mod] + h * 2 * F;
    // (unconditional loads at clamped indices: a load under a branch makes the compiler wait for it where the branch
    // ends -- in front of the ring's first DMA -- instead of at its use behind the K loop)
    const int ta = tl < F ? tl : F - 1, ts = tl < rows_valid ? tl : rows_valid - 1;
    pre_al = ga[ta]; pre_ar = ga[F + ta];
    pre_s = a.at.s[mod][m0 + ts];
    const int w = wave12 - 8;
    // waves w < ITOT % 4 carry IHI instructions per plane, the others ILO (equal when ITOT % 4 == 0)
    const GatfOperands op{a.xh[mod][g], a.x_lo_off, a.wh[mod], a.w_lo_off, a.wq[mod]};
    gatf_loader_wave<C, TERMS>(op, smem, h * F, m0, BN, D, nk, w, lane);
  } else {
    const int rg = wave12 >> 2;
    const float* bias = a.bias[mod] + h * F;
#pragma unroll
    for (int j = 0; j < NTC; ++j) bv[j] = bias[col0 + 16 * j + (lane & 15)];
    const float* wsc = C::kLo8 ? a.wscale[mod] + h * F : nullptr;
    if (rg == 0) { row0 = 0; gatf_compute<C, TERMS, C::MT0, NTC>(smem, nk, row0, col0, lane, acc, GatfNoPrefetch(), wsc); }
    else { row0 = 16 * C::MT0; gatf_compute<C, TERMS, (MTILES - C::MT0 > 0 ? MTILES - C::MT0 : 1), NTC>(smem, nk, row0, col0, lane, acc, GatfNoPrefetch(), wsc); }
  }
  __builtin_amdgcn_s_barrier();                            // every fragment read of the ring is done: its LDS is free
  GFCLK(1);

  // ---- epilogue images (floats) -------------------------------------------------------------------------------------
  // Barriers here are raw s_barrier behind an LDS-only wait (lgk_barrier): __syncthreads() also waits for every outstanding
  // global store (vmcnt(0)), which made each phase boundary wait for the 61 KB of Wh stores of phase E2 to be acknowledged
  // by L2 (phase clocks: E2 5.9 us of a 16 us epilogue).  Nothing after E1 loads from global memory, so stores simply drain
  // under the later phases.
  const int S = (N + 3) >> 2, NP = 4 * S + 1;
  float* shZ = reinterpret_cast<float*>(smem);             // [RP + 4][FS]  Wh tile (+ 4 zero rows), later the output tile
  float* shA = shZ + (RP + 4) * FS;                        // [SPT][16 NT][NP]  alpha_d[i][j] * s_j per sample
  float* shAl = shA + SPT * 16 * NT * NP;                  // [F]
  float* shAr = shAl + F;                                  // [F]
  float* shS = shAr + F;                                   // [RP]
  float* shU = shS + RP;                                   // [RP]
  float* shV = shU + RP;                                   // [RP]
  auto lgk_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // E1
  if (!loader) {
    const int mtw = (wave12 >> 2) == 0 ? C::MT0 : MTILES - C::MT0;
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      const int col = col0 + 16 * j + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < mtw) {
#pragma unroll
          for (int r = 0; r < 4; ++r) shZ[(row0 + 16 * i + 4 * (lane >> 4) + r) * FS + col] = acc[i][j][r] + bv[j];
        }
      }
    }
  } else {
    const int tl = tid - 512;                              // 0 .. 255
    for (int e = tl; e < SPT * 16 * NT * NP; e += 256) shA[e] = 0.f;
    for (int e = tl; e < 4 * FS; e += 256) shZ[RP * FS + e] = 0.f;
    static_assert(F <= 256 && RP <= 256, "one prefetched element per loader thread");
    if (tl < F) { shAl[tl] = pre_al; shAr[tl] = pre_ar; }
    if (tl < RP) shS[tl] = tl < rows_valid ? pre_s : 0.f;
  }
  lgk_barrier();
  GFCLK(2);
  const int rsub = lane >> 4, c16 = lane & 15;             // the row phases: 4 rows per wave pass, 16 lanes per row
  const float inv_n = 1.0f / (float)N;
  // E2: the projection is saved for the backward (fp32, whole row pieces), and the two halves of the logits:
  // u_i = a_l . Wh_i, v_i = a_r . Wh_i.  Lane (rsub, c16) takes the columns 4 c16 + 64 k .. + 3 (k < NTC) of row 4 pass + rsub;
  // the 16 partial sums of a row meet by DPP rotations inside the 16-lane row (every lane ends with the total).
  // The Wh tile leaves for global memory from REGISTERS, a piece per phase: 61 KB per workgroup issued in one burst took the
  // whole chip's store bandwidth for 3.5 us while nothing else could proceed (phase clocks: E2 4.3 us), whereas E3 / E4 leave
  // the memory system idle.  (E4 overwrites the image, so the values are read here.)
  constexpr int WV = (RP * F4 + 767) / 768;                // float4 pieces per thread
  float4 wv[WV];
  float* Wg = a.Wh_out[mod] + (int64_t)m0 * HF + h * F;
  auto wh_store = [&](int k) {
    const int e = tid + 768 * k;
    if (e < rows_valid * F4) {
      const int r = e / F4, q = e - r * F4;
      st4(reinterpret_cast<float4*>(Wg + (int64_t)r * HF + 4 * q), wv[k]);
    }
  };
  {
#pragma unroll
    for (int k = 0; k < WV; ++k) {
      const int e = tid + 768 * k;
      const int r = e / F4, q = e - r * F4;
      wv[k] = e < RP * F4 ? *reinterpret_cast<const float4*>(shZ + r * FS + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < WV; k += 3) wh_store(k);
    float4 al4[NTC], ar4[NTC];
#pragma unroll
    for (int k = 0; k < NTC; ++k) {
      al4[k] = *reinterpret_cast<const float4*>(shAl + 4 * c16 + 64 * k);
      ar4[k] = *reinterpret_cast<const float4*>(shAr + 4 * c16 + 64 * k);
    }
    for (int pass = wave12; 4 * pass < rows_valid; pass += 12) {
      const int r = 4 * pass + rsub;                       // < RP: the image has that many rows (rows >= rows_valid: unused values)
      float pu = 0.f, pv = 0.f;
#pragma unroll
      for (int k = 0; k < NTC; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(shZ + r * FS + 4 * c16 + 64 * k);
        pu += w.x * al4[k].x + w.y * al4[k].y + w.z * al4[k].z + w.w * al4[k].w;
        pv += w.x * ar4[k].x + w.y * ar4[k].y + w.z * ar4[k].z + w.w * ar4[k].w;
      }
      pu += dpp_move<0xb1, 0xf>(0.f, pu); pv += dpp_move<0xb1, 0xf>(0.f, pv);     // quad_perm [1,0,3,2]
      pu += dpp_move<0x4e, 0xf>(0.f, pu); pv += dpp_move<0x4e, 0xf>(0.f, pv);     // quad_perm [2,3,0,1]
      pu += dpp_move<0x124, 0xf>(0.f, pu); pv += dpp_move<0x124, 0xf>(0.f, pv);   // row_ror 4
      pu += dpp_move<0x128, 0xf>(0.f, pu); pv += dpp_move<0x128, 0xf>(0.f, pv);   // row_ror 8
      if (c16 == 0 && r < rows_valid) {
        shU[r] = pu; shV[r] = pv;
        const int sidx = (int)(((float)r + 0.5f) * inv_n), i = r - sidx * N;      // r / N, exact for these small integers
        const int64_t o = ((int64_t)(b0 + sidx) * kHeads + h) * N + i;
        a.at.u[mod][o] = pu;
        a.at.v[mod][o] = pv;
      }
    }
  }
  lgk_barrier();
  GFCLK(3);
  // E3: softmax over j of LeakyReLU(u_i + v_j + c); alpha-site dropout; the punish gate of column j.  Lane (rsub, c16) takes
  // j = c16 + 16 t (t < NT) of row 4 pass + rsub; row maximum and row sum by DPP rotations inside the 16-lane row.
  {
#pragma unroll
    for (int k = 1; k < WV; k += 3) wh_store(k);
    const float cc = a.at.gat_c[mod][h];
    const uint32_t key = a.at.key_alpha[mod][h];
    for (int pass = wave12; 4 * pass < rows_valid; pass += 12) {
      const int r = 4 * pass + rsub;
      const bool valid = r < rows_valid;
      const int sidx = valid ? (int)(((float)r + 0.5f) * inv_n) : 0, i = valid ? r - sidx * N : 0;
      const float ui = shU[valid ? r : 0];
      float e[NT];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = c16 + 16 * t;
        e[t] = -INFINITY;
        if (j < N) { const float x = ui + shV[sidx * N + j] + cc; e[t] = x > 0.f ? x : kSlope * x; }
        mx = fmaxf(mx, e[t]);
      }
      mx = fmaxf(mx, dpp_move<0xb1, 0xf>(-INFINITY, mx));
      mx = fmaxf(mx, dpp_move<0x4e, 0xf>(-INFINITY, mx));
      mx = fmaxf(mx, dpp_move<0x124, 0xf>(-INFINITY, mx));
      mx = fmaxf(mx, dpp_move<0x128, 0xf>(-INFINITY, mx));
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) { e[t] = (c16 + 16 * t) < N ? expf(e[t] - mx) : 0.f; sum += e[t]; }
      sum += dpp_move<0xb1, 0xf>(0.f, sum);
      sum += dpp_move<0x4e, 0xf>(0.f, sum);
      sum += dpp_move<0x124, 0xf>(0.f, sum);
      sum += dpp_move<0x128, 0xf>(0.f, sum);
      if (valid) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int j = c16 + 16 * t;
          if (j < N)
            shA[(sidx * 16 * NT + i) * NP + j] =
                e[t] / sum * drop_factor(key, (uint32_t)(((b0 + sidx) * N + i) * N + j)) * shS[sidx * N + j];
        }
      }
    }
  }
  lgk_barrier();
  GFCLK(4);
  // E4: P = (alpha_d * s^T) . Wh on the fp32 MFMA.  A = shA (rows i, reduction j), B = the sample's rows of shZ (reduction
  // j, columns f); reduction steps past the sample's N rows read zero (their A entries are zero too).  Items = (sample,
  // 16-column strip); a wave takes a contiguous run of them, so its A fragments change only when the sample does.  Every load
  // of an item is issued ahead of its MFMAs; the wave then overwrites its strip of shZ with the result (only the owner of a
  // strip ever reads those columns of that sample).
  {
#pragma unroll
    for (int k = 2; k < WV; k += 3) wh_store(k);
    const int r16 = lane & 15, kq = lane >> 4;
    const int items = spt_valid * JT, per = (items + 11) / 12;
    const int it0 = wave12 * per, it1 = (it0 + per) < items ? (it0 + per) : items;
    // SC = the number of reduction steps as a compile-time constant where it is one of the shipped clip counts' (a uniform
    // branch per step between dependent MFMAs cost as much as the MFMAs), 0 = the run-time bound
    auto run = [&](auto sc_tag) {
      constexpr int SC = decltype(sc_tag)::value;
      constexpr int KS = SC ? SC : 4 * NT;
      float af[NT][KS];
      int cur = -1;
      for (int item = it0; item < it1; ++item) {
        const int sidx = item / JT, J = item - sidx * JT;
        if (sidx != cur) {
          cur = sidx;
          const float* A_ = shA + sidx * 16 * NT * NP;
#pragma unroll
          for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int s4 = 0; s4 < KS; ++s4) af[I][s4] = (SC || s4 < S) ? A_[(16 * I + r16) * NP + 4 * s4 + kq] : 0.f;
        }
        float* Z_ = shZ + sidx * N * FS + 16 * J;
        float bf[KS];
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) {
          const int jr = 4 * s4 + kq;
          bf[s4] = ((SC || s4 < S) && jr < N) ? Z_[jr * FS + r16] : 0.f;
        }
        gatf_f32x4 pacc[NT];
#pragma unroll
        for (int I = 0; I < NT; ++I) pacc[I] = gatf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) {
          if (SC || s4 < S) {
#pragma unroll
            for (int I = 0; I < NT; ++I) pacc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[I][s4], bf[s4], pacc[I], 0, 0, 0);
          }
        }
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + 4 * kq + r;
            if (row < N) Z_[row * FS + r16] = pacc[I][r];
          }
      }
    };
    if (S == 4 * NT) run(std::integral_constant<int, 4 * NT>{});                      // N = 16 NT - 3 .. 16 NT (8 -> NT 1: S = 2 below)
    else if (NT == 2 && S == 5) run(std::integral_constant<int, NT == 2 ? 5 : 1>{});  // N = 17 .. 20
    else if (NT == 1 && S == 2) run(std::integral_constant<int, NT == 1 ? 2 : 1>{});  // N = 5 .. 8
    else if (NT == 2 && S == 6) run(std::integral_constant<int, NT == 2 ? 6 : 1>{});  // N = 21 .. 24
    else run(std::integral_constant<int, 0>{});
  }
  lgk_barrier();
  GFCLK(5);
  // E5: ELU, out-site dropout, fp32 + planes, 16 bytes per lane
  {
    float* out = (g ? a.at.out_s[mod] : a.at.out_c[mod]) + (int64_t)m0 * D + hl * F;
    uint16_t* zp = g ? a.at.zp_s[mod] : a.at.zp_c[mod];
    const uint32_t ko = a.at.key_out[mod][g];
    for (int e = tid; e < rows_valid * F4; e += 768) {
      const int r = e / F4, q = e - r * F4;
      const float4 v = *reinterpret_cast<const float4*>(shZ + r * FS + 4 * q);
      const int64_t o = (int64_t)r * D + 4 * q;
      const int64_t eg = (int64_t)m0 * D + hl * F + o;     // element index inside the (B, N, D) output
      float kf[4];
      drop_factor4(ko, (uint32_t)eg, kf);
      float4 y;
      y.x = elu_fast(v.x) * kf[0];
      y.y = elu_fast(v.y) * kf[1];
      y.z = elu_fast(v.z) * kf[2];
      y.w = elu_fast(v.w) * kf[3];
      st4(reinterpret_cast<float4*>(out + o), y);
      split_store4(zp, a.at.z_lo_off, eg, y);
    }
  }
  GFCLK(6);
}

#ifdef DVGR_CLK
extern "C" int dvgr_dbg_gatf_clocks(long long* host_out, int n) {
  if (n > 1024 * 8) n = 1024 * 8;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gatf_clk), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

// ---- host side ----------------------------------------------------------------------------------------------------
bool gat_fused_supported(int B, int N, int D, int prec) {
  GatfShape s;
  return prec != DVGR_F32 && D == 768 && gatf_shape(B, N, &s) && (int64_t)B * N * D < (1ll << 31);
}

template <int TERMS, int MTILES, int NT, bool LO8, int KB = 32>
static int gatf_launch(hipStream_t st, const GatFusedArgs& a) {
  constexpr int NTC = 3;
  using C = GatfCfg<TERMS, MTILES, NTC, LO8, KB>;
  const int N = a.at.N, S = (N + 3) / 4, NP = 4 * S + 1;
  const size_t epi = sizeof(float) * ((size_t)(C::RP + 4) * C::FS + (size_t)a.spt * 16 * NT * NP + 2 * C::F + 3 * C::RP);
  const size_t lds = (epi > (size_t)C::kRingBytes ? epi : (size_t)C::kRingBytes) + 16;
  if (lds > 160 * 1024) return DVGR_E_DIMS;
  // the raised dynamic-LDS limit belongs to the (function, device) pair: one flag per instantiation and device (benign race:
  // the attribute is idempotent)
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || !attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gat_fused_fwd<TERMS, MTILES, NT, NTC, LO8, KB>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) attr_set[dev] = true;
  }
  const int mtiles = (a.at.B + a.spt - 1) / a.spt;
  const int total = mtiles * kHeads * 2;
  DVGR_LAUNCH_FORKABLE((k_gat_fused_fwd<TERMS, MTILES, NT, NTC, LO8, KB>), dim3(8 * ((total + 7) / 8)), dim3(768), lds, st, a);
  DVGR_LAUNCH_CHECK();
}

template <int TERMS, bool LO8>
static int gatf_dispatch(hipStream_t st, const GatFusedArgs& a, const GatfShape& s) {
  if (s.mtiles16 == 4 && s.nt == 1) return gatf_launch<TERMS, 4, 1, LO8>(st, a);
  if (s.mtiles16 == 5 && s.nt == 2) return gatf_launch<TERMS, 5, 2, LO8>(st, a);
  if (s.mtiles16 == 6 && s.nt == 2) return gatf_launch<TERMS, 6, 2, LO8>(st, a);
  if (s.mtiles16 == 6 && s.nt == 3) return gatf_launch<TERMS, 6, 3, LO8>(st, a);
  if (s.mtiles16 == 8 && s.nt == 4) return gatf_launch<TERMS, 8, 4, LO8>(st, a);
  return DVGR_E_DIMS;
}

int gat_fused_fwd(hipStream_t st, const GatFusedArgs& a_in, int prec) {
  GatFusedArgs a = a_in;
  GatfShape s;
  if (!gat_fused_supported(a.at.B, a.at.N, a.at.D, prec) || !gatf_shape(a.at.B, a.at.N, &s)) return DVGR_E_DIMS;
  for (int m = 0; m < 2; ++m) {
    if (!a.wh[m] || !a.xh[m][0] || !a.xh[m][1] || !a.bias[m] || !a.Wh_out[m]) return DVGR_E_NULL;
    if (((uintptr_t)a.wh[m] | (uintptr_t)a.xh[m][0] | (uintptr_t)a.xh[m][1]) & 15) return DVGR_E_ALIGN;
  }
  if ((a.x_lo_off & 7) || (a.w_lo_off & 7)) return DVGR_E_ALIGN;
  a.spt = s.spt;
  if (a.kb64 && prec == DVGR_BF16X3 && (a.at.D % 64) == 0) {      // 64-deep K-tiles (whole 128-byte lines per row piece), bf16 lo planes
    GatfShape hs = s;
    if (!a.no_half && gatf_half_shape(a.at.B, a.at.N, &hs)) {     // few samples of <= 16 clips: 32-row tiles fill the chip (dvgr_gatf.h)
      a.spt = hs.spt;
      return gatf_launch<3, 2, 1, false, 64>(st, a);
    }
    if (s.mtiles16 == 4 && s.nt == 1) return gatf_launch<3, 4, 1, false, 64>(st, a);
    if (s.mtiles16 == 5 && s.nt == 2) return gatf_launch<3, 5, 2, false, 64>(st, a);
  }
  const bool lo8 = prec == DVGR_BF16X3 && a.wq[0] != nullptr;
  if (lo8) {
    for (int m = 0; m < 2; ++m)
      if (!a.wq[m] || !a.wscale[m] || ((uintptr_t)a.wq[m] & 15)) return DVGR_E_NULL;
    return gatf_dispatch<3, true>(st, a, s);
  }
  return prec == DVGR_BF16X3 ? gatf_dispatch<3, false>(st, a, s) : gatf_dispatch<1, false>(st, a, s);
}

}  // namespace dvgr
