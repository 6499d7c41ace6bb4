// dvgr_gatb.hip -- the fused GAT-view BACKWARD kernel: the S4 backward contraction dz_v += dpre_v . P1 (gradient of the
// common / specific GAT outputs through AttentionSFGCN's project[0], model/Attention.py:15-17,20-23) with the attention
// backward of model/GraphNN.py:95-111 as its epilogue, in ONE launch per layer.
//
// Why this seam and not the other one: the attention backward's OUTPUT dWh feeds dX = dWh . Wcat, a reduction over all 8 heads,
// so making it the producer stage of the dX contraction needs either a 6-fold recomputation or 63 MB of float atomics (round 4,
// DESIGN.md section 3.2).  Its INPUT side is tile-local: head h of GAT g of sample b needs exactly the N rows of sample b x the
// F columns hl F .. (hl + 1) F of dz_g (h = 4 g + hl) -- which is what ONE workgroup of the dz contraction produces when its
// output tile is SPT whole samples x one head's columns, the tile of the forward kernel (dvgr_gatf.hip).  As two launches
// (gemm_planes_kernel<TAG_BWD_SF_DX> then k_gat_attn_bwd) dz (2 x B N x D fp32, 15.7 MB at C2) is read-modify-written by the
// contraction's generic epilogue and read again by 1024 attention workgroups; phase clocks of the attention kernel alone:
// 19.7 us per workgroup, 6.9 of it its load phase (profiles/r05_attn_bwd_phase_clocks.txt).
//
// Work decomposition (C2: B = 64, N = 20, F = 192): as dvgr_gatf.hip -- 16 row tiles x (2 views x 4 heads) x 2 modalities = 256
// workgroups, 12 waves: 4 loaders (LDS-DMA ring), 8 computing (bf16x3 MFMA).  K loop: A = 16 MTILES rows of view g of dpre
// (planes written by k_sf_bwd), B = rows hl F .. of P1^T (planes transposed once per backward call by k_transpose_planes: with
// P1 as [k][n] the B tile would need the IC fragment path at a 192-column tile width), reduction over D.
// Epilogue, all 12 waves, in the ring's (now idle) LDS -- SPT samples at once:
//   (ahead of the K loop: the computing waves fetch the saved projection Wh, the loader threads the saved GAT output -- two of
//       the three fp32 tiles the attention needs; they land under the K loop)
//   B0  the loader threads issue their loads of dz's part written by k_sf_bwd (beta_g do + the list gradient); accumulators -> shP,
//       Wh -> shW; the loaders fill s, u, v, a_l, a_r and clear the padded images
//   B1  computing waves: alpha_ij = softmax_j LeakyReLU(u_i + v_j + c) recomputed (4 rows per wave pass, DPP row reductions),
//       alpha and alpha * keep; beside them the loader threads: dP = (acc + dz) * keep_out * ELU'(out) -> shP
//   B2  dalpha_ij = keep_ij s_j (dP_i . Wh_j) on the fp32 MFMA, one (sample, 16 x 16 tile) per wave at a time
//   B3  softmax and LeakyReLU backward, du_i (4 rows per wave pass)
//   B4  dv_j, dc (a thread per column / sample) beside dZ = alpha_d^T . dP on the fp32 MFMA, one (sample, 16-column strip)
//       per wave at a time; the wave overwrites its strip of shP
//   B5  dWh_j = s_j dZ_j + du_j a_l + dv_j a_r -> bf16 planes (+ fp32 where it has a reader), ds_j = dZ_j . Wh_j;
//       d a_l = du^T . Wh, d a_r = dv^T . Wh per sample (a thread per column)
// Outputs are those of k_gat_attn_bwd (dWh planes, ds / da / dc partials per (sample, head)); dz itself is not written: nobody
// else reads it at graph_layers = 1.
#include <type_traits>

#include "dvgr_gatf.h"

#define DVGR_LAUNCH_CHECK() return (int)hipGetLastError()

namespace dvgr {

typedef float gatb_f32x4 __attribute__((ext_vector_type(4)));

// Phase clocks (debug builds only, `make clk`; tools/gatb_clocks.py): thread 0 of every workgroup stamps the 100 MHz wall clock
#ifdef DVGR_CLK
__device__ long long g_gatb_clk[1024 * 8];
#define GBCLK(k)                                                                                              \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_gatb_clk[blockIdx.x * 8 + (k)] = wall_clock64();            \
  } while (0)
#else
#define GBCLK(k) do { } while (0)
#endif

// ---- P1 -> P1^T, bf16 hi / lo planes, 64 x 64 tiles through LDS -----------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_planes(const TransposeBatch b) {
  __shared__ uint16_t t[2][64][72];                       // (row stride 144 B: 16-byte aligned vectors, 36 banks apart)
  const int D = b.D, tid = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const uint16_t* src = b.j[blockIdx.z].src;
  uint16_t* dst = b.j[blockIdx.z].dst;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int v = tid + 256 * it, row = v >> 3, c8 = (v & 7) * 8;      // 512 vectors of 8 elements per plane
    const uint4 hi = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + row) * D + c0 + c8);
    const uint4 lo = *reinterpret_cast<const uint4*>(src + b.src_lo + (int64_t)(r0 + row) * D + c0 + c8);
    *reinterpret_cast<uint4*>(&t[0][row][c8]) = hi;
    *reinterpret_cast<uint4*>(&t[1][row][c8]) = lo;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int v = tid + 256 * it, orow = v >> 3, o8 = (v & 7) * 8;     // output row = source column
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        w[k] = (uint32_t)t[p][o8 + 2 * k][orow] | ((uint32_t)t[p][o8 + 2 * k + 1][orow] << 16);
      *reinterpret_cast<uint4*>(dst + (p ? b.dst_lo : 0) + (int64_t)(c0 + orow) * D + r0 + o8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

int transpose_planes(hipStream_t st, const TransposeBatch& b) {
  if (b.n < 1 || b.n > kMaxTransposeJobs || b.D < 64 || (b.D % 64) != 0) return DVGR_E_DIMS;
  DVGR_LAUNCH(k_transpose_planes, dim3(b.D / 64, b.D / 64, b.n), dim3(256), 0, st, b);
  DVGR_LAUNCH_CHECK();
}

// ---- bf16 lo plane -> 8-bit block-scaled lo plane (one workgroup per row) ------------------------------------------------------
__global__ __launch_bounds__(256) void k_quant_lo8(const Lo8Batch b) {
  __shared__ float red[4];
  const Lo8Job& J = b.j[blockIdx.y];
  const int row = blockIdx.x, D = b.D, tid = threadIdx.x;
  if (row >= J.rows) return;                                // (uniform per workgroup)
  const uint16_t* src = J.lo + (int64_t)row * D;
  float amax = 0.f;
  for (int k = tid; k < D; k += 256) amax = fmaxf(amax, fabsf(__uint_as_float((uint32_t)src[k] << 16)));
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = amax * (1.0f / 127.0f), inv = amax > 0.f ? 127.0f / amax : 0.f;
  if (tid == 0) J.scale[row] = s;
  uint8_t* q = J.q + (int64_t)row * D;
  for (int k = tid; k < D; k += 256) {
    const float x = __uint_as_float((uint32_t)src[k] << 16);
    int v = (int)rintf(x * inv) + 128;
    v = v < 1 ? 1 : v > 255 ? 255 : v;
    q[k] = (uint8_t)v;
  }
}

int quantize_lo8(hipStream_t st, const Lo8Batch& b) {
  if (b.n < 1 || b.n > kMaxLo8Jobs || b.D < 1) return DVGR_E_DIMS;
  int mx = 0;
  for (int i = 0; i < b.n; ++i) mx = b.j[i].rows > mx ? b.j[i].rows : mx;
  DVGR_LAUNCH(k_quant_lo8, dim3(mx, b.n), dim3(256), 0, st, b);
  DVGR_LAUNCH_CHECK();
}

// ---- the fused kernel ------------------------------------------------------------------------------------------------
// 16-lane row reductions by DPP rotations (every lane of the row ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xb1, 0xf>(0.f, v);      // quad_perm [1,0,3,2]
  v += dpp_move<0x4e, 0xf>(0.f, v);      // quad_perm [2,3,0,1]
  v += dpp_move<0x124, 0xf>(0.f, v);     // row_ror 4
  v += dpp_move<0x128, 0xf>(0.f, v);     // row_ror 8
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_move<0xb1, 0xf>(-INFINITY, v));
  v = fmaxf(v, dpp_move<0x4e, 0xf>(-INFINITY, v));
  v = fmaxf(v, dpp_move<0x124, 0xf>(-INFINITY, v));
  v = fmaxf(v, dpp_move<0x128, 0xf>(-INFINITY, v));
  return v;
}

template <int MTILES, int NTC>
struct GatbEpi {                                          // epilogue images, in floats
  static constexpr int RP = 16 * MTILES, F = 64 * NTC, FS = F + 4;
  __host__ __device__ static constexpr size_t floats(int N, int SPT) {
    return 2 * (size_t)(RP + 1) * FS + 2 * F + (size_t)SPT * (3 * N + 1) * (N + 1) + 6 * RP;
  }
};

template <int TERMS, int MTILES, int NT, int NTC, bool FROM_DX, bool LO8, int KB>
__global__ __launch_bounds__(768) void k_gat_fused_bwd(const GatFusedBwdArgs a) {
  using C = GatfCfg<TERMS, MTILES, NTC, LO8, KB>;
  extern __shared__ __attribute__((aligned(16))) uint16_t gatb_smem[];
  uint16_t* smem = gatb_smem;
  const int N = a.at.N, D = a.at.D, B = a.at.B, BN = B * N;
  constexpr int F = C::F, RP = C::RP, F4 = F / 4, JT = F / 16, FS = GatbEpi<MTILES, NTC>::FS;
  const int HF = kHeads * F;
  const int SPT = a.spt;                                   // samples per tile (SPT * N <= RP)
  const int mtiles = (B + SPT - 1) / SPT;
  const int total = mtiles * kHeads * 2;
  const int chunk = (int)gridDim.x >> 3;
  const int idx = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);     // XCD x takes chunk x of the tile list
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
  const int g = h >> 2, hl = h & 3;                        // view (GAT) and the head's column block inside its (B N, D) output
  const int b0 = mt * SPT;
  const int m0 = b0 * N;
  const int spt_valid = (B - b0) < SPT ? (B - b0) : SPT;
  const int rows_valid = spt_valid * N;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave12 >= 8;
  const int nk = D / C::KB;

  f32x4 acc[4][NTC];
  int row0 = 0;
  const int col0 = (wave12 & 3) * 16 * NTC;
  float pre_al = 0.f, pre_ar = 0.f, pre_s = 0.f, pre_u = 0.f, pre_v = 0.f, pre_b = 0.f;
  constexpr bool from_dx = FROM_DX;                        // dz's pre-part is formed here (GatFusedBwdArgs::dx), not read
  // The three fp32 tiles the attention needs (47 MB over the chip at C2) issued by every workgroup at once behind the K loop
  // took 6.7 us (phase clocks, first version: the HBM rate); issued ahead of the ring's first DMA they delayed the K loop by 4.7.
  // So they are fetched UNDER the K loop, whose operands come from L2 / the Infinity Cache: the saved projection Wh by the
  // computing waves behind the first K-tile's barrier (WVC float4 pieces per thread: nothing in the K loop waits for vector
  // memory) together with YC pieces per thread of the saved GAT output, the other YL pieces per thread of that tile by the loader
  // threads behind the ring's prologue (a loader's counted wait covers its own older loads: 15 pieces there blocked the ring at
  // its second K-tile, and made the loader threads' dP pass the longest thing in B1; the counted waits of the first
  // K-tiles leave them in flight).  Only the part of dz k_sf_bwd wrote follows behind the K loop, by the loader threads, under
  // the computing waves' accumulator pass and the recomputation of alpha.
  constexpr int WVC = (RP * F4 + 511) / 512;
  // (the form that reads dz's pre-part keeps the whole tile with the loader threads: its computing threads would carry dz pieces too)
#ifndef DVGR_GATB_YC_NUM
#define DVGR_GATB_YC_NUM 2
#endif
  constexpr int YC = FROM_DX ? (RP * F4 * DVGR_GATB_YC_NUM / 5 + 511) / 512 : 0;   // pieces of the GAT-output tile per computing thread: elements [0, 512 YC) ...
  constexpr int EC = 512 * YC;
  constexpr int YL = (RP * F4 - EC + 255) / 256;           // ... and per loader thread: elements [512 YC, RP F4)
  static_assert(EC < RP * F4, "the loader threads' share of the tile");
  gatb_f32x4 whv[WVC], yv[YL], zv[YL], yc[YC > 0 ? YC : 1];
  const int emax = rows_valid * F4 - 1;
  GBCLK(0);
  if (loader) {
    const int tl = tid - 512;                              // 0 .. 255
    const float* ga = a.at.gat_a[mod] + h * 2 * F;
    // (unconditional loads at clamped indices, issued ahead of the ring's first DMA: they land under the K loop)
    const int ta = tl < F ? tl : F - 1, ts = tl < rows_valid ? tl : rows_valid - 1;
    const int sx = ts / N, ix = ts - sx * N;
    const int64_t uo = ((int64_t)(b0 + sx) * kHeads + h) * N + ix;
    pre_al = ga[ta]; pre_ar = ga[F + ta];
    pre_s = a.at.s[mod][m0 + ts];
    pre_u = a.at.u[mod][uo]; pre_v = a.at.v[mod][uo];
    if (from_dx) pre_b = a.beta[mod][(int64_t)(m0 + ts) * 2 + g];
    const float* yg = (g ? a.at.out_s[mod] : a.at.out_c[mod]) + (int64_t)m0 * D + hl * F;
    auto fetch_out = [&]() {                               // behind the ring's prologue (gatf_loader: NPF)
#pragma unroll
      for (int k = 0; k < YL; ++k) {
        int e = EC + tl + 256 * k;
        e = e < emax ? e : emax;
        const int r = e / F4, q = e - r * F4;
        yv[k] = *reinterpret_cast<const gatb_f32x4*>(yg + (int64_t)r * D + 4 * q);
      }
    };
    const int w = wave12 - 8;
    const uint16_t* xa = a.ah[mod] + (int64_t)g * BN * D;  // view g of dpre (2, B N, D)
    const GatfOperands op{xa, a.a_lo_off, a.bh[mod], a.b_lo_off, a.bq[mod]};
#if defined(DVGR_CLK) && defined(DVGR_GATB_DBG_NOFETCH_OUT)  // (phase-clock experiments: the K loop without the loaders' tile fetch)
    gatf_loader_wave<C, TERMS>(op, smem, hl * F, m0, BN, D, nk, w, lane);
#else
    gatf_loader_wave<C, TERMS, YL>(op, smem, hl * F, m0, BN, D, nk, w, lane, fetch_out);
#endif
  } else {
    const float* wg = a.at.Wh[mod] + (int64_t)m0 * HF + h * F;
    const float* yg = (g ? a.at.out_s[mod] : a.at.out_c[mod]) + (int64_t)m0 * D + hl * F;
    auto fetch_wh = [&]() {                                // behind the first K-tile's barrier (gatf_compute: first)
      if constexpr (YC > 0) {
#pragma unroll
        for (int k = 0; k < YC; ++k) {
          int e = tid + 512 * k;
          e = e < emax ? e : emax;
          const int r = e / F4, q = e - r * F4;
          yc[k] = *reinterpret_cast<const gatb_f32x4*>(yg + (int64_t)r * D + 4 * q);
        }
      }
#pragma unroll
      for (int k = 0; k < WVC; ++k) {
        int e = tid + 512 * k;
        e = e < emax ? e : emax;
        const int r = e / F4, q = e - r * F4;
#if defined(DVGR_CLK) && defined(DVGR_GATB_DBG_NOFETCH_WH)   // (phase-clock experiments: the K loop without the computing waves' tile fetch)
        whv[k] = gatb_f32x4{(float)r, (float)q, 0.f, 0.f};
#else
        whv[k] = *reinterpret_cast<const gatb_f32x4*>(wg + (int64_t)r * HF + 4 * q);
#endif
      }
    };
    const int rg = wave12 >> 2;
    const float* wsc = C::kLo8 ? a.bscale[mod] + hl * F : nullptr;
    if (rg == 0) { row0 = 0; gatf_compute<C, TERMS, C::MT0, NTC>(smem, nk, row0, col0, lane, acc, fetch_wh, wsc); }
    else { row0 = 16 * C::MT0; gatf_compute<C, TERMS, (MTILES - C::MT0 > 0 ? MTILES - C::MT0 : 1), NTC>(smem, nk, row0, col0, lane, acc, fetch_wh, wsc); }
  }
  __builtin_amdgcn_s_barrier();                            // every fragment read of the ring is done: its LDS is free
  GBCLK(1);

  // ---- epilogue images (floats) -------------------------------------------------------------------------------------
  const int S = (N + 3) >> 2, NP = N + 1;
  float* shP = reinterpret_cast<float*>(smem);             // [RP + 1][FS]  acc, then dP, then dZ; row RP = 0
  float* shW = shP + (RP + 1) * FS;                        // [RP + 1][FS]  Wh (unscaled); row RP = 0
  float* shAl = shW + (RP + 1) * FS;                       // [F]
  float* shAr = shAl + F;                                  // [F]
  float* shA = shAr + F;                                   // [SPT][N][NP]      alpha
  float* shAd = shA + SPT * N * NP;                        // [SPT][N + 1][NP]  alpha * keep; row N and column N = 0
  float* shG = shAd + SPT * (N + 1) * NP;                  // [SPT][N][NP]      dalpha, then dpre
  float* shS = shG + SPT * N * NP;                         // [RP]
  float* shU = shS + RP;                                   // [RP]
  float* shV = shU + RP;                                   // [RP]
  float* shDu = shV + RP;                                  // [RP]
  float* shDv = shDu + RP;                                 // [RP]
  float* shB = shDv + RP;                                  // [RP]  beta_g of the row (from_dx)
  auto lgk_barrier = [&]() {                               // (raw barrier behind an LDS-only wait: global stores / loads stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  const int rsub = lane >> 4, c16 = lane & 15;             // the row phases: 4 rows per wave pass, 16 lanes per row
  const int r16 = lane & 15, kq = lane >> 4;               // the MFMA phases
  const float inv_n = 1.0f / (float)N;
  const uint32_t ko = a.at.key_out[mod][g];
  const uint32_t ka = a.at.key_alpha[mod][h];
  const float cc = a.at.gat_c[mod][h];

  // B0: dz's pre-part.  Either the loader threads load what k_sf_bwd wrote (consumed in B1b), or (from_dx) the computing
  // threads load the ingredients -- the running gradient dx (+ the direct gradient of the fused embedding) and the view's list
  // gradient -- and add beta_g (dx + g_embed) + g_z onto the accumulator image in B0b.  Accumulators and the prefetched Wh -> LDS.
  constexpr int WVD = WVC;
  gatb_f32x4 dxv[WVD], gzv[WVD];
  if (loader && !from_dx) {
    const int tl = tid - 512;
    const float* zg = a.at.dz[mod] + (int64_t)g * BN * D + (int64_t)m0 * D + hl * F;
#pragma unroll
    for (int k = 0; k < YL; ++k) {
      int e = EC + tl + 256 * k;
      e = e < emax ? e : emax;
      const int r = e / F4, q = e - r * F4;
      zv[k] = *reinterpret_cast<const gatb_f32x4*>(zg + (int64_t)r * D + 4 * q);
    }
  }
  if (!loader) {
    const int mtw = (wave12 >> 2) == 0 ? C::MT0 : MTILES - C::MT0;
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      const int col = col0 + 16 * j + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < mtw) {
#pragma unroll
          for (int r = 0; r < 4; ++r) shP[(row0 + 16 * i + 4 * (lane >> 4) + r) * FS + col] = acc[i][j][r];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < WVC; ++k) {
      const int e = tid + 512 * k;
      if (e <= emax) {
        const int r = e / F4, q = e - r * F4;
        *reinterpret_cast<gatb_f32x4*>(shW + r * FS + 4 * q) = whv[k];
      }
    }
    if (from_dx) {                                         // (behind the two passes above: their registers are free now)
      const float* xg = a.dx[mod] + (int64_t)m0 * D + hl * F;
      const float* eg = a.g_embed[mod] ? a.g_embed[mod] + (int64_t)m0 * D + hl * F : nullptr;
      const float* gg = a.g_z[mod][g] ? a.g_z[mod][g] + (int64_t)m0 * D + hl * F : nullptr;
#pragma unroll
      for (int k = 0; k < WVD; ++k) {
        int e = tid + 512 * k;
        e = e < emax ? e : emax;
        const int r = e / F4, q = e - r * F4;
        dxv[k] = *reinterpret_cast<const gatb_f32x4*>(xg + (int64_t)r * D + 4 * q);
        gzv[k] = gg ? *reinterpret_cast<const gatb_f32x4*>(gg + (int64_t)r * D + 4 * q) : gatb_f32x4{0.f, 0.f, 0.f, 0.f};
        if (eg) dxv[k] += *reinterpret_cast<const gatb_f32x4*>(eg + (int64_t)r * D + 4 * q);
    }
    }
  } else {
    const int tl = tid - 512;                              // 0 .. 255
    for (int e = tl; e < SPT * (N + 1) * NP; e += 256) shAd[e] = 0.f;
    for (int e = tl; e < FS; e += 256) { shP[RP * FS + e] = 0.f; shW[RP * FS + e] = 0.f; }
    static_assert(F <= 256 && RP <= 256, "one prefetched element per loader thread");
    if (tl < F) { shAl[tl] = pre_al; shAr[tl] = pre_ar; }
    if (tl < RP) {
      const bool ok = tl < rows_valid;
      shS[tl] = ok ? pre_s : 0.f; shU[tl] = ok ? pre_u : 0.f; shV[tl] = ok ? pre_v : 0.f;
      shDu[tl] = 0.f; shDv[tl] = 0.f; shB[tl] = ok ? pre_b : 0.f;
    }
  }
  lgk_barrier();
  // alpha is recomputed by whoever is free: the passes [0, PX) by the loader waves while the computing threads add dz's
  // pre-part (from_dx), the rest by the computing waves while the loader threads form dP
  const int passes = (rows_valid + 3) >> 2;
  const int PX = from_dx ? (passes < 8 ? passes : 8) : 0;
  auto alpha_pass = [&](int pass) {
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
    mx = row16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) { e[t] = (c16 + 16 * t) < N ? expf(e[t] - mx) : 0.f; sum += e[t]; }
    sum = row16_sum(sum);
    if (valid) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = c16 + 16 * t;
        if (j < N) {
          const float al = e[t] / sum;
          shA[(sidx * N + i) * NP + j] = al;
          shAd[(sidx * (N + 1) + i) * NP + j] = al * drop_factor(ka, (uint32_t)(((b0 + sidx) * N + i) * N + j));
        }
      }
    }
  };
  if (from_dx) {
    // B0b: the computing threads add beta_g (dx + g_embed) + g_z onto their elements of the accumulator image
    if (!loader) {
#pragma unroll
      for (int k = 0; k < WVD; ++k) {
        const int e = tid + 512 * k;
        if (e <= emax) {
          const int r = e / F4, q = e - r * F4;
          const float bg = shB[r];
          gatb_f32x4* pp = reinterpret_cast<gatb_f32x4*>(shP + r * FS + 4 * q);
          *pp = *pp + bg * dxv[k] + gzv[k];
        }
      }
    } else {
      for (int pass = wave12 - 8; pass < PX; pass += 4) alpha_pass(pass);
    }
    lgk_barrier();
  }
  GBCLK(2);
  // B1a (computing waves): alpha = softmax_j LeakyReLU(u_i + v_j + c) recomputed, and alpha * keep.  Lane (rsub, c16) takes
  // j = c16 + 16 t of row 4 pass + rsub.
  if (!loader)
    for (int pass = PX + wave12; pass < passes; pass += 8) alpha_pass(pass);
  // B1b (the loader threads beside B1a, the computing threads behind it): dP = (acc + dz) * keep_out * ELU'(out) -> shP for the
  // elements whose `out` they hold
  {
    if (loader) {
      const int tl = tid - 512;
      const float sc = ko ? (1.0f - kDropP) : 1.0f;        // un-dropped ELU output where kept
#pragma unroll
      for (int k = 0; k < YL; ++k) {
        const int e = EC + tl + 256 * k;
        if (e <= emax) {
          const int r = e / F4, q = e - r * F4;
          float* pp = shP + r * FS + 4 * q;
          const float4 ac = *reinterpret_cast<const float4*>(pp);
          float kf[4];
          drop_factor4(ko, (uint32_t)((int64_t)(m0 + r) * D + hl * F + 4 * q), kf);
          float4 p;
          const gatb_f32x4 zk = from_dx ? gatb_f32x4{0.f, 0.f, 0.f, 0.f} : zv[k];     // (from_dx: already inside the image)
          p.x = (ac.x + zk[0]) * kf[0] * elu_grad_from_out(yv[k][0] * sc);
          p.y = (ac.y + zk[1]) * kf[1] * elu_grad_from_out(yv[k][1] * sc);
          p.z = (ac.z + zk[2]) * kf[2] * elu_grad_from_out(yv[k][2] * sc);
          p.w = (ac.w + zk[3]) * kf[3] * elu_grad_from_out(yv[k][3] * sc);
          *reinterpret_cast<float4*>(pp) = p;
        }
      }
    } else if constexpr (YC > 0) {                         // (from_dx only: dz's pre-part is already inside the image)
      const float sc = ko ? (1.0f - kDropP) : 1.0f;
#pragma unroll
      for (int k = 0; k < YC; ++k) {
        const int e = tid + 512 * k;
        if (e <= emax) {
          const int r = e / F4, q = e - r * F4;
          float* pp = shP + r * FS + 4 * q;
          const float4 ac = *reinterpret_cast<const float4*>(pp);
          float kf[4];
          drop_factor4(ko, (uint32_t)((int64_t)(m0 + r) * D + hl * F + 4 * q), kf);
          float4 p;
          p.x = ac.x * kf[0] * elu_grad_from_out(yc[k][0] * sc);
          p.y = ac.y * kf[1] * elu_grad_from_out(yc[k][1] * sc);
          p.z = ac.z * kf[2] * elu_grad_from_out(yc[k][2] * sc);
          p.w = ac.w * kf[3] * elu_grad_from_out(yc[k][3] * sc);
          *reinterpret_cast<float4*>(pp) = p;
        }
      }
    }
  }
  lgk_barrier();
  GBCLK(3);
  // B2: dalpha_ij = keep_ij * s_j * (dP_i . Wh_j): items = (sample, I, J) 16 x 16 tiles, dealt round-robin to the waves.
  // Fragment rows beyond the sample's N are redirected to the zero row RP.
  {
    const int nitems = spt_valid * NT * NT;
    for (int item = wave12; item < nitems; item += 12) {
      const int sidx = item / (NT * NT), t2 = item - sidx * NT * NT, I = t2 / NT, J = t2 - I * NT;
      const int ra = 16 * I + r16, rb = 16 * J + r16;
      const float* pa = shP + (ra < N ? sidx * N + ra : RP) * FS + 4 * kq;
      const float* pb = shW + (rb < N ? sidx * N + rb : RP) * FS + 4 * kq;
      gatb_f32x4 dacc = gatb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int t = 0; t < F / 16; ++t) {
        const float4 x = *reinterpret_cast<const float4*>(pa + 16 * t), y = *reinterpret_cast<const float4*>(pb + 16 * t);
        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, dacc, 0, 0, 0);
      }
      const int j = 16 * J + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * kq + r;
        if (i < N && j < N)
          shG[(sidx * N + i) * NP + j] = dacc[r] * shS[sidx * N + j] * drop_factor(ka, (uint32_t)(((b0 + sidx) * N + i) * N + j));
      }
    }
  }
  lgk_barrier();
  GBCLK(4);
  // B3: softmax and LeakyReLU backward, du_i
  for (int pass = wave12; 4 * pass < rows_valid; pass += 12) {
    const int r = 4 * pass + rsub;
    const bool valid = r < rows_valid;
    const int sidx = valid ? (int)(((float)r + 0.5f) * inv_n) : 0, i = valid ? r - sidx * N : 0;
    const float ui = shU[valid ? r : 0];
    float al[NT], da[NT];
    float rowdot = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = c16 + 16 * t;
      const bool in = valid && j < N;
      al[t] = in ? shA[(sidx * N + i) * NP + j] : 0.f;
      da[t] = in ? shG[(sidx * N + i) * NP + j] : 0.f;
      rowdot += al[t] * da[t];
    }
    rowdot = row16_sum(rowdot);
    float du = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = c16 + 16 * t;
      if (valid && j < N) {
        const float pre = ui + shV[sidx * N + j] + cc;
        const float dpre = al[t] * (da[t] - rowdot) * (pre > 0.f ? 1.f : kSlope);
        shG[(sidx * N + i) * NP + j] = dpre;
        du += dpre;
      }
    }
    du = row16_sum(du);
    if (valid && c16 == 0) shDu[r] = du;
  }
  lgk_barrier();
  GBCLK(5);
  // B4a: dv_j (a thread per column of a sample), dc (a thread per sample)
  if (tid < rows_valid) {
    const int sidx = (int)(((float)tid + 0.5f) * inv_n), j = tid - sidx * N;
    float dv = 0.f;
    for (int i = 0; i < N; ++i) dv += shG[(sidx * N + i) * NP + j];
    shDv[tid] = dv;
  } else if (tid >= 512 && tid < 512 + spt_valid) {
    const int sidx = tid - 512;
    float dc = 0.f;
    for (int i = 0; i < N; ++i) dc += shDu[sidx * N + i];
    a.at.dc_part[mod][(b0 + sidx) * kHeads + h] = dc;
  }
  // B4b: dZ = alpha_d^T . dP on the fp32 MFMA: items = (sample, 16-column strip); a wave takes a contiguous run of them, so
  // its A fragments change only when the sample does; the wave overwrites its strip of shP (only the strip's owner reads those
  // columns of that sample, and every load of an item is issued ahead of its stores)
  const int items = spt_valid * JT, per = (items + 11) / 12;
  const int it0 = wave12 * per, it1 = (it0 + per) < items ? (it0 + per) : items;
  {
    auto run = [&](auto sc_tag) {
      constexpr int SC = decltype(sc_tag)::value;
      constexpr int KS = SC ? SC : 4 * NT;
      float af[NT][KS];
      int cur = -1;
      for (int item = it0; item < it1; ++item) {
        const int sidx = item / JT, Ft = item - sidx * JT;
        if (sidx != cur) {
          cur = sidx;
          const float* A_ = shAd + sidx * (N + 1) * NP;
#pragma unroll
          for (int Jt = 0; Jt < NT; ++Jt)
#pragma unroll
            for (int s4 = 0; s4 < KS; ++s4) {
              const int i = 4 * s4 + kq, j = 16 * Jt + r16;
              af[Jt][s4] = (SC || s4 < S) ? A_[(i < N ? i : N) * NP + (j < N ? j : N)] : 0.f;
            }
        }
        float* P_ = shP + sidx * N * FS + 16 * Ft;
        float bf[KS];
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) {
          const int i = 4 * s4 + kq;
          bf[s4] = ((SC || s4 < S) && i < N) ? P_[i * FS + r16] : 0.f;
        }
        gatb_f32x4 zacc[NT];
#pragma unroll
        for (int Jt = 0; Jt < NT; ++Jt) zacc[Jt] = gatb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) {
          if (SC || s4 < S) {
#pragma unroll
            for (int Jt = 0; Jt < NT; ++Jt) zacc[Jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[Jt][s4], bf[s4], zacc[Jt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int Jt = 0; Jt < NT; ++Jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * Jt + 4 * kq + r;
            if (j < N) P_[j * FS + r16] = zacc[Jt][r];
          }
      }
    };
    if (S == 4 * NT) run(std::integral_constant<int, 4 * NT>{});
    else if (NT == 2 && S == 5) run(std::integral_constant<int, NT == 2 ? 5 : 1>{});
    else if (NT == 1 && S == 2) run(std::integral_constant<int, NT == 1 ? 2 : 1>{});
    else run(std::integral_constant<int, 0>{});
  }
  lgk_barrier();
  GBCLK(6);
  // B5a: dWh_j = s_j dZ_j + du_j a_l + dv_j a_r (planes; fp32 where it has a reader), ds_j = dZ_j . Wh_j.  Lane (rsub, c16)
  // takes the columns 4 c16 + 64 k .. + 3 (k < NTC) of row 4 pass + rsub.
  {
    float4 al4[NTC], ar4[NTC];
#pragma unroll
    for (int k = 0; k < NTC; ++k) {
      al4[k] = *reinterpret_cast<const float4*>(shAl + 4 * c16 + 64 * k);
      ar4[k] = *reinterpret_cast<const float4*>(shAr + 4 * c16 + 64 * k);
    }
    float* dWh = a.at.dWh[mod] ? a.at.dWh[mod] + (int64_t)m0 * HF + h * F : nullptr;
    for (int pass = wave12; 4 * pass < rows_valid; pass += 12) {
      const int r = 4 * pass + rsub;
      const bool valid = r < rows_valid;
      const int rr = valid ? r : 0;
      const float sj = shS[rr], duj = shDu[rr], dvj = shDv[rr];
      float dsp = 0.f;
#pragma unroll
      for (int k = 0; k < NTC; ++k) {
        const float4 z = *reinterpret_cast<const float4*>(shP + rr * FS + 4 * c16 + 64 * k);
        const float4 w = *reinterpret_cast<const float4*>(shW + rr * FS + 4 * c16 + 64 * k);
        const float4 o = make_float4(sj * z.x + duj * al4[k].x + dvj * ar4[k].x, sj * z.y + duj * al4[k].y + dvj * ar4[k].y,
                                     sj * z.z + duj * al4[k].z + dvj * ar4[k].z, sj * z.w + duj * al4[k].w + dvj * ar4[k].w);
        if (valid) {
          if (dWh) st4(reinterpret_cast<float4*>(dWh + (int64_t)r * HF + 4 * c16 + 64 * k), o);
          split_store4(a.at.dWh_p[mod], a.at.lo_off, (int64_t)(m0 + r) * HF + h * F + 4 * c16 + 64 * k, o);
        }
        dsp += z.x * w.x + z.y * w.y + z.z * w.z + z.w * w.w;
      }
      dsp = row16_sum(dsp);
      if (valid && c16 == 0) {
        const int sidx = (int)(((float)r + 0.5f) * inv_n), j = r - sidx * N;
        a.at.ds_part[mod][((int64_t)(b0 + sidx) * kHeads + h) * N + j] = dsp;
      }
    }
  }
  // B5b: d a_l[f] = sum_j du_j Wh[j][f], d a_r[f] = sum_j dv_j Wh[j][f] per sample: a thread per (sample, column), j ascending
  // (consecutive lanes read consecutive columns of a row: conflict free; du / dv are broadcast reads).  (As one more fp32-MFMA
  // pass -- rows 0 / 1 of a 16-row A operand -- this cost 2.6 us: four dependent chains per wave behind scalar LDS reads.)
  for (int e = tid; e < spt_valid * F; e += 768) {
    const int sidx = e / F, f = e - sidx * F;
    const float* wcol = shW + sidx * N * FS + f;
    float dl = 0.f, dr = 0.f;
    for (int j = 0; j < N; ++j) {
      const float w = wcol[j * FS];
      dl += shDu[sidx * N + j] * w;
      dr += shDv[sidx * N + j] * w;
    }
    float* da = a.at.da_part[mod] + ((int64_t)(b0 + sidx) * kHeads + h) * 2 * F + f;
    da[0] = dl;
    da[F] = dr;
  }
  GBCLK(7);
}

#ifdef DVGR_CLK
extern "C" int dvgr_dbg_gatb_clocks(long long* host_out, int n) {
  if (n > 1024 * 8) n = 1024 * 8;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gatb_clk), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

// ---- host side ----------------------------------------------------------------------------------------------------
static size_t gatb_lds_bytes(const GatfShape& s, int N, int terms) {
  constexpr int NTC = 3, F = 64 * NTC;
  const int RP = 16 * s.mtiles16, planes = terms == 3 ? 2 : 1;
  const size_t slot = (size_t)planes * ((size_t)RP * BK + (size_t)F * BK) * 2;      // bytes
  const size_t ring = (4 * slot <= 150 * 1024 ? 4 : 3) * slot;
  // (the kernel's own layout: shP, shW, a_l / a_r, the three N x N images, SIX per-row arrays shS .. shB -- GatbEpi::floats)
  const size_t epi = sizeof(float) * (s.mtiles16 == 2   ? GatbEpi<2, NTC>::floats(N, s.spt)
                                     : s.mtiles16 == 4 ? GatbEpi<4, NTC>::floats(N, s.spt)
                                     : s.mtiles16 == 5 ? GatbEpi<5, NTC>::floats(N, s.spt)
                                                       : GatbEpi<6, NTC>::floats(N, s.spt));
  return (epi > ring ? epi : ring) + 16;
}

bool gat_fused_bwd_supported(int B, int N, int D, int prec) {
  GatfShape s;
  if (prec == DVGR_F32 || D != 768 || !gatf_shape(B, N, &s) || (int64_t)2 * B * N * D >= (1ll << 31)) return false;
  if (s.nt > 2) return false;                              // (N > 32: the epilogue images of whole samples do not fit beside each other)
  return gatb_lds_bytes(s, N, prec == DVGR_BF16X3 ? 3 : 1) <= 160 * 1024;
}

// Whether the fused form is the FASTER one (supported shapes only).  One workgroup per (row tile, head, modality): from ~200
// workgroups up the launch fills the chip (C2, C3: 256 -- step -1.3 % / -1.6 %); at half of that (C4: 32 QA pairs of 16 clips
// = 128 workgroups) each epilogue runs alone on its compute unit while the other half of the chip idles, and the two launches
// with their 192 + 512 smaller workgroups are faster (+2.9 % fused); small batches (C1: 160 rows) are bound by the host's
// enqueue rate, where one launch fewer per layer wins (-7 %).  profiles/experiments_r05.tsv, fused_gat_bwd_*.
bool gat_fused_bwd_preferred(int B, int N, bool half_tiles_ok) {
  GatfShape s;
  if (!gatf_shape(B, N, &s)) return false;
  if (half_tiles_ok) gatf_half_shape(B, N, &s);            // (round 6: C4's shape runs 256 half-height workgroups, dvgr_gatf.h)
  const int total = ((B + s.spt - 1) / s.spt) * kHeads * 2;
  return total >= 200 || B * N <= 320;
}

template <int TERMS, int MTILES, int NT, bool FROM_DX, bool LO8, int KB = 32>
static int gatb_launch(hipStream_t st, const GatFusedBwdArgs& a, const GatfShape& s) {
  constexpr int NTC = 3;
  const size_t lds = gatb_lds_bytes(s, a.at.N, TERMS);
  if (lds > 160 * 1024) return DVGR_E_DIMS;
  // the raised dynamic-LDS limit belongs to the (function, device) pair: one flag per instantiation and device (benign race:
  // the attribute is idempotent)
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || !attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gat_fused_bwd<TERMS, MTILES, NT, NTC, FROM_DX, LO8, KB>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) attr_set[dev] = true;
  }
  const int mtiles = (a.at.B + a.spt - 1) / a.spt;
  const int total = mtiles * kHeads * 2;
  DVGR_LAUNCH_FORKABLE((k_gat_fused_bwd<TERMS, MTILES, NT, NTC, FROM_DX, LO8, KB>), dim3(8 * ((total + 7) / 8)), dim3(768), lds, st, a);
  DVGR_LAUNCH_CHECK();
}

template <int TERMS, bool LO8>
static int gatb_dispatch(hipStream_t st, const GatFusedBwdArgs& a, const GatfShape& s) {
  const bool fx = a.dx[0] != nullptr;
  if (s.mtiles16 == 4 && s.nt == 1) return fx ? gatb_launch<TERMS, 4, 1, true, LO8>(st, a, s) : gatb_launch<TERMS, 4, 1, false, LO8>(st, a, s);
  if (s.mtiles16 == 5 && s.nt == 2) return fx ? gatb_launch<TERMS, 5, 2, true, LO8>(st, a, s) : gatb_launch<TERMS, 5, 2, false, LO8>(st, a, s);
  return DVGR_E_DIMS;                                      // (6 row tiles, N = 21 .. 32: the images do not fit, gat_fused_bwd_supported)
}

int gat_fused_bwd(hipStream_t st, const GatFusedBwdArgs& a_in, int prec) {
  GatFusedBwdArgs a = a_in;
  GatfShape s;
  if (!gat_fused_bwd_supported(a.at.B, a.at.N, a.at.D, prec) || !gatf_shape(a.at.B, a.at.N, &s)) return DVGR_E_DIMS;
  for (int m = 0; m < 2; ++m) {
    if ((a.dx[m] != nullptr) != (a.dx[0] != nullptr)) return DVGR_E_NULL;
    if (a.dx[m] ? !a.beta[m] : !a.at.dz[m]) return DVGR_E_NULL;
    if (!a.ah[m] || !a.bh[m] || !a.at.out_c[m] || !a.at.out_s[m] || !a.at.Wh[m] || !a.at.s[m] || !a.at.u[m] ||
        !a.at.v[m] || !a.at.gat_a[m] || !a.at.gat_c[m] || !a.at.dWh_p[m] || !a.at.ds_part[m] || !a.at.da_part[m] || !a.at.dc_part[m])
      return DVGR_E_NULL;
    if (((uintptr_t)a.ah[m] | (uintptr_t)a.bh[m]) & 15) return DVGR_E_ALIGN;
  }
  if ((a.a_lo_off & 7) || (a.b_lo_off & 7) || (a.at.lo_off & 3)) return DVGR_E_ALIGN;
  a.spt = s.spt;
  if (a.kb64 && prec == DVGR_BF16X3 && a.dx[0] != nullptr && (a.at.D % 64) == 0) {      // 64-deep K-tiles, bf16 lo planes
    GatfShape hs = s;
    if (!a.no_half && gatf_half_shape(a.at.B, a.at.N, &hs)) {
      a.spt = hs.spt;
      return gatb_launch<3, 2, 1, true, false, 64>(st, a, hs);
    }
    if (s.mtiles16 == 4 && s.nt == 1) return gatb_launch<3, 4, 1, true, false, 64>(st, a, s);
    if (s.mtiles16 == 5 && s.nt == 2) return gatb_launch<3, 5, 2, true, false, 64>(st, a, s);
  }
  const bool lo8 = prec == DVGR_BF16X3 && a.bq[0] != nullptr;
  if (lo8) {
    for (int m = 0; m < 2; ++m)
      if (!a.bq[m] || !a.bscale[m] || ((uintptr_t)a.bq[m] & 15)) return DVGR_E_NULL;
    return gatb_dispatch<3, true>(st, a, s);
  }
  return prec == DVGR_BF16X3 ? gatb_dispatch<3, false>(st, a, s) : gatb_dispatch<1, false>(st, a, s);
}

}  // namespace dvgr
