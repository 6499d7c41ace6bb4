// dvgr_stages.hip -- row-wise, attention and element-wise kernels of the DualVGR unit.
// Algebra: SURVEY.md Appendix A (restated and checked in oracle/unit_manual.py).
#include "dvgr_stages.h"
#include "dvgr_rows.h"

namespace dvgr {

#define DVGR_LAUNCH_CHECK() return (int)hipGetLastError()

// Phase clocks (debug builds only, `make clk`): thread 0 of every workgroup stamps the 100 MHz wall clock at phase
// boundaries; tools/phase_clocks.py reads them back through dvgr_dbg_clocks.
#ifdef DVGR_CLK
__device__ long long g_clk[4096 * 8];
#define CLK(k)                                                                                          \
  do {                                                                                                  \
    const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                    \
    if (threadIdx.x == 0 && wg_ < 4096) g_clk[wg_ * 8 + (k)] = wall_clock64();                         \
  } while (0)
#else
#define CLK(k) do { } while (0)
#endif

// =================================================================================================
// S0 rows forward.  grid (B, L), 256 threads.  model/utils.py:68-82
// =================================================================================================
// pass 1: one wave per (layer, sample, word) row: L2-normalise, logit.  Every row is independent, so the grid is
// L*B*T/4 workgroups instead of the L*B a per-sample kernel would have.
__global__ __launch_bounds__(256) void k_qa_rows_norm(const QaRowsArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // (l * B + b) * T + t
  const int T = a.T, D = a.D, D4 = D / 4;
  if (row >= (int64_t)a.L * a.B * T) return;
  const int l = (int)(row / ((int64_t)a.B * T));
  const int bt = (int)(row - (int64_t)l * a.B * T);
  const float4* w2 = reinterpret_cast<const float4*>(a.params + (int64_t)l * a.layer_stride + a.off_w2);
  const float b2 = a.params[(int64_t)l * a.layer_stride + a.off_b2];
  float* sav = a.saved + (int64_t)l * a.s_layer_stride;
  const float4* y = reinterpret_cast<const float4*>(a.Y + row * D);
  float4* dsv = reinterpret_cast<float4*>(sav + a.s_d + (int64_t)bt * D);
  float4 v[4];                                              // D <= 1024 in registers; larger D re-reads the row
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane + 64 * i;
    v[i] = k < D4 ? y[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  }
  for (int k = lane + 256; k < D4; k += 64) { const float4 x = y[k]; ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);            // F.normalize eps
  float rr = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane + 64 * i;
    if (k < D4) {
      const float4 d = make_float4(v[i].x / nrm, v[i].y / nrm, v[i].z / nrm, v[i].w / nrm), w = w2[k];
      dsv[k] = d;
      rr += d.x * w.x + d.y * w.y + d.z * w.z + d.w * w.w;
    }
  }
  for (int k = lane + 256; k < D4; k += 64) {
    const float4 x = y[k], w = w2[k];
    const float4 d = make_float4(x.x / nrm, x.y / nrm, x.z / nrm, x.w / nrm);
    dsv[k] = d;
    rr += d.x * w.x + d.y * w.y + d.z * w.z + d.w * w.w;
  }
  rr = wave_sum(rr) + b2;
  if (lane == 0) { a.logit[row] = rr; sav[a.s_nrm + bt] = nrm; }
}

// pass 2: per (sample, layer): softmax over ALL T, mask, renormalise, q = sum_t ah_t words_t
__global__ __launch_bounds__(256) void k_qa_rows_attn(const QaRowsArgs a) {
  __shared__ float sh_r[256];
  __shared__ float sh_ah[256];
  const int b = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const int T = a.T, Wd = a.Wd;
  float* sav = a.saved + (int64_t)l * a.s_layer_stride;
  if (tid < T) sh_r[tid] = a.logit[((int64_t)l * a.B + b) * T + tid];
  __syncthreads();
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) mx = fmaxf(mx, sh_r[t]);
  float sum = 0.f;
  for (int t = 0; t < T; ++t) sum += expf(sh_r[t] - mx);
  const int len = (int)a.qlen[b];
  float S = 0.f;
  for (int t = 0; t < T && t < len; ++t) S += expf(sh_r[t] - mx) / sum;
  S += 1e-5f;
  if (tid < T) {
    const float av = expf(sh_r[tid] - mx) / sum;             // softmax over ALL T (padding included)
    const float ah = (tid < len ? av : 0.f) / S;             // mask, renormalise
    sav[a.s_a + b * T + tid] = av;
    sav[a.s_ah + b * T + tid] = ah;
    sh_ah[tid] = ah;
  }
  if (tid == 0) sav[a.s_S + b] = S;
  __syncthreads();
  for (int w = tid; w < Wd; w += 256) {
    float q = 0.f;
    for (int t = 0; t < T; ++t) q += sh_ah[t] * a.words[((int64_t)b * T + t) * Wd + w];
    sav[a.s_q + (int64_t)b * Wd + w] = q;
  }
}

int qa_rows_fwd(hipStream_t st, const QaRowsArgs& a) {
  const int64_t rows = (int64_t)a.L * a.B * a.T;
  DVGR_LAUNCH(k_qa_rows_norm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
  DVGR_LAUNCH(k_qa_rows_attn, dim3(a.B, a.L), dim3(256), 0, st, a);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// S1 forward: punish gate.  One wave per (b, n) row.  model/utils.py:101-103
// =================================================================================================
__global__ __launch_bounds__(256) void k_punish_fwd(const float* X0, const float* X1, const float* p0,
                                                    const float* p1, float* s0, float* s1, int rows, int N, int D) {
  const int mod = blockIdx.y;
  const float* X = mod ? X1 : X0;
  const float* p = mod ? p1 : p0;
  float* s = mod ? s1 : s0;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* x = reinterpret_cast<const float4*>(X + (int64_t)row * D);
  const float4* pb = reinterpret_cast<const float4*>(p + (int64_t)(row / N) * D);
  float acc = 0.f;
  for (int k = lane; k < D / 4; k += 64) {
    const float4 a = x[k], b = pb[k];
    acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  acc = wave_sum(acc);
  if (lane == 0) s[row] = sigmoid_f(acc);
}

int punish_fwd(hipStream_t st, const float* X0, const float* X1, const float* p0, const float* p1,
               float* s0, float* s1, int B, int N, int D) {
  const int rows = B * N;
  DVGR_LAUNCH(k_punish_fwd, dim3((rows + 3) / 4, 2), dim3(256), 0, st, X0, X1, p0, p1, s0, s1, rows, N, D);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// S3 forward: one workgroup per (b, head, modality).  model/GraphNN.py:97-111 in factorised form.
// =================================================================================================
// The aggregation alpha . (s * Wh) runs on the fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products): A = the dropped
// attention matrix (rows i, reduction j), B = the punished projections (reduction j, columns f).  NT = ceil(N / 16).
// LDS images are padded with zeros up to 16 NT rows / 4 S reduction steps so that fragments need no bounds checks:
//   shZ [4 S][F + 16]   row stride = 16 banks mod 64: the B fragment (4 rows x 16 columns per wave read) is conflict free
//   shA [16 NT][4 S + 1] odd row stride: the A fragment (16 rows x 4 columns) is conflict free
// The result tile goes back through LDS so that the ELU / dropout / planes epilogue stores whole 16-byte vectors.
typedef float gat_f32x4 __attribute__((ext_vector_type(4)));

// NW = waves per workgroup (8 since round 3; 4 before).  At N > 48 the LDS images allow two (forward) / one (backward)
// workgroups per CU, and four or eight waves per CU cannot hide a phase's latencies: C5 10.72 -> 9.71 ms per step with 8 waves
// (12 and 16 measure the same); at N <= 48 it is worth 0.4 % (C2) to 1.9 % (C4).  profiles/experiments_r03.tsv: *_wave_count.
template <int NT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_gat_attn_fwd(const GatAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, h = blockIdx.y, mod = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, D = a.D, F = D / 4, FS = F + 16, HF = kHeads * F;
  const int S = (N + 3) / 4, NP = 4 * S + 1, ZR = 4 * S;    // reduction steps, shA row stride, rows of shZ
  float* shZ = sm;                 // [ZR][FS]   Wh tile, later s_j * Wh_j, finally the output tile
  float* shA = shZ + ZR * FS;      // [16 NT][NP] dropped attention alpha_d[i][j]
  float* shAl = shA + 16 * NT * NP;  // [F]
  float* shAr = shAl + F;          // [F]
  float* shS = shAr + F;           // [N]
  float* shU = shS + N;            // [N]
  float* shV = shU + N;            // [N]
  const float* Wh = a.Wh[mod] + (int64_t)b * N * HF + h * F;
  CLK(0);
  {                                                         // the tile as float4s: a wave per row, a lane per float4 column
    const int F4 = F / 4;                                   // (no index division on this VALU-bound path); rows N .. ZR-1: zeros
    for (int j = wave; j < ZR; j += NW)
      for (int q = lane; q < F4; q += 64) {
        const float4 v = j < N ? *reinterpret_cast<const float4*>(Wh + (int64_t)j * HF + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(shZ + j * FS + 4 * q) = v;
      }
  }
  for (int e = tid; e < 16 * NT * NP; e += 64 * NW) shA[e] = 0.f;
  for (int f = tid; f < F; f += 64 * NW) {
    shAl[f] = a.gat_a[mod][h * 2 * F + f];
    shAr[f] = a.gat_a[mod][h * 2 * F + F + f];
  }
  if (tid < N) shS[tid] = a.s[mod][b * N + tid];
  __syncthreads();
  CLK(1);
  for (int i = wave; i < N; i += NW) {                      // u_i = a_l . Wh_i, v_i = a_r . Wh_i
    float pu = 0.f, pv = 0.f;
    for (int f = lane; f < F; f += 64) {
      const float w = shZ[i * FS + f];
      pu += w * shAl[f];
      pv += w * shAr[f];
    }
    pu = wave_sum(pu); pv = wave_sum(pv);
    if (lane == 0) {
      shU[i] = pu; shV[i] = pv;
      a.u[mod][((int64_t)b * kHeads + h) * N + i] = pu;
      a.v[mod][((int64_t)b * kHeads + h) * N + i] = pv;
    }
  }
  __syncthreads();
  CLK(2);
  {                                                         // softmax over j of LeakyReLU(u_i + v_j + c): one wave per row
    const float c = a.gat_c[mod][h];
    const uint32_t key = a.key_alpha[mod][h];
    for (int i = wave; i < N; i += NW) {
      float e = -INFINITY;
      if (lane < N) { e = shU[i] + shV[lane] + c; e = e > 0.f ? e : kSlope * e; }
      const float mx = wave_max(e);
      const float pe = lane < N ? expf(e - mx) : 0.f;
      const float sum = wave_sum(pe);
      if (lane < N) shA[i * NP + lane] = pe / sum * drop_factor(key, (uint32_t)((b * N + i) * N + lane));
    }
  }
  for (int j = wave; j < N; j += NW) {                      // punish AFTER the logits (GraphNN.py:102-104)
    const float sj = shS[j];
    for (int q = lane; q < F / 4; q += 64) {
      float4* p = reinterpret_cast<float4*>(shZ + j * FS + 4 * q);
      float4 v = *p;
      v.x *= sj; v.y *= sj; v.z *= sj; v.w *= sj;
      *p = v;
    }
  }
  __syncthreads();
  CLK(3);
  // ---- aggregation on the MFMA: wave w owns the column tiles J = w, w + 4, ... of the F / 16.  Only the owner of a
  // column tile ever reads those 16 columns of shZ, so it overwrites them with its result tile without a barrier. ----
  const int r16 = lane & 15, kq = lane >> 4, JT = F / 16;
  float af[NT][4 * NT];                                     // this lane's A fragments of every (row tile, k-step)
#pragma unroll
  for (int I = 0; I < NT; ++I)
#pragma unroll
    for (int s4 = 0; s4 < 4 * NT; ++s4) af[I][s4] = s4 < S ? shA[(16 * I + r16) * NP + 4 * s4 + kq] : 0.f;
  for (int J = wave; J < JT; J += NW) {
    gat_f32x4 acc[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) acc[I] = gat_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4 * NT; ++s4) {
      if (s4 < S) {
        const float bf = shZ[(4 * s4 + kq) * FS + 16 * J + r16];
#pragma unroll
        for (int I = 0; I < NT; ++I) acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[I][s4], bf, acc[I], 0, 0, 0);
      }
    }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + 4 * kq + r;
        if (row < N) shZ[row * FS + 16 * J + r16] = acc[I][r];
      }
  }
  __syncthreads();
  CLK(4);
  const int g = h >> 2, hl = h & 3;
  float* out = (g ? a.out_s[mod] : a.out_c[mod]) + (int64_t)b * N * D + hl * F;
  const uint32_t ko = a.key_out[mod][g];
  uint16_t* zp = g ? a.zp_s[mod] : a.zp_c[mod];
  for (int i = wave; i < N; i += NW)                         // ELU, dropout, fp32 + planes, 16 bytes per lane
    for (int q = lane; q < F / 4; q += 64) {
      const float4 v = *reinterpret_cast<const float4*>(shZ + i * FS + 4 * q);
      const int64_t o = (int64_t)i * D + 4 * q;
      const uint32_t e0 = (uint32_t)(((int64_t)b * N) * D + hl * F + o);
      float kf[4];
      drop_factor4(ko, e0, kf);
      float4 y;
      y.x = elu_fast(v.x) * kf[0];
      y.y = elu_fast(v.y) * kf[1];
      y.z = elu_fast(v.z) * kf[2];
      y.w = elu_fast(v.w) * kf[3];
      st4(reinterpret_cast<float4*>(out + o), y);
      split_store4(zp, a.z_lo_off, (int64_t)b * N * D + hl * F + o, y);
    }
  CLK(5);
}

static int gat_nt(int N) { return (N + 15) / 16; }
#ifndef DVGR_ATTN_WAVES_WIDE
#define DVGR_ATTN_WAVES_WIDE 8
#endif
constexpr int kAttnWavesWide = DVGR_ATTN_WAVES_WIDE;   // waves per workgroup of the NT = 4 (N > 48) attention kernels
#ifndef DVGR_ATTN_WAVES_SMALL
#define DVGR_ATTN_WAVES_SMALL 8
#endif
constexpr int kAttnWavesSmall = DVGR_ATTN_WAVES_SMALL; // ... of the NT <= 3 ones
static size_t gat_fwd_lds(int N, int F) {
  const size_t S = (N + 3) / 4, NT = gat_nt(N);
  return sizeof(float) * (4 * S * (F + 16) + 16 * NT * (4 * S + 1) + 2 * F + 3 * N + 16);
}
static int gat_nb(int N) { return N <= 8 ? 8 : N <= 16 ? 16 : N <= 24 ? 24 : N <= 32 ? 32 : 64; }

template <int NT>
static int gat_attn_fwd_nt(hipStream_t st, const GatAttnArgs& a) {
  constexpr int NW = NT == 4 ? kAttnWavesWide : kAttnWavesSmall;
  const size_t lds = gat_fwd_lds(a.N, a.D / 4);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gat_attn_fwd<NT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DVGR_LAUNCH((k_gat_attn_fwd<NT, NW>), dim3(a.B, kHeads, 2), dim3(64 * NW), lds, st, a);
  DVGR_LAUNCH_CHECK();
}

int gat_attn_fwd(hipStream_t st, const GatAttnArgs& a) {
  switch (gat_nt(a.N)) {
    case 1: return gat_attn_fwd_nt<1>(st, a);
    case 2: return gat_attn_fwd_nt<2>(st, a);
    case 3: return gat_attn_fwd_nt<3>(st, a);
    default: return gat_attn_fwd_nt<4>(st, a);
  }
}

// =================================================================================================
// S5 forward: one wave per (b, n) row.  model/Attention.py:20-23 + model/models.py:168-169
// =================================================================================================
__global__ __launch_bounds__(256) void k_sf_combine_fwd(const SfCombineArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  sf_combine_row(a, blockIdx.y, row, threadIdx.x & 63);
}

int sf_combine_fwd(hipStream_t st, const SfCombineArgs& a) {
  DVGR_LAUNCH(k_sf_combine_fwd, dim3((a.rows + 3) / 4, 2), dim3(256), 0, st, a);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// S6 pooling.  model/fusions/fusions.py:430-440
// =================================================================================================
__global__ __launch_bounds__(256) void k_mfb_pool_fwd(const float* x0, const float* x1, float* z, int64_t n,
                                                      uint16_t* zp, int64_t lo_off) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index into z (rows * 256)
  if (i >= n) return;
  const float2 a = reinterpret_cast<const float2*>(x0)[i];
  const float2 b = reinterpret_cast<const float2*>(x1)[i];
  const float v = a.x * b.x + a.y * b.y;
  z[i] = v;
  split_store1(zp, lo_off, i, v);
}
int mfb_pool_fwd(hipStream_t st, const float* x0, const float* x1, float* z, int rows, uint16_t* zp, int64_t lo_off) {
  const int64_t n = (int64_t)rows * kMfbPooled;
  DVGR_LAUNCH(k_mfb_pool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x0, x1, z, n, zp, lo_off);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// Backward element-wise pieces
// =================================================================================================
__global__ __launch_bounds__(256) void k_elu_bwd(const float* g, const float* y, float* out, int64_t n4,
                                                 uint16_t* op, int64_t lo_off) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index
  if (i >= n4) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g) {
    const float4 gg = reinterpret_cast<const float4*>(g)[i], yy = reinterpret_cast<const float4*>(y)[i];
    v = make_float4(gg.x * elu_grad_from_out(yy.x), gg.y * elu_grad_from_out(yy.y), gg.z * elu_grad_from_out(yy.z),
                    gg.w * elu_grad_from_out(yy.w));
  }
  reinterpret_cast<float4*>(out)[i] = v;
  split_store4(op, lo_off, 4 * i, v);
}
int elu_bwd(hipStream_t st, const float* g, const float* y, float* out, int64_t n, uint16_t* op, int64_t lo_off) {
  const int64_t n4 = n / 4;
  DVGR_LAUNCH(k_elu_bwd, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, g, y, out, n4, op, lo_off);
  DVGR_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void k_mfb_pool_bwd(const float* dz, const float* x0, const float* x1,
                                                      float* dp0, float* dp1, int64_t n, uint16_t* p0, uint16_t* p1,
                                                      int64_t lo_off) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index into dz (rows * 256)
  if (i >= n) return;
  const float d = dz[i];
  const float2 a = reinterpret_cast<const float2*>(x0)[i];
  const float2 b = reinterpret_cast<const float2*>(x1)[i];
  float2 r0, r1;
  r0.x = d * b.x * elu_grad_from_out(a.x); r0.y = d * b.y * elu_grad_from_out(a.y);
  r1.x = d * a.x * elu_grad_from_out(b.x); r1.y = d * a.y * elu_grad_from_out(b.y);
  reinterpret_cast<float2*>(dp0)[i] = r0;
  reinterpret_cast<float2*>(dp1)[i] = r1;
  split_store1(p0, lo_off, 2 * i, r0.x); split_store1(p0, lo_off, 2 * i + 1, r0.y);
  split_store1(p1, lo_off, 2 * i, r1.x); split_store1(p1, lo_off, 2 * i + 1, r1.y);
}
int mfb_pool_bwd(hipStream_t st, const float* dz, const float* x0, const float* x1, float* dp0, float* dp1, int rows,
                 uint16_t* p0, uint16_t* p1, int64_t lo_off) {
  const int64_t n = (int64_t)rows * kMfbPooled;
  DVGR_LAUNCH_FORKABLE(k_mfb_pool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dz, x0, x1, dp0, dp1, n, p0, p1, lo_off);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// S5/S4 backward (row part).  One wave per row.  SURVEY.md Appendix A.4
// =================================================================================================
__global__ __launch_bounds__(256) void k_sf_bwd(const SfBwdArgs a) {
  const int mod = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  const int D = a.D, D4 = D / 4;
  const int64_t r0 = (int64_t)row * D, r1 = ((int64_t)a.rows + row) * D;
  const float4* dX = reinterpret_cast<const float4*>(a.dX[mod] + r0);
  const float4* ge = a.g_embed[mod] ? reinterpret_cast<const float4*>(a.g_embed[mod] + r0) : nullptr;
  const float4* z0 = reinterpret_cast<const float4*>(a.z0[mod] + r0);
  const float4* z1 = reinterpret_cast<const float4*>(a.z1[mod] + r0);
  // the dX half of the layer above's S1 backward, on the row while it is read (SfBwdArgs::fin_*): d = dX + add + dtau p.  Where
  // the row does not stay in registers both passes form d from the SAME inputs (no read of a value this launch stored); the
  // second pass stores it.
  const bool fin = a.fin_dX[mod] != nullptr;               // (uniform)
  float dtau = 0.f;
  const float4* fadd = nullptr;
  const float4* fp = nullptr;
  if (fin) {
    const int bs = row / a.fin_N, n = row - bs * a.fin_N;
    float part = lane < kHeads ? a.fin_ds_part[mod][((int64_t)bs * kHeads + lane) * a.fin_N + n] : 0.f;
    part = wave_sum(part);
    const float sg = a.fin_s[mod][row];
    dtau = part * sg * (1.f - sg);
    fadd = a.fin_add[mod] ? reinterpret_cast<const float4*>(a.fin_add[mod] + r0) : nullptr;
    fp = reinterpret_cast<const float4*>(a.fin_p[mod] + (int64_t)bs * D);
  }
  auto row_grad = [&](int k) -> float4 {
    float4 d = dX[k];
    if (fin) {
      const float4 pp = fp[k];
      if (fadd) { const float4 e = fadd[k]; d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
      d.x += dtau * pp.x; d.y += dtau * pp.y; d.z += dtau * pp.z; d.w += dtau * pp.w;
    }
    return d;
  };
  // The row's gradient d stays in registers between the two passes where the row fits (D <= 768: three float4 per lane) --
  // the second pass then re-reads neither dX nor the folded-in tensors (round 6; until then dX came back from L2).
  constexpr int KC3 = 3;
  const bool cached = D4 <= 64 * KC3;                      // (uniform)
  float4 dreg[KC3];
#pragma unroll
  for (int it = 0; it < KC3; ++it) dreg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  float db0 = 0.f, db1 = 0.f;
  auto dots = [&](int k, float4 d) {
    if (ge) { const float4 e = ge[k]; d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
    const float4 u = z0[k], v = z1[k];
    db0 += d.x * u.x + d.y * u.y + d.z * u.z + d.w * u.w;
    db1 += d.x * v.x + d.y * v.y + d.z * v.z + d.w * v.w;
  };
  if (cached) {
#pragma unroll
    for (int it = 0; it < KC3; ++it) {
      const int k = lane + 64 * it;
      if (k < D4) { dreg[it] = row_grad(k); dots(k, dreg[it]); }
    }
  } else {
    for (int k = lane; k < D4; k += 64) dots(k, row_grad(k));
  }
  db0 = wave_sum(db0); db1 = wave_sum(db1);
  const float b0 = a.beta[mod][row * 2], b1 = a.beta[mod][row * 2 + 1];
  const float mean = b0 * db0 + b1 * db1;
  const float dw0 = b0 * (db0 - mean), dw1 = b1 * (db1 - mean);
  if (lane == 0) { a.dw[mod][row] = dw0; a.dw[mod][a.rows + row] = dw1; }
  const float4* t0 = reinterpret_cast<const float4*>(a.tn[mod] + r0);
  const float4* t1 = reinterpret_cast<const float4*>(a.tn[mod] + r1);
  const float4* p2 = reinterpret_cast<const float4*>(a.p2[mod]);
  const bool wdz = a.dz[mod] != nullptr;                   // (uniform) false: the fused GAT-view backward forms dz's pre-part itself
  const float4* g0 = (wdz && a.g_z0[mod]) ? reinterpret_cast<const float4*>(a.g_z0[mod] + r0) : nullptr;
  const float4* g1 = (wdz && a.g_z1[mod]) ? reinterpret_cast<const float4*>(a.g_z1[mod] + r0) : nullptr;
  float4* dpre0 = a.dpre[mod] ? reinterpret_cast<float4*>(a.dpre[mod] + r0) : nullptr;   // fp32 copy: only where it has a reader
  float4* dpre1 = a.dpre[mod] ? reinterpret_cast<float4*>(a.dpre[mod] + r1) : nullptr;
  float4* dz0 = wdz ? reinterpret_cast<float4*>(a.dz[mod] + r0) : nullptr;
  float4* dz1 = wdz ? reinterpret_cast<float4*>(a.dz[mod] + r1) : nullptr;
  float4* dXf = fin ? reinterpret_cast<float4*>(a.fin_dX[mod] + r0) : nullptr;
  auto outputs = [&](int k, float4 d) {
    if (fin) dXf[k] = d;                                   // the completed running gradient (read by the fused backward's B0 and the dX contraction)
    if (ge) { const float4 e = ge[k]; d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
    const float4 ta = t0[k], tb = t1[k], p = p2[k];
    const float4 q0 = make_float4(dw0 * p.x * (1.f - ta.x * ta.x), dw0 * p.y * (1.f - ta.y * ta.y),
                                  dw0 * p.z * (1.f - ta.z * ta.z), dw0 * p.w * (1.f - ta.w * ta.w));
    const float4 q1 = make_float4(dw1 * p.x * (1.f - tb.x * tb.x), dw1 * p.y * (1.f - tb.y * tb.y),
                                  dw1 * p.z * (1.f - tb.z * tb.z), dw1 * p.w * (1.f - tb.w * tb.w));
    if (dpre0) { st4(dpre0 + k, q0); st4(dpre1 + k, q1); }
    split_store4(a.dpre_p[mod], a.lo_off, r0 + 4 * k, q0);
    split_store4(a.dpre_p[mod], a.lo_off, r1 + 4 * k, q1);
    if (wdz) {
      float4 o0 = make_float4(b0 * d.x, b0 * d.y, b0 * d.z, b0 * d.w);
      float4 o1 = make_float4(b1 * d.x, b1 * d.y, b1 * d.z, b1 * d.w);
      if (g0) { const float4 e = g0[k]; o0.x += e.x; o0.y += e.y; o0.z += e.z; o0.w += e.w; }
      if (g1) { const float4 e = g1[k]; o1.x += e.x; o1.y += e.y; o1.z += e.z; o1.w += e.w; }
      st4(dz0 + k, o0);
      st4(dz1 + k, o1);
    }
  };
  if (cached) {
#pragma unroll
    for (int it = 0; it < KC3; ++it) {
      const int k = lane + 64 * it;
      if (k < D4) outputs(k, dreg[it]);
    }
  } else {
    for (int k = lane; k < D4; k += 64) outputs(k, row_grad(k));
  }
}
// (Round 6 also tried this pass as straight-line code for D = 768 -- FIN / ADD / GE compile-time, every load of a pass issued before
// the first wait, 2 memory round trips per row instead of ~10 behind the run-time branches above: the launch measures the same,
// 29.6 against 30.2 us beside the weight-gradient launch (rocprofv3), the step +-0: the kernel runs at the share of the memory
// system the side stream leaves it, not at its own latency chain.  Not kept.)
int sf_bwd(hipStream_t st, const SfBwdArgs& a) {
  DVGR_LAUNCH(k_sf_bwd, dim3((a.rows + 3) / 4, 2), dim3(256), 0, st, a);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// S3 backward: one workgroup per (b, head, modality).  SURVEY.md Appendix A.3
// =================================================================================================
// Both contractions run on the fp32 MFMA (exact fp32 products), NT = ceil(N / 16), S = ceil(N / 4):
//   dalpha_raw[i][j] = dP_i . Wh_j       A = dP rows, B = Wh rows, reduction over F in float4 slots (as k_aux_gram)
//   dZ[j][f] = sum_i alpha_d[i][j] dP[i][f]   A = alpha_d^T (registers), B = dP; a wave owns whole 16-column strips of f,
//              so it overwrites its strip of shP with dZ without a barrier (only the owner reads those columns)
// LDS images are zero padded to 16 NT rows (shW, shP) / 4 S rows and 16 NT columns (shAd) so fragments need no guards.
// TS = float4 column slots per lane in the epilogue: 1 when F <= 256 (D = 768: F4 = 48 of 64 lanes), else 2.  With one slot the
// kernel needs <= 64 VGPRs, i.e. FOUR 8-wave workgroups per CU (what its 39 KB of LDS images allow at N = 20) instead of three:
// the 1024 workgroups of a C2 launch then run as one round instead of 1.33.
template <int NT, int NW = 4, int TS = 2>
__global__ __launch_bounds__(64 * NW) void k_gat_attn_bwd(const GatAttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, h = blockIdx.y, mod = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int N = a.N, D = a.D, F = D / 4, FS = F + 4, HF = kHeads * F, F4 = F / 4;   // FS: 16-byte aligned rows
  const int S = (N + 3) / 4, NP = N + 1;                    // k-steps over the nodes; row stride of the N x N images
  // Fragment rows / columns beyond N are redirected to one all-zero row (index N) / column (index N) instead of padding
  // every image to 16 NT rows: at N = 20 that is 39 KB of LDS per workgroup (4 per CU) instead of 62 KB (2 per CU).
  const int WR = ((N + 1) * FS > 2 * NW * F) ? (N + 1) * FS : 2 * NW * F;   // shW doubles as the d a_l / d a_r partial buffer at the end
  const int g = h >> 2, hl = h & 3;
  float* shW = sm;                 // [N + 1][FS] Wh (unscaled); row N = 0
  float* shP = shW + WR;           // [N + 1][FS] dP, later dZ; row N = 0
  float* shA = shP + (N + 1) * FS; // [N][NP]     alpha
  float* shAd = shA + N * NP;      // [N + 1][NP] alpha * keep; row N and column N = 0
  float* shG = shAd + (N + 1) * NP;  // [N][NP]   dalpha, then dpre
  float* shS = shG + N * NP;       // [N]
  float* shU = shS + N;            // [N]
  float* shV = shU + N;            // [N]
  float* shDu = shV + N;           // [N]
  float* shDv = shDu + N;          // [N]
  float* shRed = shDv + N;         // [NW <= 16]
  const float* Wh = a.Wh[mod] + (int64_t)b * N * HF + h * F;
  const float* dz = a.dz[mod] + (int64_t)g * a.B * N * D + (int64_t)b * N * D + hl * F;   // view g of (2, B*N, D)
  const float* outp = (g ? a.out_s[mod] : a.out_c[mod]) + (int64_t)b * N * D + hl * F;
  const uint32_t ko = a.key_out[mod][g];
  const uint32_t ka = a.key_alpha[mod][h];
  CLK(0);
  {                                                         // three tiles as float4s: a wave per row, a lane per float4 column
    const float sc = ko ? (1.0f - kDropP) : 1.0f;           // un-dropped ELU output where kept
    for (int i = wave; i <= N; i += NW)                      // row N is the zero row
      for (int q = lane; q < F4; q += 64) {
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), wv = p;
        if (i < N) {
          wv = *reinterpret_cast<const float4*>(Wh + (int64_t)i * HF + 4 * q);
          const float4 z = *reinterpret_cast<const float4*>(dz + (int64_t)i * D + 4 * q);
          const float4 y = *reinterpret_cast<const float4*>(outp + (int64_t)i * D + 4 * q);
          float kf[4];
          drop_factor4(ko, (uint32_t)(((int64_t)b * N + i) * D + hl * F + 4 * q), kf);
          p.x = z.x * kf[0] * elu_grad_from_out(y.x * sc);
          p.y = z.y * kf[1] * elu_grad_from_out(y.y * sc);
          p.z = z.z * kf[2] * elu_grad_from_out(y.z * sc);
          p.w = z.w * kf[3] * elu_grad_from_out(y.w * sc);
        }
        *reinterpret_cast<float4*>(shW + i * FS + 4 * q) = wv;
        *reinterpret_cast<float4*>(shP + i * FS + 4 * q) = p;
      }
  }
  for (int e = tid; e < (N + 1) * NP; e += 64 * NW) shAd[e] = 0.f;
  if (tid < N) {
    shS[tid] = a.s[mod][b * N + tid];
    shU[tid] = a.u[mod][((int64_t)b * kHeads + h) * N + tid];
    shV[tid] = a.v[mod][((int64_t)b * kHeads + h) * N + tid];
  }
  __syncthreads();
  CLK(1);
  const float c = a.gat_c[mod][h];
  for (int i = wave; i < N; i += NW) {                      // recompute alpha and alpha * keep: one wave per row
    float e = -INFINITY;
    if (lane < N) { e = shU[i] + shV[lane] + c; e = e > 0.f ? e : kSlope * e; }
    const float mx = wave_max(e);
    const float pe = lane < N ? expf(e - mx) : 0.f;
    const float sum = wave_sum(pe);
    if (lane < N) {
      const float al = pe / sum;
      shA[i * NP + lane] = al;
      shAd[i * NP + lane] = al * drop_factor(ka, (uint32_t)((b * N + i) * N + lane));
    }
  }
  CLK(2);
  // ---- dalpha_ij = keep_ij * s_j * (dP_i . Wh_j): NT x NT tiles dealt round-robin to the waves ----
  for (int tile = wave; tile < NT * NT; tile += NW) {
    const int I = tile / NT, J = tile - I * NT;
    gat_f32x4 acc = gat_f32x4{0.f, 0.f, 0.f, 0.f};
    const int ra = 16 * I + r16, rb = 16 * J + r16;
    const float* pa = shP + (ra < N ? ra : N) * FS + 4 * kq;
    const float* pb = shW + (rb < N ? rb : N) * FS + 4 * kq;
    for (int t = 0; t < F / 16; ++t) {
      const float4 x = *reinterpret_cast<const float4*>(pa + 16 * t), y = *reinterpret_cast<const float4*>(pb + 16 * t);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, acc, 0, 0, 0);
    }
    const int j = 16 * J + r16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * I + 4 * kq + r;
      if (i < N && j < N) shG[i * NP + j] = acc[r] * shS[j] * drop_factor(ka, (uint32_t)((b * N + i) * N + j));
    }
  }
  __syncthreads();
  CLK(3);
  float dc_w = 0.f;
  for (int i = wave; i < N; i += NW) {                      // softmax and LeakyReLU backward, du_i: one wave per row
    const float al = lane < N ? shA[i * NP + lane] : 0.f;
    const float da = lane < N ? shG[i * NP + lane] : 0.f;
    const float rowdot = wave_sum(al * da);
    const float pre = lane < N ? shU[i] + shV[lane] + c : 0.f;
    const float dpre = al * (da - rowdot) * (pre > 0.f ? 1.f : kSlope);
    if (lane < N) shG[i * NP + lane] = dpre;
    const float du = wave_sum(lane < N ? dpre : 0.f);
    if (lane == 0) shDu[i] = du;
    dc_w += du;
  }
  if (lane == 0) shRed[wave] = dc_w;
  __syncthreads();
  if (tid < N) {                                            // dv_j
    float dv = 0.f;
    for (int i = 0; i < N; ++i) dv += shG[i * NP + tid];
    shDv[tid] = dv;
  }
  if (tid == 0) {
    float dc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) dc += shRed[w];
    a.dc_part[mod][b * kHeads + h] = dc;
  }
  CLK(4);
  // ---- dZ = alpha_d^T . dP: wave w owns the column strips Ft = w, w + 4, ... and overwrites them in shP ----
  {
    float af[NT][4 * NT];                                   // A fragments: alpha_d[4 s + kq][16 Jt + r16]
#pragma unroll
    for (int Jt = 0; Jt < NT; ++Jt)
#pragma unroll
      for (int s4 = 0; s4 < 4 * NT; ++s4) {
        const int i = 4 * s4 + kq, j = 16 * Jt + r16;
        af[Jt][s4] = s4 < S ? shAd[(i < N ? i : N) * NP + (j < N ? j : N)] : 0.f;
      }
    for (int Ft = wave; Ft < F / 16; Ft += NW) {
      gat_f32x4 acc[NT];
#pragma unroll
      for (int Jt = 0; Jt < NT; ++Jt) acc[Jt] = gat_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4 * NT; ++s4) {
        if (s4 < S) {
          const int i = 4 * s4 + kq;
          const float bf = shP[(i < N ? i : N) * FS + 16 * Ft + r16];
#pragma unroll
          for (int Jt = 0; Jt < NT; ++Jt) acc[Jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[Jt][s4], bf, acc[Jt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int Jt = 0; Jt < NT; ++Jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * Jt + 4 * kq + r;
          if (j < N) shP[j * FS + 16 * Ft + r16] = acc[Jt][r];
        }
    }
  }
  __syncthreads();
  CLK(5);
  // ---- epilogue, one wave per row j: dWh_j = s_j dZ_j + du_j a_l + dv_j a_r (fp32 + planes), ds_j = dZ_j . Wh_j,
  //      d a_l += du_j Wh_j, d a_r += dv_j Wh_j (per-wave partials in registers, then summed through LDS) ----
  float* dWh = a.dWh[mod] ? a.dWh[mod] + (int64_t)b * N * HF + h * F : nullptr;   // fp32 copy: only where it has a reader
  float4 pal[TS], par[TS], gal[TS], gar[TS];                    // this lane's columns 4 (lane + 64 t) .. + 3, t < 2 (F <= 512)
#pragma unroll
  for (int t = 0; t < TS; ++t) {
    pal[t] = make_float4(0.f, 0.f, 0.f, 0.f); par[t] = pal[t]; gal[t] = pal[t]; gar[t] = pal[t];
    const int q = lane + 64 * t;
    if (q < F4) {
      gal[t] = *reinterpret_cast<const float4*>(a.gat_a[mod] + h * 2 * F + 4 * q);
      gar[t] = *reinterpret_cast<const float4*>(a.gat_a[mod] + h * 2 * F + F + 4 * q);
    }
  }
  for (int j = wave; j < N; j += NW) {
    const float sj = shS[j], duj = shDu[j], dvj = shDv[j];
    float dsp = 0.f;
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      const int q = lane + 64 * t;
      if (q < F4) {
        const float4 z = *reinterpret_cast<const float4*>(shP + j * FS + 4 * q), w = *reinterpret_cast<const float4*>(shW + j * FS + 4 * q);
        const float4 al = gal[t], ar = gar[t];
        const float4 o = make_float4(sj * z.x + duj * al.x + dvj * ar.x, sj * z.y + duj * al.y + dvj * ar.y,
                                     sj * z.z + duj * al.z + dvj * ar.z, sj * z.w + duj * al.w + dvj * ar.w);
        if (dWh) st4(reinterpret_cast<float4*>(dWh + (int64_t)j * HF + 4 * q), o);
        split_store4(a.dWh_p[mod], a.lo_off, (int64_t)b * N * HF + h * F + (int64_t)j * HF + 4 * q, o);
        dsp += z.x * w.x + z.y * w.y + z.z * w.z + z.w * w.w;
        pal[t].x += duj * w.x; pal[t].y += duj * w.y; pal[t].z += duj * w.z; pal[t].w += duj * w.w;
        par[t].x += dvj * w.x; par[t].y += dvj * w.y; par[t].z += dvj * w.z; par[t].w += dvj * w.w;
      }
    }
    dsp = wave_sum(dsp);
    if (lane == 0) a.ds_part[mod][((int64_t)b * kHeads + h) * N + j] = dsp;
  }
  __syncthreads();                                          // every wave is done with shW: it now holds the partials
  float* shDa = shW;                                        // [NW][2 F]
#pragma unroll
  for (int t = 0; t < TS; ++t) {
    const int q = lane + 64 * t;
    if (q < F4) {
      *reinterpret_cast<float4*>(shDa + wave * 2 * F + 4 * q) = pal[t];
      *reinterpret_cast<float4*>(shDa + wave * 2 * F + F + 4 * q) = par[t];
    }
  }
  __syncthreads();
  for (int f = tid; f < 2 * F; f += 64 * NW) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 2) acc += shDa[w * 2 * F + f] + shDa[(w + 1) * 2 * F + f];
    a.da_part[mod][((int64_t)b * kHeads + h) * 2 * F + f] = acc;
  }
  CLK(6);
}

// =================================================================================================
// S3 backward for N > 32, column-chunked (round 4).  The kernel above keeps the whole Wh and dP tiles of its (sample, head) in
// LDS next to three N x N images: 153 KB at N = 64 -- ONE 8-wave workgroup per CU, whose load phase (150 KB at ~20 GB/s per
// CU), two MFMA phases and row phases run strictly one after the other (the largest single item of a C5 step: 1.5 of 9.6 ms).
// Here the F columns go through LDS in chunks of 64 (two 17 KB images), twice:
//   pass 1  dalpha = dP . Wh^T accumulated over the chunks in registers (two 16 x 16 tiles per wave),
//   then    softmax / LeakyReLU backward on the N x N images (alpha is kept, alpha * keep is rebuilt from the counter hash),
//   pass 2  per chunk: dZ = alpha_d^T . dP, dWh = s dZ + du a_l + dv a_r (fp32 / planes), ds += dZ . Wh, d a_l / d a_r partials.
// 74 KB at N = 64: TWO workgroups per CU, so one's loads run under the other's MFMA / row phases.  The price is a second read
// of dz, out and Wh (served by L2 / the Infinity Cache: the whole working set of a launch is 150 MB).
// =================================================================================================
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_gat_attn_bwd_chunked(const GatAttnBwdArgs a) {   // 4 waves per SIMD = two workgroups per CU: <= 128 VGPRs
  static_assert(NW == 8, "wave -> (strip, row-tile half) map of pass 2");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int FC = 64, FCS = FC + 4;                     // chunk width, row stride of the chunk images
  const int b = blockIdx.x, h = blockIdx.y, mod = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int N = a.N, D = a.D, F = D / 4, HF = kHeads * F;
  const int S = (N + 3) / 4, NP = N + 1;
  const int g = h >> 2, hl = h & 3;
  float* shW = sm;                    // [N + 1][FCS] Wh chunk; row N = 0
  float* shP = shW + (N + 1) * FCS;   // [N + 1][FCS] dP chunk, later dZ chunk; row N = 0
  float* shA = shP + (N + 1) * FCS;   // [N][NP] alpha
  float* shG = shA + N * NP;          // [N][NP] dalpha, then dpre
  float* shS = shG + N * NP;          // [N]
  float* shU = shS + N;
  float* shV = shU + N;
  float* shDu = shV + N;
  float* shDv = shDu + N;
  float* shDs = shDv + N;             // [N] ds accumulated over the chunks
  float* shRed = shDs + N;            // [16]
  float* shDa = shRed + 16;           // [NW][2 FC] partials of d a_l / d a_r of one chunk
  const float* Wh = a.Wh[mod] + (int64_t)b * N * HF + h * F;
  const float* dz = a.dz[mod] + (int64_t)g * a.B * N * D + (int64_t)b * N * D + hl * F;
  const float* outp = (g ? a.out_s[mod] : a.out_c[mod]) + (int64_t)b * N * D + hl * F;
  const uint32_t ko = a.key_out[mod][g];
  const uint32_t ka = a.key_alpha[mod][h];
  const float sc = ko ? (1.0f - kDropP) : 1.0f;           // un-dropped ELU output where kept
  // chunk c of both tiles as float4s: four rows per wave instruction (16 lanes x 16 B = a whole 256-byte row piece each)
  auto load_chunk = [&](int c) {
    for (int i = 4 * wave + kq; i <= N; i += 4 * NW) {      // row N is the zero row
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f), wv = p;
      if (i < N) {
        const int col = FC * c + 4 * r16;
        wv = *reinterpret_cast<const float4*>(Wh + (int64_t)i * HF + col);
        const float4 z = *reinterpret_cast<const float4*>(dz + (int64_t)i * D + col);
        const float4 y = *reinterpret_cast<const float4*>(outp + (int64_t)i * D + col);
        float kf[4];
        drop_factor4(ko, (uint32_t)(((int64_t)b * N + i) * D + hl * F + col), kf);
        p.x = z.x * kf[0] * elu_grad_from_out(y.x * sc);
        p.y = z.y * kf[1] * elu_grad_from_out(y.y * sc);
        p.z = z.z * kf[2] * elu_grad_from_out(y.z * sc);
        p.w = z.w * kf[3] * elu_grad_from_out(y.w * sc);
      }
      *reinterpret_cast<float4*>(shW + i * FCS + 4 * r16) = wv;
      *reinterpret_cast<float4*>(shP + i * FCS + 4 * r16) = p;
    }
  };
  if (tid < N) {
    shS[tid] = a.s[mod][b * N + tid];
    shU[tid] = a.u[mod][((int64_t)b * kHeads + h) * N + tid];
    shV[tid] = a.v[mod][((int64_t)b * kHeads + h) * N + tid];
    shDs[tid] = 0.f;
  }
  load_chunk(0);
  __syncthreads();
  const float c = a.gat_c[mod][h];
  for (int i = wave; i < N; i += NW) {                      // recompute alpha: one wave per row
    float e = -INFINITY;
    if (lane < N) { e = shU[i] + shV[lane] + c; e = e > 0.f ? e : kSlope * e; }
    const float mx = wave_max(e);
    const float pe = lane < N ? expf(e - mx) : 0.f;
    const float sum = wave_sum(pe);
    if (lane < N) shA[i * NP + lane] = pe / sum;
  }
  // ---- pass 1: dalpha_ij = keep_ij s_j (dP_i . Wh_j), tiles (I, J) dealt to the waves, accumulated over the chunks ----
  constexpr int TPW = (NT * NT + NW - 1) / NW;              // tiles per wave
  gat_f32x4 dacc[TPW];
#pragma unroll
  for (int k = 0; k < TPW; ++k) dacc[k] = gat_f32x4{0.f, 0.f, 0.f, 0.f};
  const int nch = F / FC;
  for (int cidx = 0; cidx < nch; ++cidx) {
    if (cidx > 0) { load_chunk(cidx); __syncthreads(); }
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int tile = wave + NW * k;
      if (tile < NT * NT) {
        const int I = tile / NT, J = tile - I * NT;
        const int ra = 16 * I + r16, rb = 16 * J + r16;
        const float* pa = shP + (ra < N ? ra : N) * FCS + 4 * kq;
        const float* pb = shW + (rb < N ? rb : N) * FCS + 4 * kq;
#pragma unroll
        for (int t = 0; t < FC / 16; ++t) {
          const float4 x = *reinterpret_cast<const float4*>(pa + 16 * t), y = *reinterpret_cast<const float4*>(pb + 16 * t);
          dacc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, dacc[k], 0, 0, 0);
          dacc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, dacc[k], 0, 0, 0);
          dacc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, dacc[k], 0, 0, 0);
          dacc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, dacc[k], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                        // the chunk images may be overwritten
  }
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int tile = wave + NW * k;
    if (tile < NT * NT) {
      const int I = tile / NT, J = tile - I * NT, j = 16 * J + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * kq + r;
        if (i < N && j < N) shG[i * NP + j] = dacc[k][r] * shS[j] * drop_factor(ka, (uint32_t)((b * N + i) * N + j));
      }
    }
  }
  __syncthreads();
  // ---- softmax and LeakyReLU backward, du_i: one wave per row; then dv_j, dc ----
  float dc_w = 0.f;
  for (int i = wave; i < N; i += NW) {
    const float al = lane < N ? shA[i * NP + lane] : 0.f;
    const float da = lane < N ? shG[i * NP + lane] : 0.f;
    const float rowdot = wave_sum(al * da);
    const float pre = lane < N ? shU[i] + shV[lane] + c : 0.f;
    const float dpre = al * (da - rowdot) * (pre > 0.f ? 1.f : kSlope);
    if (lane < N) shG[i * NP + lane] = dpre;
    const float du = wave_sum(lane < N ? dpre : 0.f);
    if (lane == 0) shDu[i] = du;
    dc_w += du;
  }
  if (lane == 0) shRed[wave] = dc_w;
  __syncthreads();
  if (tid < N) {
    float dv = 0.f;
    for (int i = 0; i < N; ++i) dv += shG[i * NP + tid];
    shDv[tid] = dv;
  }
  if (tid == 0) {
    float dc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) dc += shRed[w];
    a.dc_part[mod][b * kHeads + h] = dc;
  }
  // ---- pass 2.  Wave w = (strip Ft = w & 3 of the chunk's four 16-column strips, half Jh = w >> 2 of the row tiles of dZ) ----
  constexpr int JH = (NT + 1) / 2;                          // row tiles per half
  const int Ft = wave & 3, Jh = wave >> 2;
  float af[JH][4 * NT];                                     // A fragments: alpha_d[i = 4 s4 + kq][j = 16 Jt + r16], Jt = JH Jh + jj
#pragma unroll
  for (int jj = 0; jj < JH; ++jj)
#pragma unroll
    for (int s4 = 0; s4 < 4 * NT; ++s4) {
      const int i = 4 * s4 + kq, j = 16 * (JH * Jh + jj) + r16;
      af[jj][s4] = (s4 < S && i < N && j < N) ? shA[i * NP + j] * drop_factor(ka, (uint32_t)((b * N + i) * N + j)) : 0.f;
    }
  float* dWh = a.dWh[mod] ? a.dWh[mod] + (int64_t)b * N * HF + h * F : nullptr;
  for (int cidx = 0; cidx < nch; ++cidx) {
    load_chunk(cidx);                                       // (the last barrier above / below released the images)
    __syncthreads();
    gat_f32x4 zacc[JH];
#pragma unroll
    for (int jj = 0; jj < JH; ++jj) zacc[jj] = gat_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4 * NT; ++s4) {
      if (s4 < S) {
        const int i = 4 * s4 + kq;
        const float bf = shP[(i < N ? i : N) * FCS + 16 * Ft + r16];
#pragma unroll
        for (int jj = 0; jj < JH; ++jj) zacc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[jj][s4], bf, zacc[jj], 0, 0, 0);
      }
    }
    __syncthreads();                                        // both halves have read the strip: dZ may replace dP
#pragma unroll
    for (int jj = 0; jj < JH; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * (JH * Jh + jj) + 4 * kq + r;
        if (j < N && JH * Jh + jj < NT) shP[j * FCS + 16 * Ft + r16] = zacc[jj][r];
      }
    __syncthreads();
    // chunk epilogue: four rows per wave pass, 16 lanes x 4 columns per row
    const int col = FC * cidx + 4 * r16;
    const float4 al4 = *reinterpret_cast<const float4*>(a.gat_a[mod] + h * 2 * F + col);
    const float4 ar4 = *reinterpret_cast<const float4*>(a.gat_a[mod] + h * 2 * F + F + col);
    float4 pal = make_float4(0.f, 0.f, 0.f, 0.f), par = pal;
    for (int j = 4 * wave + kq; j < 4 * ((N + 3) / 4); j += 4 * NW) {
      const bool ok = j < N;
      const int jr = ok ? j : N;
      const float sj = ok ? shS[j] : 0.f, duj = ok ? shDu[j] : 0.f, dvj = ok ? shDv[j] : 0.f;
      const float4 z = *reinterpret_cast<const float4*>(shP + jr * FCS + 4 * r16), w = *reinterpret_cast<const float4*>(shW + jr * FCS + 4 * r16);
      const float4 o = make_float4(sj * z.x + duj * al4.x + dvj * ar4.x, sj * z.y + duj * al4.y + dvj * ar4.y,
                                   sj * z.z + duj * al4.z + dvj * ar4.z, sj * z.w + duj * al4.w + dvj * ar4.w);
      if (ok) {
        if (dWh) st4(reinterpret_cast<float4*>(dWh + (int64_t)j * HF + col), o);
        split_store4(a.dWh_p[mod], a.lo_off, (int64_t)b * N * HF + h * F + (int64_t)j * HF + col, o);
      }
      float dsp = ok ? z.x * w.x + z.y * w.y + z.z * w.z + z.w * w.w : 0.f;
      dsp += dpp_move<0xb1, 0xf>(0.f, dsp);
      dsp += dpp_move<0x4e, 0xf>(0.f, dsp);
      dsp += dpp_move<0x124, 0xf>(0.f, dsp);
      dsp += dpp_move<0x128, 0xf>(0.f, dsp);
      if (ok && r16 == 0) shDs[j] += dsp;                   // (row j belongs to this wave in every chunk)
      pal.x += duj * w.x; pal.y += duj * w.y; pal.z += duj * w.z; pal.w += duj * w.w;
      par.x += dvj * w.x; par.y += dvj * w.y; par.z += dvj * w.z; par.w += dvj * w.w;
    }
    // the four row groups of a wave hold the same columns: fold them (lanes l, l ^ 16, l ^ 32), then the waves through LDS
    float v8[8] = {pal.x, pal.y, pal.z, pal.w, par.x, par.y, par.z, par.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v8[e] += __shfl_xor(v8[e], 16);
      v8[e] += __shfl_xor(v8[e], 32);
    }
    if (kq == 0) {
      *reinterpret_cast<float4*>(shDa + wave * 2 * FC + 4 * r16) = make_float4(v8[0], v8[1], v8[2], v8[3]);
      *reinterpret_cast<float4*>(shDa + wave * 2 * FC + FC + 4 * r16) = make_float4(v8[4], v8[5], v8[6], v8[7]);
    }
    __syncthreads();                                        // partials complete; every wave is done with the chunk images
    for (int f = tid; f < 2 * FC; f += 64 * NW) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) acc += shDa[w * 2 * FC + f];
      const int half = f / FC, fc = f - half * FC;          // 0: d a_l, 1: d a_r
      a.da_part[mod][((int64_t)b * kHeads + h) * 2 * F + half * F + FC * cidx + fc] = acc;
    }
  }
  __syncthreads();
  if (tid < N) a.ds_part[mod][((int64_t)b * kHeads + h) * N + tid] = shDs[tid];
}

static size_t gat_bwd_chunked_lds(int N) {
  const size_t NP = N + 1, FCS = 64 + 4, NW = 8;
  return sizeof(float) * (2 * (N + 1) * FCS + 2 * (size_t)N * NP + 6 * N + 16 + NW * 2 * 64);
}

static size_t gat_bwd_lds(int N, int F) {
  const size_t NP = N + 1, FS = F + 4;
  const size_t NW = gat_nt(N) == 4 ? kAttnWavesWide : kAttnWavesSmall;
  const size_t WR = ((N + 1) * FS > 2 * NW * (size_t)F) ? (N + 1) * FS : 2 * NW * (size_t)F;
  return sizeof(float) * (WR + (N + 1) * FS + (3 * (size_t)N + 1) * NP + 5 * N + 16);
}
size_t gat_lds_bytes_max(int N, int F) {
  const size_t f = gat_fwd_lds(N, F), b = gat_bwd_lds(N, F);
  return f > b ? f : b;
}

template <int NT>
static int gat_attn_bwd_nt(hipStream_t st, const GatAttnBwdArgs& a) {
  constexpr int NW = NT == 4 ? kAttnWavesWide : kAttnWavesSmall;
  const size_t lds = gat_bwd_lds(a.N, a.D / 4);
  if (a.D / 16 <= 64) {                                     // one float4 slot per lane covers the head's F columns
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gat_attn_bwd<NT, NW, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DVGR_LAUNCH((k_gat_attn_bwd<NT, NW, 1>), dim3(a.B, kHeads, 2), dim3(64 * NW), lds, st, a);
    DVGR_LAUNCH_CHECK();
  }
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gat_attn_bwd<NT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DVGR_LAUNCH((k_gat_attn_bwd<NT, NW>), dim3(a.B, kHeads, 2), dim3(64 * NW), lds, st, a);
  DVGR_LAUNCH_CHECK();
}

template <int NT>
static int gat_attn_bwd_chunked_nt(hipStream_t st, const GatAttnBwdArgs& a) {
  const size_t lds = gat_bwd_chunked_lds(a.N);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gat_attn_bwd_chunked<NT, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DVGR_LAUNCH((k_gat_attn_bwd_chunked<NT, 8>), dim3(a.B, kHeads, 2), dim3(64 * 8), lds, st, a);
  DVGR_LAUNCH_CHECK();
}

int gat_attn_bwd(hipStream_t st, const GatAttnBwdArgs& a) {
  if (a.D / 4 > 512) return DVGR_E_DIMS;                    // the epilogue keeps two float4 column slots per lane
#ifdef DVGR_TIMING_AIDS
  static const bool no_chunked = []() { const char* e = getenv("DVGR_NO_CHUNKED_ATTN_BWD"); return e && e[0] == '1'; }();   // A/B timing aid
#else
  constexpr bool no_chunked = false;
#endif
  // N > 32: the whole-tile kernel's images allow ONE workgroup per CU; the column-chunked form fits two (80 KB each)
  if (!no_chunked && a.N > 32 && ((a.D / 4) % 64) == 0 && gat_bwd_chunked_lds(a.N) <= 80 * 1024)
    return gat_nt(a.N) == 3 ? gat_attn_bwd_chunked_nt<3>(st, a) : gat_attn_bwd_chunked_nt<4>(st, a);
  switch (gat_nt(a.N)) {
    case 1: return gat_attn_bwd_nt<1>(st, a);
    case 2: return gat_attn_bwd_nt<2>(st, a);
    case 3: return gat_attn_bwd_nt<3>(st, a);
    default: return gat_attn_bwd_nt<4>(st, a);
  }
}

// =================================================================================================
// S1 backward.  grid (B, 2).  SURVEY.md Appendix A.2
// =================================================================================================
__global__ __launch_bounds__(256) void k_punish_bwd(const PunishBwdArgs a) {
  __shared__ float shT[64];
  const int b = blockIdx.x, mod = blockIdx.y, tid = threadIdx.x;
  const int N = a.N, D = a.D;
  if (tid < N) {
    float ds = 0.f;
    const int copies = a.copies > 1 ? a.copies : 1;
    for (int c = 0; c < copies; ++c)
      for (int h = 0; h < kHeads; ++h) ds += a.ds_part[mod][(((int64_t)c * a.B + b) * kHeads + h) * N + tid];
    const float s = a.s[mod][b * N + tid];
    shT[tid] = ds * s * (1.f - s);
  }
  __syncthreads();
  const int d = blockIdx.z * 256 + tid;                     // one feature column per thread
  if (d >= D) return;
  const float* X = a.X[mod] + (int64_t)b * N * D;
  float* dX = a.dX[mod] + (int64_t)b * N * D;
  const int parts = a.parts ? a.parts : 3;
  if (a.zero && (parts & 1) && mod == 0 && blockIdx.z == 0)
    for (int w = tid; w < a.zero_cols; w += 256) a.zero[(int64_t)b * a.zero_cols + w] = 0.f;
  if (parts == 1) {                                         // dp alone: the query branch can start before dX is final
    float acc = 0.f;
#pragma unroll 4
    for (int n = 0; n < N; ++n) acc += shT[n] * X[(int64_t)n * D + d];
    a.dp_cat[(int64_t)b * 2 * D + mod * D + d] = acc;
    return;
  }
  const float pd = a.p[mod][(int64_t)b * D + d];
  const float* add = a.add[mod] ? a.add[mod] + (int64_t)b * N * D : nullptr;
  float acc = 0.f;
#pragma unroll 4
  for (int n = 0; n < N; ++n) {
    const float g = dX[(int64_t)n * D + d];
    const float e = add ? add[(int64_t)n * D + d] : 0.f;
    if (parts & 1) acc += shT[n] * X[(int64_t)n * D + d];
    dX[(int64_t)n * D + d] = (g + e) + shT[n] * pd;
  }
  if (parts & 1) a.dp_cat[(int64_t)b * 2 * D + mod * D + d] = acc;
}
int punish_bwd(hipStream_t st, const PunishBwdArgs& a) {
  DVGR_LAUNCH_FORKABLE(k_punish_bwd, dim3(a.B, 2, (a.D + 255) / 256), dim3(256), 0, st, a);
  DVGR_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void k_dz_build(const DzBuildArgs a) {
  const int mod = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, n4 = a.n / 4;     // float4 index
  if (i >= n4) return;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 c = a.g_c[mod] ? reinterpret_cast<const float4*>(a.g_c[mod])[i] : z;
  float4 s = a.g_s[mod] ? reinterpret_cast<const float4*>(a.g_s[mod])[i] : z;
  if (a.in0[mod]) { const float4 e = reinterpret_cast<const float4*>(a.in0[mod])[i]; s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w; }
  if (a.in1[mod]) { const float4 e = reinterpret_cast<const float4*>(a.in1[mod])[i]; s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w; }
  reinterpret_cast<float4*>(a.dz[mod])[i] = c;
  reinterpret_cast<float4*>(a.dz[mod])[n4 + i] = s;
}
int dz_build(hipStream_t st, const DzBuildArgs& a) {
  const int64_t n4 = a.n / 4;
  DVGR_LAUNCH(k_dz_build, dim3((unsigned)((n4 + 255) / 256), 2), dim3(256), 0, st, a);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// S0 backward rows.  grid (B), loops over layers.  SURVEY.md Appendix A.1
// =================================================================================================
constexpr int kQaChunks = 4;   // workgroups per batch element: each redoes the (tiny) softmax part, then owns 1/4 of the rows
// pass 1 of the S0 backward: one wave per (layer, sample, word) row: s1 = words_t . dqbar, s2 = d_t . w2
__global__ __launch_bounds__(256) void k_qa_bwd_rows(const QaBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // (l * B + b) * T + t
  const int T = a.T, D = a.D, Wd = a.Wd;
  if (row >= (int64_t)a.L * a.B * T) return;
  const int l = (int)(row / ((int64_t)a.B * T));
  const int bt = (int)(row - (int64_t)l * a.B * T), b = bt / T;
  const float* sav = a.saved + (int64_t)l * a.s_layer_stride;
  const float4* wrow = reinterpret_cast<const float4*>(a.words + (int64_t)bt * Wd);
  const float4* dq4 = reinterpret_cast<const float4*>(a.dqbar + ((int64_t)l * a.B + b) * Wd);
  const float4* drow = reinterpret_cast<const float4*>(sav + a.s_d + (int64_t)bt * D);
  const float4* w24 = reinterpret_cast<const float4*>(a.params + (int64_t)l * a.layer_stride + a.off_w2);
  float s1 = 0.f, s2 = 0.f;
  for (int w = lane; w < Wd / 4; w += 64) {
    const float4 x = wrow[w], y = dq4[w];
    s1 += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
  }
#pragma unroll 4
  for (int k = lane; k < D / 4; k += 64) {
    const float4 x = drow[k], y = w24[k];
    s2 += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { a.rowdots[row] = s1; a.rowdots[(int64_t)a.L * a.B * T + row] = s2; }
}

__global__ __launch_bounds__(256) void k_qa_bwd(const QaBwdArgs a) {
  __shared__ float sh_dah[256], sh_c[256], sh_ah[256], sh_a[256], sh_da[256], sh_dr[256];
  const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const int T = a.T, D = a.D, Wd = a.Wd;
  const int len = (int)a.qlen[b];
  for (int l = 0; l < a.L; ++l) {
    const float* sav = a.saved + (int64_t)l * a.s_layer_stride;
    const float* w2 = a.params + (int64_t)l * a.layer_stride + a.off_w2;
    const float* dqb = a.dqbar + ((int64_t)l * a.B + b) * Wd;
    const float* dsv = sav + a.s_d + (int64_t)b * T * D;
    if (tid < T) {                                          // row dots from k_qa_bwd_rows
      sh_dah[tid] = a.rowdots[((int64_t)l * a.B + b) * T + tid];
      sh_c[tid] = a.rowdots[((int64_t)(a.L + l) * a.B + b) * T + tid];
    }
    if (tid < T) { sh_ah[tid] = sav[a.s_ah + b * T + tid]; sh_a[tid] = sav[a.s_a + b * T + tid]; }
    __syncthreads();
    float dot = 0.f;
    for (int t = 0; t < T; ++t) dot += sh_dah[t] * sh_ah[t];
    const float S = sav[a.s_S + b];
    if (tid < T) sh_da[tid] = (tid < len) ? (sh_dah[tid] - dot) / S : 0.f;
    __syncthreads();
    float adot = 0.f;
    for (int t = 0; t < T; ++t) adot += sh_a[t] * sh_da[t];
    if (tid < T) {
      const float dr = sh_a[tid] * (sh_da[tid] - adot);
      sh_dr[tid] = dr;
      if (ch == 0) a.dr[(int64_t)l * a.B * T + b * T + tid] = dr;
    }
    __syncthreads();
    // rows t = ch, ch + kQaChunks, ... of this batch element, four at a time: all loads of a group of rows are issued
    // before the group's first store, so the rows' memory round trips overlap instead of forming a chain (loads and
    // stores share one in-order counter; 20 -> 15 us in the step, profiles/experiments_r03.tsv)
    float* dy = a.dy + ((int64_t)l * a.B + b) * T * D;
    const bool fresh = a.first && l == 0;
    for (int tb = ch; tb < T; tb += 4 * kQaChunks) {
      for (int w = tid; w < Wd; w += 256) {                 // dwords_t (+)= ah_t * dqbar
        const float dq = dqb[w];
        float oldv[4] = {0.f, 0.f, 0.f, 0.f};
        if (!fresh) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = tb + r * kQaChunks;
            if (t < T) oldv[r] = a.gwords[((int64_t)b * T + t) * Wd + w];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = tb + r * kQaChunks;
          if (t < T) a.gwords[((int64_t)b * T + t) * Wd + w] = oldv[r] + sh_ah[t] * dq;
        }
      }
      for (int k = tid; k < D / 4; k += 256) {              // dy = (dd - d (d.dd)) / nrm, dd = dr * w2
        const float4 w = reinterpret_cast<const float4*>(w2)[k];
        float4 dv[4];
        float sc[4], ct[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = tb + r * kQaChunks;
          if (t < T) {
            dv[r] = reinterpret_cast<const float4*>(dsv + (int64_t)t * D)[k];
            sc[r] = sh_dr[t] / sav[a.s_nrm + b * T + t];
            ct[r] = sh_c[t];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = tb + r * kQaChunks;
          if (t < T) {
            const float4 o = make_float4(sc[r] * (w.x - dv[r].x * ct[r]), sc[r] * (w.y - dv[r].y * ct[r]),
                                         sc[r] * (w.z - dv[r].z * ct[r]), sc[r] * (w.w - dv[r].w * ct[r]));
            st4(reinterpret_cast<float4*>(dy + (int64_t)t * D) + k, o);
            split_store4(a.dy_p, a.lo_off, (((int64_t)l * a.B + b) * T + t) * D + 4 * k, o);
          }
        }
      }
    }
    __syncthreads();
  }
}
int qa_bwd(hipStream_t st, const QaBwdArgs& a) {
  const int64_t rows = (int64_t)a.L * a.B * a.T;
  DVGR_LAUNCH(k_qa_bwd_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
  DVGR_LAUNCH(k_qa_bwd, dim3(a.B, kQaChunks), dim3(256), 0, st, a);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// Column sums: grid (ceil(cols / 64), jobs), 1024 threads = 16 waves; lane -> column, wave -> rows.
// =================================================================================================
// 256 threads = 4 waves; a lane owns 4 adjacent columns (16-byte loads, 1 KB per row per wave), a workgroup 256 columns
// of one row chunk; waves interleave rows, four loads in flight per lane.  Columns need not be a multiple of 4 (the
// tail lanes fall back to scalar loads); ld and the base pointers are 16-byte aligned for every caller.
__device__ __forceinline__ float4 colsum_load4(const float* row, int c, int cols) {
  if (c + 4 <= cols && ((reinterpret_cast<uintptr_t>(row + c) & 15) == 0)) return *reinterpret_cast<const float4*>(row + c);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols) v.x = row[c];
  if (c + 1 < cols) v.y = row[c + 1];
  if (c + 2 < cols) v.z = row[c + 2];
  if (c + 3 < cols) v.w = row[c + 3];
  return v;
}
__device__ __forceinline__ void colsum_segment(const float* M, const float* w, int rows, int64_t ld, int cols, int c,
                                               int ch, int nch, int wave, float4& acc) {
  const int per = (rows + nch - 1) / nch;
  const int b0 = ch * per, e0 = (b0 + per) < rows ? (b0 + per) : rows;
  int r = b0 + wave;
  for (; r + 12 < e0; r += 16) {
    const float4 v0 = colsum_load4(M + (int64_t)r * ld, c, cols), v1 = colsum_load4(M + (int64_t)(r + 4) * ld, c, cols);
    const float4 v2 = colsum_load4(M + (int64_t)(r + 8) * ld, c, cols), v3 = colsum_load4(M + (int64_t)(r + 12) * ld, c, cols);
    const float w0 = w ? w[r] : 1.f, w1 = w ? w[r + 4] : 1.f, w2 = w ? w[r + 8] : 1.f, w3 = w ? w[r + 12] : 1.f;
    acc.x += (w0 * v0.x + w1 * v1.x) + (w2 * v2.x + w3 * v3.x);
    acc.y += (w0 * v0.y + w1 * v1.y) + (w2 * v2.y + w3 * v3.y);
    acc.z += (w0 * v0.z + w1 * v1.z) + (w2 * v2.z + w3 * v3.z);
    acc.w += (w0 * v0.w + w1 * v1.w) + (w2 * v2.w + w3 * v3.w);
  }
  for (; r < e0; r += 4) {
    const float4 v = colsum_load4(M + (int64_t)r * ld, c, cols);
    const float wr = w ? w[r] : 1.f;
    acc.x += wr * v.x; acc.y += wr * v.y; acc.z += wr * v.z; acc.w += wr * v.w;
  }
}
__global__ __launch_bounds__(256) void k_colsum(const ColsumBatch b) {
  __shared__ float4 red[4][64];
  const ColsumJob& J = b.j[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x * 256 >= J.cols) return;                   // uniform per workgroup
  const int c = blockIdx.x * 256 + 4 * lane;
  const int nch = gridDim.z, ch = blockIdx.z;               // row chunk of this workgroup
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < J.cols) {
    colsum_segment(J.M0, J.w0, J.rows0, J.ld, J.cols, c, ch, nch, wave, acc);
    if (J.M1) colsum_segment(J.M1, J.w1, J.rows1, J.ld, J.cols, c, ch, nch, wave, acc);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < J.cols) {
    const float4 a0 = red[0][lane], a1 = red[1][lane], a2 = red[2][lane], a3 = red[3][lane];
    const float s[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                        (a0.w + a1.w) + (a2.w + a3.w)};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (c + q < J.cols) { if (nch > 1) atomicAdd(J.out + c + q, s[q]); else J.out[c + q] = s[q]; }
  }
}
int colsum(hipStream_t st, const ColsumBatch& b) {
  if (b.n < 1 || b.n > kMaxColsumJobs) return DVGR_E_DIMS;
  int maxc = 0;
  for (int i = 0; i < b.n; ++i) maxc = b.j[i].cols > maxc ? b.j[i].cols : maxc;
  const int chunks = b.chunks > 1 ? b.chunks : 1;
  DVGR_LAUNCH(k_colsum, dim3((maxc + 255) / 256, b.n, chunks), dim3(256), 0, st, b);
  DVGR_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void k_drop_x(const float* X0, const float* X1, float* out0, float* out1,
                                                uint32_t k00, uint32_t k01, uint32_t k10, uint32_t k11, int64_t n4,
                                                uint16_t* p0, uint16_t* p1, int64_t lo_off) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index
  if (i >= n4) return;
  const int mod = blockIdx.y;
  const float4 x = reinterpret_cast<const float4*>(mod ? X1 : X0)[i];
  float* out = mod ? out1 : out0;
  const uint32_t keys[2] = {mod ? k10 : k00, mod ? k11 : k01};
  const uint32_t e = (uint32_t)(i * 4);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const uint32_t h0 = mix32(((e >> 1) * 0x9E3779B1u) ^ keys[g]);
    const uint32_t h1 = mix32((((e >> 1) + 1u) * 0x9E3779B1u) ^ keys[g]);
    float4 y;
    y.x = x.x * (((h0 & 0xFFFFu) >= kDropThresh16) ? kKeepScale : 0.f);
    y.y = x.y * (((h0 >> 16) >= kDropThresh16) ? kKeepScale : 0.f);
    y.z = x.z * (((h1 & 0xFFFFu) >= kDropThresh16) ? kKeepScale : 0.f);
    y.w = x.w * (((h1 >> 16) >= kDropThresh16) ? kKeepScale : 0.f);
    reinterpret_cast<float4*>(out + (int64_t)g * n4 * 4)[i] = y;
    split_store4(mod ? p1 : p0, lo_off, (int64_t)g * n4 * 4 + 4 * i, y);
  }
}
int drop_x(hipStream_t st, const float* X0, const float* X1, float* out0, float* out1, const uint32_t (&key)[2][2], int64_t n,
           uint16_t* p0, uint16_t* p1, int64_t lo_off) {
  const int64_t n4 = n / 4;
  DVGR_LAUNCH(k_drop_x, dim3((unsigned)((n4 + 255) / 256), 2), dim3(256), 0, st, X0, X1, out0, out1,
                     key[0][0], key[0][1], key[1][0], key[1][1], n4, p0, p1, lo_off);
  DVGR_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void k_split_planes(const SplitBatch b) {
  if (b.zero3 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 3) b.zero3[threadIdx.x] = 0.f;   // (the forward's loss scalars)
  const SplitJob& J = b.j[blockIdx.y];
  const int64_t n4 = J.n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(J.src)[i];
    split_store4(J.hi, b.lo_off, 4 * i, x);
    if (J.drop_hi[0]) {                                      // uniform per job
      const uint32_t e = (uint32_t)(i * 4);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t h0 = mix32(((e >> 1) * 0x9E3779B1u) ^ J.drop_key[g]);
        const uint32_t h1 = mix32((((e >> 1) + 1u) * 0x9E3779B1u) ^ J.drop_key[g]);
        float4 y;
        y.x = x.x * (((h0 & 0xFFFFu) >= kDropThresh16) ? kKeepScale : 0.f);
        y.y = x.y * (((h0 >> 16) >= kDropThresh16) ? kKeepScale : 0.f);
        y.z = x.z * (((h1 & 0xFFFFu) >= kDropThresh16) ? kKeepScale : 0.f);
        y.w = x.w * (((h1 >> 16) >= kDropThresh16) ? kKeepScale : 0.f);
        split_store4(J.drop_hi[g], b.lo_off, 4 * i, y);
      }
    }
  }
}
int split_planes(hipStream_t st, const SplitBatch& b) {
  if (b.n < 1 || b.n > 4) return DVGR_E_DIMS;
  int64_t mx = 0;
  for (int i = 0; i < b.n; ++i) mx = b.j[i].n > mx ? b.j[i].n : mx;
  const int64_t blocks = (mx / 4 + 255) / 256;
  DVGR_LAUNCH_FORKABLE(k_split_planes, dim3((unsigned)(blocks > 2048 ? 2048 : blocks), b.n), dim3(256), 0, st, b);
  DVGR_LAUNCH_CHECK();
}

// =================================================================================================
// bf16 shadow copy
// =================================================================================================
__global__ __launch_bounds__(256) void k_to_bf16(const float* src, uint16_t* dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t u = __float_as_uint(src[i]);
  uint32_t r;
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) r = (u >> 16) | 0x40u;          // keep NaN a NaN
  else r = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;                       // round to nearest even
  dst[i] = (uint16_t)r;
}
int to_bf16(hipStream_t st, const float* src, void* dst, int64_t n) {
  DVGR_LAUNCH(k_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, (uint16_t*)dst, n);
  DVGR_LAUNCH_CHECK();
}

}  // namespace dvgr

// test hook: materialise the keep decisions of one dropout site
namespace dvgr {
__global__ __launch_bounds__(256) void k_keep_mask(uint32_t key, int64_t n, float* keep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = drop_keep(key, (uint32_t)i) ? 1.0f : 0.0f;
}
int keep_mask_launch(hipStream_t st, uint32_t key, int64_t n, float* keep) {
  DVGR_LAUNCH(k_keep_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, key, n, keep);
  return (int)hipGetLastError();
}
// ---- zero a list of ranges -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zero_ranges(const ZeroList z) {
  float* p = z.p[blockIdx.y];
  const int64_t n = z.n[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
      reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
  }
}

// The head of the backward as ONE launch: the zero list (k_zero_ranges) and the ELU derivative of the MFB output
// (k_elu_bwd) -- two 6 us launches on an idle chip whose cost was their two boundaries.  Blocks [0, nb_elu) take one float4 of
// the element-wise job each, the rest walk the zero list as k_zero_ranges does (gx blocks per range).
__global__ __launch_bounds__(256) void k_bwd_head(const ZeroList z, int gx, int nb_elu, const float* g, const float* y, float* out,
                                                  int64_t n4, uint16_t* op, int64_t lo_off) {
  if ((int)blockIdx.x < nb_elu) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index
    if (i >= n4) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g) {
      const float4 gg = reinterpret_cast<const float4*>(g)[i], yy = reinterpret_cast<const float4*>(y)[i];
      v = make_float4(gg.x * elu_grad_from_out(yy.x), gg.y * elu_grad_from_out(yy.y), gg.z * elu_grad_from_out(yy.z),
                      gg.w * elu_grad_from_out(yy.w));
    }
    reinterpret_cast<float4*>(out)[i] = v;
    split_store4(op, lo_off, 4 * i, v);
    return;
  }
  const int b = (int)blockIdx.x - nb_elu, seg = b / gx, bx = b - seg * gx;
  float* p = z.p[seg];
  const int64_t n = z.n[seg];
  const int64_t stride = (int64_t)gx * 256;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const int64_t n4z = n >> 2;
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < n4z; i += stride)
      reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (n4z << 2) + (int64_t)bx * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
  } else {
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
  }
}

int bwd_head(hipStream_t st, const ZeroList& z, const float* g, const float* y, float* out, int64_t n, uint16_t* op, int64_t lo_off) {
  if (z.count > kMaxZeroSegs || (n & 3)) return DVGR_E_DIMS;
  int64_t longest = 0;
  for (int i = 0; i < z.count; ++i) longest = z.n[i] > longest ? z.n[i] : longest;
  int gx = (int)((longest / 4 + 255) / 256);
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  const int64_t n4 = n / 4;
  const int nb_elu = (int)((n4 + 255) / 256);
  DVGR_LAUNCH(k_bwd_head, dim3((unsigned)(nb_elu + gx * (z.count > 0 ? z.count : 0))), dim3(256), 0, st, z, gx, nb_elu, g, y, out, n4, op, lo_off);
  return (int)hipGetLastError();
}

int zero_ranges(hipStream_t st, const ZeroList& z) {
  if (z.count <= 0) return 0;
  if (z.count > kMaxZeroSegs) return DVGR_E_DIMS;
  int64_t longest = 0;
  for (int i = 0; i < z.count; ++i) longest = z.n[i] > longest ? z.n[i] : longest;
  int gx = (int)((longest / 4 + 255) / 256);
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  DVGR_LAUNCH(k_zero_ranges, dim3(gx, z.count), dim3(256), 0, st, z);
  return (int)hipGetLastError();
}

}  // namespace dvgr

#ifdef DVGR_CLK
extern "C" int dvgr_dbg_clocks(long long* host_out, int n) {
  if (n > 4096 * 8) n = 4096 * 8;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dvgr::g_clk), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif
