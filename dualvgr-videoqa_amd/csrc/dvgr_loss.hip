// dvgr_loss.hip -- fused auxiliary losses of the unit's list outputs, value AND gradient in one pass
// (SURVEY.md section 8f, row f1).  Restates utils.py:10-31 + train.py:148-154 of the reference:
//   common_l = mean_{b,i,j} (C(com_app_l) - C(com_mot_l))^2,   C(E) = Ehat Ehat^T, Ehat = rows of (E - mean over
//              nodes) L2-normalised over D (eps 1e-12)
//   dep_l    = sum_b [ tr(R K(aq_l) R K(com_app_l)) + tr(R K(mq_l) R K(com_mot_l)) ],  K(E) = E E^T, R = I - 1/N
//   aux      = alpha * mean_l common_l + beta * dep_scale * mean_l dep_l
// Algebra used: with c = R E (rows centred over the nodes), R K R = c c^T =: G(E) and tr(R K1 R K2) = <G1, G2>;
// d<G1,G2>/dE1 = 2 G2 c1.  The normalised Gram is C_ij = G_ij / (n_i n_j), n_i = max(sqrt(G_ii), 1e-12), and
// (ehat_i . dehat_i) = sum_j M_ij C_ij for dehat = M ehat, so every N x N quantity comes from the four centred
// Grams of a (b, l) pair and the D-wide work is two passes over the rows: Gram, then gradient.
#include "dvgr_common.h"
#include "../../include/dvgr.h"

#define DVGR_TRY_HIP(x) do { const int rc_ = (x); if (rc_ != 0) return rc_; } while (0)

namespace dvgr {

struct AuxArgs {
  const float* t[4];     // com_app, com_mot, aq_fusion, mq_fusion: (L, B, N, D)
  float* g[4];           // gradients of aux w.r.t. the same tensors
  float* gram;           // [L][B][4][N][N] centred Grams
  float* wt;             // [L][B][4][N][NBP] gradient matrices: g_t[i] = sum_j wt_t[j][i] c_t[j]
  float* loss;           // [3]: sum(vis), mean_l common_l, mean_l dep_l
  float wc, wh;          // alpha / (L B N^2), beta * dep_scale / L
  int B, N, D, L, NBP;   // L = the list entries THIS call works on (a group of them: aux_losses' first / count)
  int Ltot;              // all list entries of the step: the means are over these
};

// ---- pass 1: centred Gram of one tensor of one (b, l) ---------------------------------------------------
// c c^T on the fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products).  The four waves split D; a wave walks its columns
// in chunks of 64: the chunk of all N rows is loaded ONCE with whole 256-byte row pieces per instruction (16 lanes x 16 B per
// row, four rows per instruction), centred on the way (the column means of a chunk come from the loaded values themselves:
// in-lane over the row groups, then across the four lanes that hold a column) and staged in the wave's own LDS tile, from which
// the MFMA fragments are read (lane (r, q): row 16 I + r, columns 4 q .. 4 q + 3 of a 16-column k-step; the same fragment is
// the A operand of row block I and the B operand of column block J >= I).  The waves' partial tiles are summed through LDS.
// (Until round 3 the fragments came straight from global memory -- 16 rows x 64 bytes per instruction -- behind a separate
// pass over the rows for the means: 356 us at N = 64, 1.7 TB/s.)
typedef float floatx4 __attribute__((ext_vector_type(4)));
// chunk width CW: 64 columns at N > 32; 32 at N <= 32, where the kernel runs beside the forward's last contractions and its
// LDS (4 waves x 16 NT rows x (CW + 4) floats + the partial tiles) has to fit next to their 128 KB on a compute unit
template <int NT> struct GramCfg { static constexpr int CW = NT >= 3 ? 64 : 32, STR = CW + 4; };

template <int NT>
__global__ __launch_bounds__(256) void k_aux_gram(const AuxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, l = blockIdx.y >> 2, t = blockIdx.y & 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, D = a.D;
  constexpr int NR = 16 * NT;                         // staged rows (rows >= N are zero)
  constexpr int CW = GramCfg<NT>::CW, STR = GramCfg<NT>::STR, LPR = CW / 4, RPI = 64 / LPR;   // lanes per row piece, rows per instruction
  float* stage = sm + wave * NR * STR;                // [NR][CW + 4] per wave
  float* red = sm + 4 * NR * STR;                     // [NT (NT + 1) / 2][256]
  const float* E = a.t[t] + ((int64_t)l * a.B + b) * N * D;
  const int r16 = lane & 15, kq = lane >> 4;
  floatx4 acc[NT][NT];
#pragma unroll
  for (int I = 0; I < NT; ++I)
#pragma unroll
    for (int J = 0; J < NT; ++J) acc[I][J] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int kw = D / 4;                               // D % 256 == 0 or the tail chunk is narrower: handled by `cols`
  const float inv_n = 1.0f / (float)N;
  const int lc = lane % LPR, lr = lane / LPR;          // load lane: columns 4 lc .. 4 lc + 3 of the rows lr, lr + RPI, ...
  for (int c0 = wave * kw; c0 < (wave + 1) * kw; c0 += CW) {
    const int cols = ((wave + 1) * kw - c0) < CW ? ((wave + 1) * kw - c0) : CW;      // multiple of 16 (D % 64 == 0)
    const bool cin = 4 * lc < cols;
    float4 v[NR / RPI];
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NR / RPI; ++i) {
      const int row = RPI * i + lr;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cin && row < N) v[i] = *reinterpret_cast<const float4*>(E + (int64_t)row * D + c0 + 4 * lc);
      sum.x += v[i].x; sum.y += v[i].y; sum.z += v[i].z; sum.w += v[i].w;
    }
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) {              // over the lanes that hold the same columns
      sum.x += __shfl_xor(sum.x, m); sum.y += __shfl_xor(sum.y, m); sum.z += __shfl_xor(sum.z, m); sum.w += __shfl_xor(sum.w, m);
    }
    sum.x *= inv_n; sum.y *= inv_n; sum.z *= inv_n; sum.w *= inv_n;
#pragma unroll
    for (int i = 0; i < NR / RPI; ++i) {
      const int row = RPI * i + lr;
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cin && row < N) c = make_float4(v[i].x - sum.x, v[i].y - sum.y, v[i].z - sum.z, v[i].w - sum.w);
      *reinterpret_cast<float4*>(stage + row * STR + 4 * lc) = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the tile is wave-private: LDS ops of one wave execute in order
    for (int ks = 0; ks < cols / 16; ++ks) {
      float4 f[NT];
#pragma unroll
      for (int I = 0; I < NT; ++I) f[I] = *reinterpret_cast<const float4*>(stage + (16 * I + r16) * STR + 16 * ks + 4 * kq);
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I; J < NT; ++J) {
          acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[I].x, f[J].x, acc[I][J], 0, 0, 0);
          acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[I].y, f[J].y, acc[I][J], 0, 0, 0);
          acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[I].z, f[J].z, acc[I][J], 0, 0, 0);
          acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[I].w, f[J].w, acc[I][J], 0, 0, 0);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  // the four waves add their partial tiles into one LDS image in a fixed order (deterministic)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      int tix = 0;
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I; J < NT; ++J, ++tix)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = red + tix * 256 + r * 64 + lane;
            *p = (w == 0) ? acc[I][J][r] : *p + acc[I][J][r];
          }
    }
    __syncthreads();
  }
  float* G = a.gram + (((int64_t)l * a.B + b) * 4 + t) * N * N;
  const int r = tid >> 6;
  {
    int tix = 0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = I; J < NT; ++J, ++tix) {
        const float v = red[tix * 256 + tid];
        const int row = 16 * I + 4 * (lane >> 4) + r, col = 16 * J + (lane & 15);
        if (row < N && col < N) { G[row * N + col] = v; G[col * N + row] = v; }
      }
  }
}

// The same pass with the fragments read straight from global memory (16 rows x 64 bytes per instruction) behind a separate
// pass over the rows for the column means: 13 KB of LDS instead of 21.  Kept for N <= 32, where it runs beside the forward's
// last contractions and measures 0.6 % of a C2 step FASTER than the staged form (aux_gram_staged_c2); at N = 64 the staged form
// is worth 1.2 % of a C5 step (aux_gram_staged_c5).
template <int NT>
__global__ __launch_bounds__(256) void k_aux_gram_direct(const AuxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, l = blockIdx.y >> 2, t = blockIdx.y & 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, D = a.D;
  float* mu = sm;                                     // [D]
  float* red = sm + D;                                // [NT (NT + 1) / 2][256]
  const float* E = a.t[t] + ((int64_t)l * a.B + b) * N * D;
  for (int d4 = tid; d4 < D / 4; d4 += 256) {           // column means, 4 rows in flight per thread
    const float4* col = reinterpret_cast<const float4*>(E) + d4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + 4 <= N; i += 4) {
      const float4 v0 = col[(int64_t)i * (D / 4)], v1 = col[(int64_t)(i + 1) * (D / 4)];
      const float4 v2 = col[(int64_t)(i + 2) * (D / 4)], v3 = col[(int64_t)(i + 3) * (D / 4)];
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; i < N; ++i) {
      const float4 v = col[(int64_t)i * (D / 4)];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = 1.0f / (float)N;
    *reinterpret_cast<float4*>(mu + 4 * d4) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
  __syncthreads();
  const int r16 = lane & 15, kq = lane >> 4;
  floatx4 acc[NT][NT];
#pragma unroll
  for (int I = 0; I < NT; ++I)
#pragma unroll
    for (int J = 0; J < NT; ++J) acc[I][J] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int kw = D / 4;                               // D % 64 == 0: every wave owns a multiple of 16 columns
  const int nk = kw / 16;
  // two operand sets: the loads of step it + 1 are in flight while the MFMAs of step it issue
  auto load = [&](int it, float (&f)[NT][4]) {
    const int k0 = wave * kw + it * 16;
    const bool in = it < nk;
    const float4 m = in ? *reinterpret_cast<const float4*>(mu + k0 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int I = 0; I < NT; ++I) {
      const int row = 16 * I + r16;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in && row < N) {
        v = *reinterpret_cast<const float4*>(E + (int64_t)row * D + k0 + 4 * kq);
        v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
      }
      f[I][0] = v.x; f[I][1] = v.y; f[I][2] = v.z; f[I][3] = v.w;
    }
  };
  auto mma = [&](const float (&f)[NT][4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I; J < NT; ++J) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[I][e], f[J][e], acc[I][J], 0, 0, 0);
  };
  float f0[NT][4], f1[NT][4];
  load(0, f0);
  for (int it = 0; it < nk; it += 2) {
    load(it + 1, f1);
    mma(f0);
    load(it + 2, f0);
    mma(f1);
  }
  // the four waves add their partial tiles into one LDS image in a fixed order (deterministic; 1/4 of the LDS of four
  // separate images, which matters for occupancy at N = 64)
  constexpr int NTP = NT * (NT + 1) / 2;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      int tix = 0;
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I; J < NT; ++J, ++tix)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = red + tix * 256 + r * 64 + lane;
            *p = (w == 0) ? acc[I][J][r] : *p + acc[I][J][r];
          }
    }
    __syncthreads();
  }
  float* G = a.gram + (((int64_t)l * a.B + b) * 4 + t) * N * N;
  const int r = tid >> 6;
  {
    int tix = 0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = I; J < NT; ++J, ++tix) {
        const float v = red[tix * 256 + tid];
        const int row = 16 * I + 4 * (lane >> 4) + r, col = 16 * J + (lane & 15);
        if (row < N && col < N) { G[row * N + col] = v; G[col * N + row] = v; }
      }
  }
  (void)NTP;
}

// ---- pass 2: N x N algebra of one (b, l) ------------------------------------------------------------------------
// Everything downstream of the Grams is linear in the centred rows c_t, so the whole gradient of tensor t collapses into
// ONE N x N matrix:  g_t[i] = sum_j W_t[j][i] c_t[j]  with
//   W_t[j][i] = Mh_t[j][i]                                   HSIC:    2 wh G_partner              (all four tensors)
//             + Mc_t[j][i] / (n_j n_i) - delta_ij s_i/n_i^2  common:  through the row normalisation (t = com_app, com_mot)
//             - (1/N) sum_i' [the common part of row j]      ... and through the centring over the nodes
// where Mc = +-4 wc (C1 - C2), s_i = sum_j Mc_ij C_ij.  Rows are stored padded to NBP floats (32-byte aligned) so that
// pass 3 can fetch them with scalar loads.
__global__ __launch_bounds__(256) void k_aux_combine(const AuxArgs a) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, l = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, NN = N * N, NBP = a.NBP;
  const int64_t bl = (int64_t)l * a.B + b;
  float* G = sm;                                      // [4][NN]
  float* A1 = G + 4 * NN;                             // [NN] common part of W for com_app, then com_mot
  float* A2 = A1 + NN;
  float* inv = A2 + NN;                               // [2][N] 1 / n_i
  float* srow = inv + 2 * N;                          // [2][N]
  float* rmean = srow + 2 * N;                        // [2][N]
  float* red = rmean + 2 * N;                         // [8]
  for (int e = tid; e < 4 * NN; e += 256) G[e] = a.gram[bl * 4 * NN + e];
  __syncthreads();
  if (tid < 2 * N) {
    const int t = tid / N, i = tid - t * N;
    inv[tid] = 1.0f / fmaxf(sqrtf(fmaxf(G[t * NN + i * N + i], 0.f)), 1e-12f);
  }
  __syncthreads();
  float pc = 0.f, ph = 0.f;
  for (int e = tid; e < NN; e += 256) {
    const int i = e / N, j = e - i * N;
    const float c1 = G[e] * (inv[i] * inv[j]), c2 = G[NN + e] * (inv[N + i] * inv[N + j]);
    const float diff = c1 - c2;
    pc += diff * diff;
    ph += G[2 * NN + e] * G[e] + G[3 * NN + e] * G[NN + e];
    const float m = 4.f * a.wc * diff;                // dehat1 = 2 Gc ehat1 with Gc = 2 wc diff;  dehat2 = -(...)
    A1[e] = m;                                        // Mc of com_app (com_mot: -m); C recomputed below
  }
  pc = wave_sum(pc); ph = wave_sum(ph);
  if (lane == 0) { red[wave] = pc; red[4 + wave] = ph; }
  __syncthreads();
  if (tid < 2 * N) {                                  // s_i = sum_j Mc_ij C_ij
    const int t = tid / N, i = tid - t * N;
    float s = 0.f;
    for (int j = 0; j < N; ++j) {                     // (G and Mc are symmetric: [j][i] puts consecutive threads on consecutive banks --
      const float c = G[t * NN + j * N + i] * (inv[t * N + i] * inv[t * N + j]);   //  [i][j] is a stride of N floats, a 32-way conflict at N = 64)
      s += (t ? -A1[j * N + i] : A1[j * N + i]) * c;
    }
    srow[tid] = s;
  }
  __syncthreads();
  for (int e = tid; e < NN; e += 256) {               // common part: Mc_ji / (n_j n_i) - delta_ij s_i / n_i^2
    const int j = e / N, i = e - j * N;
    const float m = A1[e];
    A2[e] = -m * (inv[N + j] * inv[N + i]) - (i == j ? srow[N + i] * inv[N + i] * inv[N + i] : 0.f);
    A1[e] = m * (inv[j] * inv[i]) - (i == j ? srow[i] * inv[i] * inv[i] : 0.f);
  }
  __syncthreads();
  if (tid < 2 * N) {
    const int t = tid / N, j = tid - t * N;
    const float* A = t ? A2 : A1;
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += A[i * N + j];      // (symmetric as well)
    rmean[tid] = s / (float)N;
  }
  __syncthreads();
  float* W = a.wt + bl * 4 * N * NBP;
  for (int e = tid; e < N * NBP; e += 256) {
    const int j = e / NBP, i = e - j * NBP;
    const bool in = i < N;
    const int q = j * N + (in ? i : 0);
    W[(0 * N + j) * NBP + i] = in ? 2.f * a.wh * G[2 * NN + q] + A1[q] - rmean[j] : 0.f;       // com_app <- aq
    W[(1 * N + j) * NBP + i] = in ? 2.f * a.wh * G[3 * NN + q] + A2[q] - rmean[N + j] : 0.f;   // com_mot <- mq
    W[(2 * N + j) * NBP + i] = in ? 2.f * a.wh * G[q] : 0.f;                                    // aq      <- com_app
    W[(3 * N + j) * NBP + i] = in ? 2.f * a.wh * G[NN + q] : 0.f;                               // mq      <- com_mot
  }
  if (tid == 0) {
    atomicAdd(a.loss + 1, (red[0] + red[1] + red[2] + red[3]) / ((float)a.B * N * N * a.Ltot));
    atomicAdd(a.loss + 2, (red[4] + red[5] + red[6] + red[7]) / (float)a.Ltot);
  }
}

// ---- pass 3: g_t = W_t^T c_t on the fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products) ------------------------
// One workgroup per (b, l, tensor); a wave takes 64-column chunks of D.  A = W^T (rows i, reduction index j) is loaded ONCE
// per wave as fragments (lane (r, q): W[4 ks + q][16 I + r]); B = the centred rows c[j][d].  The four column blocks of a
// chunk are INTERLEAVED -- block e holds the columns 4 r + e -- so a lane's B operands of one k-step are ONE float4 load
// (16 lanes x 16 B = a whole 256-byte row piece) and its four results of an output row are ONE float4 store.  The column
// means come from the loaded fragments themselves (in-lane sum over the k-steps, then over the four lane groups), and the
// rows are centred BEFORE the products (test_fused_loss_rows_with_a_large_shared_component).
// (Until round 3 this pass ran on the VALU, one column per thread with scalar-loaded W rows: 28 us at N = 20, 650 us at
// N = 64, where it left the main stream idle at the end of the forward.)
template <int NT>
__global__ __launch_bounds__(256) void k_aux_grad(const AuxArgs a) {
  const int b = blockIdx.x, l = blockIdx.y >> 2, t = blockIdx.y & 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, D = a.D, NB = a.NBP;
  const int64_t bl = (int64_t)l * a.B + b;
  const float* __restrict__ W = a.wt + (bl * 4 + t) * N * NB;
  const float* __restrict__ E = a.t[t] + bl * N * D;
  float* __restrict__ g = a.g[t] + bl * N * D;
  const int r = lane & 15, kq = lane >> 4;
  constexpr int KS = NT * 4;                          // k-steps of four rows j: 16 NT >= N
  const int ksn = (N + 3) >> 2;
  // W^T fragments come from an LDS image of W (zero padded to 16 NT x 16 NT; row stride = 16 mod 32 floats: the four lane
  // groups of a fragment read hit different banks); in registers they cost 16 NT VGPRs = the second wave per SIMD at N = 64
  constexpr int NBS = (NT & 1) ? 16 * NT : 16 * NT + 16;
  __shared__ float Wl[16 * NT * NBS];
  for (int e = tid; e < 16 * NT * 16 * NT; e += 256) {
    const int j = e / (16 * NT), i = e - j * (16 * NT);
    Wl[j * NBS + i] = (j < N && i < NB) ? W[j * NB + i] : 0.f;
  }
  __syncthreads();
  const float inv_n = 1.0f / (float)N;
  for (int c = wave; c < D / 64; c += 4) {
    const int d0 = c * 64 + 4 * r;
    float4 bf[KS];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int j = 4 * ks + kq;
      bf[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ks < ksn && j < N) bf[ks] = *reinterpret_cast<const float4*>(E + (int64_t)j * D + d0);
      s.x += bf[ks].x; s.y += bf[ks].y; s.z += bf[ks].z; s.w += bf[ks].w;
    }
    s.x += __shfl_xor(s.x, 16); s.y += __shfl_xor(s.y, 16); s.z += __shfl_xor(s.z, 16); s.w += __shfl_xor(s.w, 16);
    s.x += __shfl_xor(s.x, 32); s.y += __shfl_xor(s.y, 32); s.z += __shfl_xor(s.z, 32); s.w += __shfl_xor(s.w, 32);
    s.x *= inv_n; s.y *= inv_n; s.z *= inv_n; s.w *= inv_n;
    floatx4 acc[NT][4];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[I][e] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks < ksn) {                                   // (wave-uniform)
        const bool in = 4 * ks + kq < N;
        const float b0 = in ? bf[ks].x - s.x : 0.f, b1 = in ? bf[ks].y - s.y : 0.f;
        const float b2 = in ? bf[ks].z - s.z : 0.f, b3 = in ? bf[ks].w - s.w : 0.f;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
          const float af = Wl[(4 * ks + kq) * NBS + 16 * I + r];
          acc[I][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b0, acc[I][0], 0, 0, 0);
          acc[I][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b1, acc[I][1], 0, 0, 0);
          acc[I][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b2, acc[I][2], 0, 0, 0);
          acc[I][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b3, acc[I][3], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * I + 4 * kq + q;
        if (i < N)
          *reinterpret_cast<float4*>(g + (int64_t)i * D + d0) = make_float4(acc[I][0][q], acc[I][1][q], acc[I][2][q], acc[I][3][q]);
      }
  }
}

__global__ __launch_bounds__(256) void k_sum_and_ones(const float* x, float* g_ones, int64_t n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    s += x[i];
    if (g_ones) g_ones[i] = 1.0f;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

template <int NT>
static int launch_gram_nt(hipStream_t st, const AuxArgs& a) {
  if (NT <= 2) {
    const size_t lds = sizeof(float) * ((size_t)a.D + (NT * (NT + 1) / 2) * 256);
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_aux_gram_direct<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DVGR_LAUNCH(k_aux_gram_direct<NT>, dim3(a.B, a.L * 4), dim3(256), lds, st, a);
    return (int)hipGetLastError();
  }
  const size_t lds = sizeof(float) * ((size_t)4 * 16 * NT * GramCfg<NT>::STR + (NT * (NT + 1) / 2) * 256);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_aux_gram<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DVGR_LAUNCH(k_aux_gram<NT>, dim3(a.B, a.L * 4), dim3(256), lds, st, a);
  return (int)hipGetLastError();
}
static int launch_gram(hipStream_t st, const AuxArgs& a) {
  switch ((a.N + 15) / 16) {
    case 1: return launch_gram_nt<1>(st, a);
    case 2: return launch_gram_nt<2>(st, a);
    case 3: return launch_gram_nt<3>(st, a);
    default: return launch_gram_nt<4>(st, a);
  }
}

int sum_and_ones(hipStream_t st, const float* x, float* g_ones, int64_t n, float* out) {
  DVGR_LAUNCH(k_sum_and_ones, dim3(256), dim3(256), 0, st, x, g_ones, n, out);
  return (int)hipGetLastError();
}

static int nb_of(int N) { return N <= 8 ? 8 : N <= 16 ? 16 : N <= 24 ? 24 : N <= 32 ? 32 : 64; }

template <int NT>
static int launch_grad(hipStream_t st, const AuxArgs& a) {
  DVGR_LAUNCH(k_aux_grad<NT>, dim3(a.B, a.L * 4), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

size_t aux_ws_floats(const dvgr_dims& d) {
  const size_t LB = (size_t)d.L * d.B, NN = (size_t)d.N * d.N;
  return LB * 4 * NN + LB * 4 * (size_t)d.N * nb_of(d.N) + 64;
}

// first / count: the list entries [first, first + count) only (count < 0: all d.L of them).  The losses are means over ALL d.L
// entries and every kernel works on one (sample, entry) at a time, so the forward sends the entries out in groups as they come
// into existence; the values add up in loss3, the gradients of different entries are different tensors.
int aux_losses(hipStream_t st, const dvgr_dims& d, const dvgr_io* io, float alpha, float beta, float dep_scale,
               float* loss3, float* g_vis, float* const g[4], float* ws, bool zero_loss3, int first, int count) {
  if (count < 0) { first = 0; count = d.L; }
  if (first < 0 || count < 1 || first + count > d.L) return DVGR_E_DIMS;
  AuxArgs a{};
  const int64_t e_off = (int64_t)first * d.B * d.N * d.D;
  a.t[0] = io->com_app + e_off; a.t[1] = io->com_mot + e_off; a.t[2] = io->aq_fusion + e_off; a.t[3] = io->mq_fusion + e_off;
  for (int i = 0; i < 4; ++i) a.g[i] = g[i] + e_off;
  const size_t LB = (size_t)d.L * d.B, NN = (size_t)d.N * d.N;
  a.NBP = nb_of(d.N);
  a.gram = ws + (size_t)first * d.B * 4 * NN;
  a.wt = ws + round64((int64_t)(LB * 4 * NN)) + (size_t)first * d.B * 4 * d.N * a.NBP;
  a.loss = loss3;
  a.wc = alpha / ((float)d.L * d.B * d.N * d.N);
  a.wh = beta * dep_scale / (float)d.L;
  a.B = d.B; a.N = d.N; a.D = d.D; a.L = count; a.Ltot = d.L;
  hipError_t e = hipSuccess;
  if (zero_loss3) {                        // (the forward clears the three scalars itself, on the way of its first kernel)
    e = hipMemsetAsync(loss3, 0, 3 * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t nv = (int64_t)d.B * d.N * d.D;
  if (io->vis) DVGR_LAUNCH(k_sum_and_ones, dim3(256), dim3(256), 0, st, io->vis, g_vis, nv, loss3);
  DVGR_TRY_HIP(launch_gram(st, a));
  const size_t lds2 = sizeof(float) * (6 * NN + 6 * d.N + 8);
  if (lds2 > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_aux_combine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  DVGR_LAUNCH(k_aux_combine, dim3(a.B, a.L), dim3(256), lds2, st, a);
  if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  // (sending the entries out in groups of unit layers as they come into existence, each group under the layers that
  // follow, changes nothing at C2 or C5: profiles/experiments_r03.tsv, c5_aux_groups)
  switch ((d.N + 15) / 16) {
    case 1: return launch_grad<1>(st, a);
    case 2: return launch_grad<2>(st, a);
    case 3: return launch_grad<3>(st, a);
    default: return launch_grad<4>(st, a);
  }
}

}  // namespace dvgr
