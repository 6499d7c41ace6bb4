// dvgr_head.hip -- clip read-out + classifier after the unit (include/dvgr_head.h).
// Restates model/AnswerDecoder.py:155-202 as chained by model/models.py:80-81 (checked against oracle/head_torch.py):
//   xd = drop0(vis);  e = ELU(xd Wv^T);  a = softmax_n(e w + b0);  vd = sum_n a_n xd_n
//   cat = drop1([vd | qe Wq^T + bq]);  he = ELU(cat W1^T + b1);  y = gamma (he - mean) / sqrt(var + eps) + beta
//   logits = drop2(y) W2^T + b2
// Contractions go through the unit's engine (dvgr_gemm.h); the rest are the small kernels below.
#include "../../include/dvgr_head.h"
#include "dvgr_gemm.h"
#include "dvgr_side.h"
#include "dvgr_stages.h"

using namespace dvgr;

#define HEAD_TRY(expr)             \
  do {                             \
    const int _e = (expr);         \
    if (_e != 0) return _e;        \
  } while (0)
#define HEAD_LAUNCH_OK() do { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

constexpr float kBnEps = 1e-5f, kBnMomentum = 0.1f;
constexpr uint32_t kHeadSiteBase = 100000;    // dropout sites of the head, clear of the unit's (layer * 24 + ...)

struct HeadParams { int64_t wv, wa, ba, wq, bq, w1, b1, gamma, beta, w2, b2, total; };
HeadParams head_params(const dvgr_head_dims& d) {
  HeadParams p{};
  const int64_t D = d.D, C = d.C;
  int64_t o = 0;
  p.wv = o; o += round64(D * D);
  p.wa = o; o += round64(D);
  p.ba = o; o += 64;
  p.wq = o; o += round64(D * D);
  p.bq = o; o += round64(D);
  p.w1 = o; o += round64(D * 2 * D);
  p.b1 = o; o += round64(D);
  p.gamma = o; o += round64(D);
  p.beta = o; o += round64(D);
  p.w2 = o; o += round64(C * D);
  p.b2 = o; o += round64(C);
  p.total = o;
  return p;
}

struct HeadSaved { int64_t xd, e, attn, cat, he, mean, istd, yd, total; };
HeadSaved head_saved(const dvgr_head_dims& d) {
  HeadSaved s{};
  const int64_t B = d.B, N = d.N, D = d.D;
  int64_t o = 0;
  s.xd = o; o += round64(B * N * D);
  s.e = o; o += round64(B * N * D);
  s.attn = o; o += round64(B * N);
  s.cat = o; o += round64(B * 2 * D);
  s.he = o; o += round64(B * D);
  s.mean = o; o += round64(D);
  s.istd = o; o += round64(D);
  s.yd = o; o += round64(B * D);
  s.total = o;
  return s;
}

struct HeadWs { int64_t dyd, dpre1, dcat, dl, dvp, total; };
HeadWs head_ws(const dvgr_head_dims& d) {
  HeadWs w{};
  const int64_t B = d.B, N = d.N, D = d.D;
  int64_t o = 0;
  w.dyd = o; o += round64(B * D);
  w.dpre1 = o; o += round64(B * D);
  w.dcat = o; o += round64(B * 2 * D);
  w.dl = o; o += round64(B * N);
  w.dvp = o; o += round64(B * N * D);
  w.total = o;
  return w;
}

uint32_t head_key(int train, uint64_t seed, int site) { return train ? site_key(seed, kHeadSiteBase + (uint32_t)site) : 0u; }

GemmProblem nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N,
               int K, int act) {
  GemmProblem p{};
  p.A = operand(A, lda); p.B = operand(B, ldb);
  p.C = C; p.ldc = ldc; p.bias = bias; p.M = M; p.N = N; p.K = K; p.kseg = K; p.act = act;
  return p;
}
GemmProblem nn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, int accumulate) {
  GemmProblem p{};
  p.A = operand(A, lda); p.B = operand(B, ldb);
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.kseg = K; p.accumulate = accumulate;
  return p;
}
GemmProblem tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmProblem p{};
  p.A = operand(A, lda); p.B = operand(B, ldb);
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.kseg = K;
  p.atomic_ok = 1;   // weight gradients land in gparams, zeroed at the start of the backward
  return p;
}
int launch1(hipStream_t st, int la, int lb, const GemmProblem& p, int prec) {
  GemmBatch b{};
  b.p[b.n++] = p;
  return gemm_launch(st, la, lb, b, TAG_GENERIC, prec);
}

// ---- element-wise dropout: out = x * keep / (1 - p); rows of `cols` elements, row stride ld (in == out allowed) ----
__global__ __launch_bounds__(256) void k_head_drop(const float* x, float* out, uint32_t key, int64_t rows, int cols, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols, c = i - r * cols;
  out[r * ld + c] = x[r * ld + c] * drop_factor(key, (uint32_t)i);
}
int head_drop(hipStream_t st, const float* x, float* out, uint32_t key, int64_t rows, int cols, int64_t ld) {
  DVGR_LAUNCH(k_head_drop, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, st, x, out, key, rows, cols, ld);
  return (int)hipGetLastError();
}

// ---- read-out forward: one workgroup per sample ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_readout_fwd(const float* e, const float* xd, const float* wa, const float* ba,
                                                     float* attn, float* cat, int N, int D) {
  __shared__ float sh_l[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* eb = e + (int64_t)b * N * D;
  const float* xb = xd + (int64_t)b * N * D;
  for (int n = wave; n < N; n += 4) {                       // logit_n = e_n . w + b0
    float s = 0.f;
    for (int k = lane; k < D / 4; k += 64) {
      const float4 u = reinterpret_cast<const float4*>(eb + (int64_t)n * D)[k], w = reinterpret_cast<const float4*>(wa)[k];
      s += u.x * w.x + u.y * w.y + u.z * w.z + u.w * w.w;
    }
    s = wave_sum(s);
    if (lane == 0) sh_l[n] = s + ba[0];
  }
  __syncthreads();
  if (wave == 0) {                                          // softmax over the clips (N <= 64: one wave)
    const float l = lane < N ? sh_l[lane] : -INFINITY;
    const float mx = wave_max(l);
    const float pe = lane < N ? expf(l - mx) : 0.f;
    const float sum = wave_sum(pe);
    if (lane < N) { sh_l[lane] = pe / sum; attn[(int64_t)b * N + lane] = pe / sum; }
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) {                      // v_distill = sum_n a_n xd_n, into the left half of `cat`
    float acc = 0.f;
#pragma unroll 4
    for (int n = 0; n < N; ++n) acc += sh_l[n] * xb[(int64_t)n * D + d];
    cat[(int64_t)b * 2 * D + d] = acc;
  }
}

// ---- BatchNorm1d + dropout: one thread per feature ------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bn_fwd(const float* he, const float* gamma, const float* beta, float* run_mean,
                                                float* run_var, float* s_mean, float* s_istd, float* yd, uint32_t key, int B,
                                                int D, int train) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= D) return;
  float mean, var;
  if (train) {
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) s += he[(int64_t)b * D + d];
    mean = s / (float)B;
    float q = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) { const float c = he[(int64_t)b * D + d] - mean; q += c * c; }
    var = q / (float)B;                                     // biased: what the normalisation uses
    run_mean[d] = (1.f - kBnMomentum) * run_mean[d] + kBnMomentum * mean;
    run_var[d] = (1.f - kBnMomentum) * run_var[d] + kBnMomentum * (B > 1 ? q / (float)(B - 1) : var);
  } else {
    mean = run_mean[d];
    var = run_var[d];
  }
  const float istd = 1.0f / sqrtf(var + kBnEps);
  s_mean[d] = mean;
  s_istd[d] = istd;
  const float g = gamma[d], bt = beta[d];
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    const int64_t i = (int64_t)b * D + d;
    yd[i] = (g * (he[i] - mean) * istd + bt) * drop_factor(key, (uint32_t)i);
  }
}

// dyd -> (dgamma, dbeta, dpre1 = dhe * ELU'(he))
__global__ __launch_bounds__(64) void k_bn_bwd(const float* dyd, const float* he, const float* gamma, const float* s_mean,
                                                const float* s_istd, float* dgamma, float* dbeta, float* dpre1, uint32_t key,
                                                int B, int D, int train) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= D) return;
  const float mean = s_mean[d], istd = s_istd[d], g = gamma[d];
  float sb = 0.f, sg = 0.f;
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    const int64_t i = (int64_t)b * D + d;
    const float dy = dyd[i] * drop_factor(key, (uint32_t)i);
    sb += dy;
    sg += dy * (he[i] - mean) * istd;
  }
  dgamma[d] = sg;
  dbeta[d] = sb;
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    const int64_t i = (int64_t)b * D + d;
    const float dy = dyd[i] * drop_factor(key, (uint32_t)i);
    const float xh = (he[i] - mean) * istd;
    const float dhe = train ? g * istd * (dy - sb / (float)B - xh * sg / (float)B) : g * istd * dy;
    dpre1[i] = dhe * elu_grad_from_out(he[i]);
  }
}

// ---- read-out backward: one workgroup per sample --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_readout_bwd(const float* dcat, const float* xd, const float* e, const float* attn,
                                                     const float* wa, float* dl_out, float* dxd, float* dvp, int N, int D) {
  __shared__ float sh_a[64], sh_dl[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* dvd = dcat + (int64_t)b * 2 * D;             // gradient of v_distill = left half of dcat
  const float* xb = xd + (int64_t)b * N * D;
  const float* eb = e + (int64_t)b * N * D;
  if (tid < N) sh_a[tid] = attn[(int64_t)b * N + tid];
  for (int n = wave; n < N; n += 4) {                       // da_n = dvd . xd_n
    float s = 0.f;
    for (int k = lane; k < D / 4; k += 64) {
      const float4 u = reinterpret_cast<const float4*>(dvd)[k], x = reinterpret_cast<const float4*>(xb + (int64_t)n * D)[k];
      s += u.x * x.x + u.y * x.y + u.z * x.z + u.w * x.w;
    }
    s = wave_sum(s);
    if (lane == 0) sh_dl[n] = s;
  }
  __syncthreads();
  if (wave == 0) {                                          // softmax backward
    const float a = lane < N ? sh_a[lane] : 0.f, da = lane < N ? sh_dl[lane] : 0.f;
    const float dot = wave_sum(a * da);
    const float dl = a * (da - dot);
    if (lane < N) { sh_dl[lane] = dl; dl_out[(int64_t)b * N + lane] = dl; }
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    const float g = dvd[d], w = wa[d];
#pragma unroll 4
    for (int n = 0; n < N; ++n) {
      const int64_t i = (int64_t)b * N * D + (int64_t)n * D + d;
      dxd[i] = sh_a[n] * g;                                 // through the weighted sum; dvp . Wv is added by the contraction
      dvp[i] = sh_dl[n] * w * elu_grad_from_out(eb[(int64_t)n * D + d]);
    }
  }
}

// ---- cross-entropy, value + gradient: one wave per sample --------------------------------------------------------
__global__ __launch_bounds__(256) void k_cross_entropy(const float* logits, const int64_t* target, float* loss, float* g, int B,
                                                       int C, int Cv, float scale) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* row = logits + (int64_t)b * C;
  float mx = -INFINITY;
  for (int c = lane; c < Cv; c += 64) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < Cv; c += 64) se += expf(row[c] - mx);
  se = wave_sum(se);
  const int t = (int)target[b];
  const float inv = scale / (float)B;
  for (int c = lane; c < C; c += 64)
    g[(int64_t)b * C + c] = c < Cv ? (expf(row[c] - mx) / se - (c == t ? 1.f : 0.f)) * inv : 0.f;
  if (lane == 0) atomicAdd(loss, (logf(se) + mx - row[t]) / (float)B);
}

int check_head(const dvgr_head_dims* d) {
  if (!d) return DVGR_E_NULL;
  if (d->B < 1 || d->N < 1 || d->N > 64 || d->D < 64 || (d->D % 64) || d->C < 4 || (d->C % 4)) return DVGR_E_DIMS;
  if (d->dtype != DVGR_F32 && d->dtype != DVGR_BF16 && d->dtype != DVGR_BF16X3) return DVGR_E_DIMS;
  return 0;
}
inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

namespace dvgr { int keep_mask_launch(hipStream_t st, uint32_t key, int64_t n, float* keep); }

extern "C" {

int dvgr_head_check_dims(const dvgr_head_dims* d) { return check_head(d); }
int64_t dvgr_head_param_count(const dvgr_head_dims* d) { return (d && !check_head(d)) ? head_params(*d).total : 0; }
int dvgr_head_param_offset(const dvgr_head_dims* d, int kind, int64_t* offset, int64_t* rows, int64_t* cols) {
  HEAD_TRY(check_head(d));
  if (!offset || !rows || !cols) return DVGR_E_NULL;
  const HeadParams p = head_params(*d);
  const int64_t D = d->D, C = d->C;
  switch (kind) {
    case DVGR_HP_VPROJ_W: *offset = p.wv; *rows = D; *cols = D; break;
    case DVGR_HP_ATTN_W: *offset = p.wa; *rows = 1; *cols = D; break;
    case DVGR_HP_ATTN_B: *offset = p.ba; *rows = 1; *cols = 1; break;
    case DVGR_HP_QPROJ_W: *offset = p.wq; *rows = D; *cols = D; break;
    case DVGR_HP_QPROJ_B: *offset = p.bq; *rows = 1; *cols = D; break;
    case DVGR_HP_FC1_W: *offset = p.w1; *rows = D; *cols = 2 * D; break;
    case DVGR_HP_FC1_B: *offset = p.b1; *rows = 1; *cols = D; break;
    case DVGR_HP_BN_W: *offset = p.gamma; *rows = 1; *cols = D; break;
    case DVGR_HP_BN_B: *offset = p.beta; *rows = 1; *cols = D; break;
    case DVGR_HP_FC2_W: *offset = p.w2; *rows = C; *cols = D; break;
    case DVGR_HP_FC2_B: *offset = p.b2; *rows = 1; *cols = C; break;
    default: return DVGR_E_KIND;
  }
  return 0;
}
size_t dvgr_head_saved_bytes(const dvgr_head_dims* d) { return (d && !check_head(d)) ? (size_t)head_saved(*d).total * 4 : 0; }
size_t dvgr_head_workspace_bytes(const dvgr_head_dims* d) { return (d && !check_head(d)) ? (size_t)head_ws(*d).total * 4 : 0; }

int dvgr_head_forward(const dvgr_head_dims* dp, const float* params, const dvgr_head_io* io, void* saved_v, size_t saved_bytes,
                      void* ws_v, size_t ws_bytes, int train, uint64_t seed, void* stream) {
  HEAD_TRY(check_head(dp));
  if (!params || !io || !saved_v || !ws_v || !io->vis || !io->qe || !io->logits || !io->run_mean || !io->run_var) return DVGR_E_NULL;
  if (saved_bytes < dvgr_head_saved_bytes(dp) || ws_bytes < dvgr_head_workspace_bytes(dp)) return DVGR_E_WORKSPACE;
  if (!al16(params) || !al16(io->vis) || !al16(io->qe) || !al16(io->logits) || !al16(saved_v) || !al16(ws_v)) return DVGR_E_ALIGN;
  const dvgr_head_dims d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const HeadParams P = head_params(d);
  const HeadSaved S = head_saved(d);
  float* sv = (float*)saved_v;
  const int B = d.B, N = d.N, D = d.D, C = d.C, BN = B * N, prec = d.dtype;
  const int drop = (train & 1) && !(train & DVGR_HEAD_TRAIN_NO_DROPOUT);   // the three nn.Dropout(0.15) sites
  const int bn_train = train & 1;                                         // BatchNorm1d: batch statistics + running update
  // read-out
  const float* xd = io->vis;
  if (drop) {
    HEAD_TRY(head_drop(st, io->vis, sv + S.xd, head_key(1, seed, 0), (int64_t)BN, D, D));
    xd = sv + S.xd;
  }
  HEAD_TRY(launch1(st, KC, KC, nt(xd, D, params + P.wv, D, sv + S.e, D, nullptr, BN, D, D, ACT_ELU), prec));
  DVGR_LAUNCH(k_readout_fwd, dim3(B), dim3(256), 0, st, sv + S.e, xd, params + P.wa, params + P.ba, sv + S.attn, sv + S.cat, N, D);
  HEAD_LAUNCH_OK();
  // classifier
  HEAD_TRY(launch1(st, KC, KC, nt(io->qe, D, params + P.wq, D, sv + S.cat + D, 2 * D, params + P.bq, B, D, D, ACT_NONE), prec));
  if (drop) HEAD_TRY(head_drop(st, sv + S.cat, sv + S.cat, head_key(1, seed, 1), B, 2 * D, 2 * D));
  HEAD_TRY(launch1(st, KC, KC, nt(sv + S.cat, 2 * D, params + P.w1, 2 * D, sv + S.he, D, params + P.b1, B, D, 2 * D, ACT_ELU), prec));
  DVGR_LAUNCH(k_bn_fwd, dim3((D + 63) / 64), dim3(64), 0, st, sv + S.he, params + P.gamma, params + P.beta, io->run_mean,
                     io->run_var, sv + S.mean, sv + S.istd, sv + S.yd, head_key(drop, seed, 2), B, D, bn_train);
  HEAD_LAUNCH_OK();
  HEAD_TRY(launch1(st, KC, KC, nt(sv + S.yd, D, params + P.w2, D, io->logits, C, params + P.b2, B, C, D, ACT_NONE), prec));
  return 0;
}

int dvgr_head_backward(const dvgr_head_dims* dp, const float* params, const dvgr_head_io* io, const dvgr_head_grads* g,
                       const void* saved_v, size_t saved_bytes, void* ws_v, size_t ws_bytes, int train, uint64_t seed, void* stream) {
  HEAD_TRY(check_head(dp));
  if (!params || !io || !g || !saved_v || !ws_v || !io->vis || !io->qe || !g->g_logits || !g->g_vis || !g->g_qe || !g->gparams)
    return DVGR_E_NULL;
  if (saved_bytes < dvgr_head_saved_bytes(dp) || ws_bytes < dvgr_head_workspace_bytes(dp)) return DVGR_E_WORKSPACE;
  if (!al16(params) || !al16(g->g_logits) || !al16(g->g_vis) || !al16(g->g_qe) || !al16(g->gparams) || !al16(saved_v) || !al16(ws_v))
    return DVGR_E_ALIGN;
  const dvgr_head_dims d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const HeadParams P = head_params(d);
  const HeadSaved S = head_saved(d);
  const HeadWs W = head_ws(d);
  const float* sv = (const float*)saved_v;
  float* ws = (float*)ws_v;
  float* gp = g->gparams;
  const int B = d.B, N = d.N, D = d.D, C = d.C, BN = B * N, prec = d.dtype;
  const int drop = (train & 1) && !(train & DVGR_HEAD_TRAIN_NO_DROPOUT);
  const int bn_train = train & 1;
  const float* xd = drop ? sv + S.xd : io->vis;
  {
    const hipError_t e = hipMemsetAsync(gp, 0, (size_t)P.total * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  // Two streams, as the unit's backward: the chain of input gradients stays on `st`, the weight gradients and column
  // sums follow on the side stream behind two forks (each fork costs `st` a ~6 us bubble, so not one per dependency).
  SideStream* side = side_stream(st);
  hipStream_t hs = side ? side->st : st;
  ColsumBatch cb{};
  // logits = yd W2^T + b2
  cb.j[cb.n++] = ColsumJob{g->g_logits, nullptr, nullptr, nullptr, B, 0, C, C, gp + P.b2};
  HEAD_TRY(launch1(st, KC, IC, nn(g->g_logits, C, params + P.w2, D, ws + W.dyd, D, B, D, C, 0), prec));
  // dropout 2, BatchNorm1d, ELU
  DVGR_LAUNCH(k_bn_bwd, dim3((D + 63) / 64), dim3(64), 0, st, ws + W.dyd, sv + S.he, params + P.gamma, sv + S.mean,
                     sv + S.istd, gp + P.gamma, gp + P.beta, ws + W.dpre1, head_key(drop, seed, 2), B, D, bn_train);
  HEAD_LAUNCH_OK();
  if (side) HEAD_TRY(side->order(st, hs));                  // fork 1: gparams is zeroed, dpre1 final
  HEAD_TRY(launch1(hs, IC, IC, tn(g->g_logits, C, sv + S.yd, D, gp + P.w2, D, C, D, B), prec));
  HEAD_TRY(launch1(hs, IC, IC, tn(ws + W.dpre1, D, sv + S.cat, 2 * D, gp + P.w1, 2 * D, D, 2 * D, B), prec));
  cb.j[cb.n++] = ColsumJob{ws + W.dpre1, nullptr, nullptr, nullptr, B, 0, D, D, gp + P.b1};
  // he = ELU(cat W1^T + b1)
  HEAD_TRY(launch1(st, KC, IC, nn(ws + W.dpre1, D, params + P.w1, 2 * D, ws + W.dcat, 2 * D, B, 2 * D, D, 0), prec));
  if (drop) HEAD_TRY(head_drop(st, ws + W.dcat, ws + W.dcat, head_key(1, seed, 1), B, 2 * D, 2 * D));
  // question_proj: right half of dcat
  cb.j[cb.n++] = ColsumJob{ws + W.dcat + D, nullptr, nullptr, nullptr, B, 0, D, 2 * D, gp + P.bq};
  HEAD_TRY(launch1(st, KC, IC, nn(ws + W.dcat + D, 2 * D, params + P.wq, D, g->g_qe, D, B, D, D, 0), prec));
  // read-out
  DVGR_LAUNCH(k_readout_bwd, dim3(B), dim3(256), 0, st, ws + W.dcat, xd, sv + S.e, sv + S.attn, params + P.wa, ws + W.dl,
                     g->g_vis, ws + W.dvp, N, D);
  HEAD_LAUNCH_OK();
  if (side) HEAD_TRY(side->order(st, hs));                  // fork 2: dcat, dl, dvp final
  HEAD_TRY(launch1(hs, IC, IC, tn(ws + W.dcat + D, 2 * D, io->qe, D, gp + P.wq, D, D, D, B), prec));
  HEAD_TRY(launch1(hs, IC, IC, tn(ws + W.dvp, D, xd, D, gp + P.wv, D, D, D, BN), prec));
  cb.j[cb.n++] = ColsumJob{sv + S.e, nullptr, ws + W.dl, nullptr, BN, 0, D, D, gp + P.wa};
  cb.j[cb.n++] = ColsumJob{ws + W.dl, nullptr, nullptr, nullptr, BN, 0, 1, 1, gp + P.ba};
  cb.chunks = 8;
  HEAD_TRY(colsum(hs, cb));
  HEAD_TRY(launch1(st, KC, IC, nn(ws + W.dvp, D, params + P.wv, D, g->g_vis, D, BN, D, D, 1), prec));
  if (drop) HEAD_TRY(head_drop(st, g->g_vis, g->g_vis, head_key(1, seed, 0), (int64_t)BN, D, D));
  if (side) HEAD_TRY(side->order(hs, st));                  // join
  return 0;
}

int dvgr_head_cross_entropy(const dvgr_head_dims* dp, int C_valid, const float* logits, const int64_t* target, float scale,
                            float* loss, float* g_logits, void* stream) {
  HEAD_TRY(check_head(dp));
  if (!logits || !target || !loss || !g_logits) return DVGR_E_NULL;
  if (C_valid < 1 || C_valid > dp->C) return DVGR_E_DIMS;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  DVGR_LAUNCH(k_cross_entropy, dim3((dp->B + 3) / 4), dim3(256), 0, st, logits, target, loss, g_logits, dp->B, dp->C, C_valid, scale);
  return (int)hipGetLastError();
}

int dvgr_head_dbg_keep_mask(uint64_t seed, int site, int64_t n, float* keep, void* stream) {
  if (!keep || site < 0 || site > 2) return DVGR_E_NULL;
  return keep_mask_launch((hipStream_t)stream, head_key(1, seed, site), n, keep);
}

}  // extern "C"
