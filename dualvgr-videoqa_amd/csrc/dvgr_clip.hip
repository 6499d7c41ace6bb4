// dvgr_clip.hip -- appearance encoder in front of the unit (include/dvgr_clip.h).
// Restates model/Preprocessing.py:191-234 (checked against oracle/clip_torch.py = torch.nn.LSTM on the CPU):
//   e = tanh(drop_in(x));  G = e [Wih_f; Wih_r]^T + (b_ih + b_hh)            one contraction over all T * S rows
//   step k = 0..T-1 (forward direction at frame k, reverse direction at frame T-1-k, both in the same launches):
//     G[t] += h_prev Whh^T;  i, f, o = sigmoid(.), g = tanh(.) of it;  c = f c_prev + i g;  h = o tanh(c)
//   out = drop_out([h_f(T-1) | h_r(0)])
// Backward: the steps in reverse (cell kernel + dh_prev = dG[t] Whh per step), then the weight gradients as three
// contractions with the whole T * S extent as the reduction (dWih = dG^T e, dWhh = dG^T h_prev) and one column sum.
// Rows are time-major (row = t * S + s) so every step's operands are contiguous row blocks.
// Contractions go through the unit's engine (dvgr_gemm.h) with bf16 hi/lo planes written by the producing kernels.
#include "../../include/dvgr_clip.h"
#include "dvgr_gemm.h"
#include "dvgr_stages.h"

using namespace dvgr;

#define CLIP_TRY(expr)             \
  do {                             \
    const int _e = (expr);         \
    if (_e != 0) return _e;        \
  } while (0)
#define CLIP_LAUNCH_OK() do { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

constexpr uint32_t kClipSiteBase = 200000;        // dropout sites of the encoder, clear of the unit's and the head's
constexpr uint32_t kThreshIn = 9830;              // floor(0.15 * 65536)   model/Preprocessing.py:199
constexpr float kScaleIn = 1.0f / (1.0f - 0.15f);
constexpr uint32_t kThreshOut = 11796;            // floor(0.18 * 65536)   model/Preprocessing.py:200
constexpr float kScaleOut = 1.0f / (1.0f - 0.18f);

struct ClipParams { int64_t wih, whh, bih, bhh, total; };
ClipParams clip_params(const dvgr_clip_dims& d) {
  ClipParams p{};
  const int64_t H = d.H, V = d.V;
  int64_t o = 0;
  p.wih = o; o += 8 * H * V;       // [dir][4H][V]
  p.whh = o; o += 8 * H * H;       // [dir][4H][H]
  p.bih = o; o += 8 * H;
  p.bhh = o; o += 8 * H;
  p.total = o;
  return p;
}

// saved for the backward; *_p are bf16 hi/lo plane pairs (hi plane, then lo plane: together the size of the fp32 tensor)
struct ClipSaved { int64_t e, e_p, g, c, h, h_p, total; };
ClipSaved clip_saved(const dvgr_clip_dims& d) {
  ClipSaved s{};
  const int64_t TS = (int64_t)d.T * d.S, H = d.H, V = d.V;
  int64_t o = 0;
  s.e = o; o += TS * V;
  s.e_p = o; o += TS * V;
  s.g = o; o += TS * 8 * H;        // gate pre-activations (forward), then their gradients (backward)
  s.c = o; o += TS * 2 * H;
  s.h = o; o += TS * 2 * H;
  s.h_p = o; o += TS * 2 * H;
  s.total = o;
  return s;
}

struct ClipWs { int64_t w_p, bias, g_p, dh, dc, carry, de, total; };
ClipWs clip_ws(const dvgr_clip_dims& d) {
  ClipWs w{};
  const int64_t TS = (int64_t)d.T * d.S, H = d.H, V = d.V, S = d.S;
  int64_t o = 0;
  w.w_p = o; o += 8 * H * V + 8 * H * H;   // planes of [wih | whh]
  w.bias = o; o += 8 * H;
  w.g_p = o; o += TS * 8 * H;              // planes of the step gradients
  w.dh = o; o += round64(S * 2 * H);
  w.dc = o; o += round64(S * 2 * H);
  w.carry = o; o += round64(S * 2 * H);    // dh of steps a sequence sat out (variable lengths), waiting for its last active step
  w.de = o; o += TS * V;                   // gradient of the embedded input, time-major (only when g_x is wanted)
  w.total = o;
  return w;
}

int check_clip(const dvgr_clip_dims* d) {
  if (!d) return DVGR_E_NULL;
  if (d->S < 1 || d->T < 1 || d->T > 64 || d->V < 4 || (d->V % 4) || d->H < 32 || (d->H % 32)) return DVGR_E_DIMS;
  if (d->dtype != DVGR_F32 && d->dtype != DVGR_BF16 && d->dtype != DVGR_BF16X3) return DVGR_E_DIMS;
  const int64_t TS = (int64_t)d->T * d->S, wide = d->V > 8 * d->H ? d->V : 8 * d->H;
  if (TS * wide >= (1ll << 31)) return DVGR_E_DIMS;      // 32-bit element offsets inside the engine / dropout counters
  return 0;
}
inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
uint32_t clip_key(int train, uint64_t seed, int site) { return train ? site_key(seed, kClipSiteBase + (uint32_t)site) : 0u; }

__device__ __forceinline__ float keep_scaled(uint32_t key, uint32_t idx, uint32_t thresh, float scale) {
  if (key == 0u) return 1.0f;
  return drop_keep_t(key, idx, thresh) ? scale : 0.0f;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- e[t*S + s][:] = tanh(drop(x[s][t][:])), fp32 + planes -------------------------------------------------------
__global__ __launch_bounds__(256) void k_clip_embed(const float* __restrict__ x, float* __restrict__ e, uint16_t* e_p, int64_t lo_off,
                                                    uint32_t key, int S, int T, int V, int embed) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n4 = (int64_t)S * T * V / 4;
  if (i4 >= n4) return;
  const int64_t idx = 4 * i4;
  const int v = (int)(idx % V);
  const int64_t st = idx / V;
  const int t = (int)(st % T);
  const int64_t s = st / T;
  float4 a = reinterpret_cast<const float4*>(x)[i4];
  if (embed) {
  a.x = tanhf(a.x * keep_scaled(key, (uint32_t)idx, kThreshIn, kScaleIn));
  a.y = tanhf(a.y * keep_scaled(key, (uint32_t)idx + 1, kThreshIn, kScaleIn));
  a.z = tanhf(a.z * keep_scaled(key, (uint32_t)idx + 2, kThreshIn, kScaleIn));
  a.w = tanhf(a.w * keep_scaled(key, (uint32_t)idx + 3, kThreshIn, kScaleIn));
  }
  const int64_t dst = ((int64_t)t * S + s) * V + v;
  *reinterpret_cast<float4*>(e + dst) = a;
  split_store4(e_p, lo_off, dst, a);
}

__global__ __launch_bounds__(256) void k_clip_bias(const float* bih, const float* bhh, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = bih[i] + bhh[i];
}

// ---- one step of both directions: gates -> (c, h) ---------------------------------------------------------------
// thread = 4 consecutive hidden units of one (clip, direction); row of direction d = t[d] * S + s
__global__ __launch_bounds__(256) void k_clip_cell_fwd(const float* __restrict__ G, float* __restrict__ Cs, float* __restrict__ Hs,
                                                       uint16_t* h_p, int64_t lo_off, int S, int H, int t0, int t1, int has_prev,
                                                       const int64_t* __restrict__ lens, float* __restrict__ out_seq, int T) {
  const int per_row = 2 * H / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)S * per_row) return;
  const int s = (int)(i / per_row), c4 = 4 * (int)(i % per_row);
  const int d = c4 / H, j = c4 - d * H;
  const int t = d ? t1 : t0, tp = d ? t1 + 1 : t0 - 1;
  const float* g = G + ((int64_t)t * S + s) * 8 * H + d * 4 * H + j;
  const float4 gi = *reinterpret_cast<const float4*>(g), gf = *reinterpret_cast<const float4*>(g + H),
               gg = *reinterpret_cast<const float4*>(g + 2 * H), go = *reinterpret_cast<const float4*>(g + 3 * H);
  float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_prev) cp = *reinterpret_cast<const float4*>(Cs + ((int64_t)tp * S + s) * 2 * H + c4);
  const int64_t o = ((int64_t)t * S + s) * 2 * H + c4;
  if (lens && t >= (int)lens[s]) {        // inactive step: the state passes through, the step emits zeros
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_prev) hp = *reinterpret_cast<const float4*>(Hs + ((int64_t)tp * S + s) * 2 * H + c4);
    *reinterpret_cast<float4*>(Cs + o) = cp;
    *reinterpret_cast<float4*>(Hs + o) = hp;
    split_store4(h_p, lo_off, o, hp);
    if (out_seq) *reinterpret_cast<float4*>(out_seq + ((int64_t)s * T + t) * 2 * H + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float pi[4] = {gi.x, gi.y, gi.z, gi.w}, pf[4] = {gf.x, gf.y, gf.z, gf.w}, pg[4] = {gg.x, gg.y, gg.z, gg.w},
              po[4] = {go.x, go.y, go.z, go.w}, pc[4] = {cp.x, cp.y, cp.z, cp.w};
  float ai[4], af[4], ag[4], ao[4], c[4], h[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ai[q] = sigmoidf_(pi[q]); af[q] = sigmoidf_(pf[q]); ag[q] = tanhf(pg[q]); ao[q] = sigmoidf_(po[q]);
    c[q] = af[q] * pc[q] + ai[q] * ag[q];
    h[q] = ao[q] * tanhf(c[q]);
  }
  // the pre-activations stay in G: the backward re-applies the four cheap activations instead of this kernel writing them
  *reinterpret_cast<float4*>(Cs + o) = make_float4(c[0], c[1], c[2], c[3]);
  const float4 hv = make_float4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<float4*>(Hs + o) = hv;
  split_store4(h_p, lo_off, o, hv);
  if (out_seq) *reinterpret_cast<float4*>(out_seq + ((int64_t)s * T + t) * 2 * H + c4) = hv;
}

// ---- out = drop([h_f(T-1) | h_r(0)]);  the same kernel turns g_out into the initial dh (mask applied, dc = 0) -------
__global__ __launch_bounds__(256) void k_clip_out(const float* __restrict__ src_f, const float* __restrict__ src_r, float* __restrict__ out,
                                                  float* zero, float* zero2, uint32_t key, int S, int H) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)S * 2 * H) return;
  const int c = (int)(i % (2 * H));
  const float v = src_f ? (c < H ? src_f : src_r)[i] : 0.f;
  out[i] = v * keep_scaled(key, (uint32_t)i, kThreshOut, kScaleOut);
  if (zero) zero[i] = 0.f;
  if (zero2) zero2[i] = 0.f;
}

// ---- one step of the backward of both directions ------------------------------------------------------------------
// in: pre-activations of (i, f, g, o) in G[t], c[t], c[t_prev], dh (this step), dc (carried);  out: pre-activation gradients
// over them (fp32 + planes), dc for the previous step.
__global__ __launch_bounds__(256) void k_clip_cell_bwd(float* __restrict__ G, uint16_t* g_p, int64_t lo_off, const float* __restrict__ Cs,
                                                       float* __restrict__ dh, float* __restrict__ dc, int S, int H, int t0,
                                                       int t1, int has_prev, const int64_t* __restrict__ lens,
                                                       float* __restrict__ carry, const float* __restrict__ g_seq, int T) {
  const int per_row = 2 * H / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)S * per_row) return;
  const int s = (int)(i / per_row), c4 = 4 * (int)(i % per_row);
  const int d = c4 / H, j = c4 - d * H;
  const int t = d ? t1 : t0, tp = d ? t1 + 1 : t0 - 1;
  const int64_t go_ = ((int64_t)t * S + s) * 8 * H + d * 4 * H + j;
  float* g = G + go_;
  const float4 vi = *reinterpret_cast<const float4*>(g), vf = *reinterpret_cast<const float4*>(g + H),
               vg = *reinterpret_cast<const float4*>(g + 2 * H), vo = *reinterpret_cast<const float4*>(g + 3 * H);
  const float4 vc = *reinterpret_cast<const float4*>(Cs + ((int64_t)t * S + s) * 2 * H + c4);
  float4 vp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_prev) vp = *reinterpret_cast<const float4*>(Cs + ((int64_t)tp * S + s) * 2 * H + c4);
  const int64_t so = (int64_t)s * 2 * H + c4;
  float4 vdh = *reinterpret_cast<const float4*>(dh + so);
  *reinterpret_cast<float4*>(dh + so) = make_float4(0.f, 0.f, 0.f, 0.f);   // the next contraction ADDS dG Whh (it may split K)
  const float4 vdc = *reinterpret_cast<const float4*>(dc + so);
  if (lens) {                              // variable lengths: add what the steps this sequence sat out passed down
    const float4 cr = *reinterpret_cast<const float4*>(carry + so);
    vdh.x += cr.x; vdh.y += cr.y; vdh.z += cr.z; vdh.w += cr.w;
    if (t >= (int)lens[s]) {               // inactive step: no gate gradients, (dh, dc) pass through unchanged
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(g) = z; *reinterpret_cast<float4*>(g + H) = z;
      *reinterpret_cast<float4*>(g + 2 * H) = z; *reinterpret_cast<float4*>(g + 3 * H) = z;
      split_store4(g_p, lo_off, go_, z); split_store4(g_p, lo_off, go_ + H, z);
      split_store4(g_p, lo_off, go_ + 2 * H, z); split_store4(g_p, lo_off, go_ + 3 * H, z);
      *reinterpret_cast<float4*>(carry + so) = vdh;
      return;
    }
    *reinterpret_cast<float4*>(carry + so) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (g_seq) {
    const float4 gs = *reinterpret_cast<const float4*>(g_seq + ((int64_t)s * T + t) * 2 * H + c4);
    vdh.x += gs.x; vdh.y += gs.y; vdh.z += gs.z; vdh.w += gs.w;
  }
  const float ai[4] = {sigmoidf_(vi.x), sigmoidf_(vi.y), sigmoidf_(vi.z), sigmoidf_(vi.w)},
              af[4] = {sigmoidf_(vf.x), sigmoidf_(vf.y), sigmoidf_(vf.z), sigmoidf_(vf.w)},
              ag[4] = {tanhf(vg.x), tanhf(vg.y), tanhf(vg.z), tanhf(vg.w)},
              ao[4] = {sigmoidf_(vo.x), sigmoidf_(vo.y), sigmoidf_(vo.z), sigmoidf_(vo.w)}, c[4] = {vc.x, vc.y, vc.z, vc.w},
              cp[4] = {vp.x, vp.y, vp.z, vp.w}, gh[4] = {vdh.x, vdh.y, vdh.z, vdh.w}, gc[4] = {vdc.x, vdc.y, vdc.z, vdc.w};
  float di[4], df[4], dg[4], dgo[4], dcp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float tc = tanhf(c[q]);
    const float dct = gc[q] + gh[q] * ao[q] * (1.f - tc * tc);
    dgo[q] = gh[q] * tc * ao[q] * (1.f - ao[q]);
    di[q] = dct * ag[q] * ai[q] * (1.f - ai[q]);
    dg[q] = dct * ai[q] * (1.f - ag[q] * ag[q]);
    df[q] = dct * cp[q] * af[q] * (1.f - af[q]);
    dcp[q] = dct * af[q];
  }
  const float4 wi = make_float4(di[0], di[1], di[2], di[3]), wf = make_float4(df[0], df[1], df[2], df[3]),
               wg = make_float4(dg[0], dg[1], dg[2], dg[3]), wo = make_float4(dgo[0], dgo[1], dgo[2], dgo[3]);
  *reinterpret_cast<float4*>(g) = wi;
  *reinterpret_cast<float4*>(g + H) = wf;
  *reinterpret_cast<float4*>(g + 2 * H) = wg;
  *reinterpret_cast<float4*>(g + 3 * H) = wo;
  split_store4(g_p, lo_off, go_, wi);
  split_store4(g_p, lo_off, go_ + H, wf);
  split_store4(g_p, lo_off, go_ + 2 * H, wg);
  split_store4(g_p, lo_off, go_ + 3 * H, wo);
  *reinterpret_cast<float4*>(dc + so) = make_float4(dcp[0], dcp[1], dcp[2], dcp[3]);
}

// ---- g_x[s][t][:] = dE[t*S + s][:] (through tanh and the input dropout when the prologue ran here) ---------------
__global__ __launch_bounds__(256) void k_clip_gx(const float* __restrict__ de, const float* __restrict__ e, float* __restrict__ gx,
                                                 uint32_t key, int S, int T, int V, int embed) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= (int64_t)S * T * V / 4) return;
  const int64_t idx = 4 * i4;
  const int v = (int)(idx % V);
  const int64_t st = idx / V;
  const int t = (int)(st % T);
  const int64_t s = st / T;
  const int64_t src = ((int64_t)t * S + s) * V + v;
  float4 g = *reinterpret_cast<const float4*>(de + src);
  if (embed) {
    const float4 a = *reinterpret_cast<const float4*>(e + src);
    g.x *= (1.f - a.x * a.x) * keep_scaled(key, (uint32_t)idx, kThreshIn, kScaleIn);
    g.y *= (1.f - a.y * a.y) * keep_scaled(key, (uint32_t)idx + 1, kThreshIn, kScaleIn);
    g.z *= (1.f - a.z * a.z) * keep_scaled(key, (uint32_t)idx + 2, kThreshIn, kScaleIn);
    g.w *= (1.f - a.w * a.w) * keep_scaled(key, (uint32_t)idx + 3, kThreshIn, kScaleIn);
  }
  reinterpret_cast<float4*>(gx)[i4] = g;
}

__global__ __launch_bounds__(256) void k_clip_keep_mask(uint32_t key, int64_t n, float* keep, uint32_t thresh) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = drop_keep_t(key, (uint32_t)i, thresh) ? 1.0f : 0.0f;
}

GemmOperand opnd(const float* p, int64_t ld, const uint16_t* hi, int64_t lo_off) {
  GemmOperand o = operand(p, ld);
  o.h0 = hi; o.h1 = hi; o.lo_off = lo_off;
  return o;
}
GemmProblem problem(const GemmOperand& A, const GemmOperand& B, float* C, int64_t ldc, const float* bias, int M, int N, int K,
                    int accumulate) {
  GemmProblem p{};
  p.A = A; p.B = B; p.C = C; p.ldc = ldc; p.bias = bias; p.M = M; p.N = N; p.K = K; p.kseg = K; p.accumulate = accumulate;
  return p;
}
inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

// planes of [wih | whh] (contiguous in the parameter buffer)
int split_weights(hipStream_t st, const float* params, const ClipParams& P, uint16_t* w_p, int64_t n) {
  SplitBatch sb{};
  sb.j[0] = SplitJob{params + P.wih, w_p, n};
  sb.n = 1; sb.lo_off = n;
  return split_planes(st, sb);
}

}  // namespace

extern "C" {

int dvgr_clip_check_dims(const dvgr_clip_dims* d) { return check_clip(d); }
int64_t dvgr_clip_param_count(const dvgr_clip_dims* d) { return (d && !check_clip(d)) ? clip_params(*d).total : 0; }
int dvgr_clip_param_offset(const dvgr_clip_dims* d, int kind, int64_t* offset, int64_t* rows, int64_t* cols) {
  CLIP_TRY(check_clip(d));
  if (!offset || !rows || !cols) return DVGR_E_NULL;
  const ClipParams p = clip_params(*d);
  const int64_t H = d->H, V = d->V;
  switch (kind) {
    case DVGR_CP_WIH_F: *offset = p.wih; *rows = 4 * H; *cols = V; break;
    case DVGR_CP_WIH_R: *offset = p.wih + 4 * H * V; *rows = 4 * H; *cols = V; break;
    case DVGR_CP_WHH_F: *offset = p.whh; *rows = 4 * H; *cols = H; break;
    case DVGR_CP_WHH_R: *offset = p.whh + 4 * H * H; *rows = 4 * H; *cols = H; break;
    case DVGR_CP_BIH_F: *offset = p.bih; *rows = 1; *cols = 4 * H; break;
    case DVGR_CP_BIH_R: *offset = p.bih + 4 * H; *rows = 1; *cols = 4 * H; break;
    case DVGR_CP_BHH_F: *offset = p.bhh; *rows = 1; *cols = 4 * H; break;
    case DVGR_CP_BHH_R: *offset = p.bhh + 4 * H; *rows = 1; *cols = 4 * H; break;
    default: return DVGR_E_KIND;
  }
  return 0;
}
size_t dvgr_clip_saved_bytes(const dvgr_clip_dims* d) { return (d && !check_clip(d)) ? (size_t)clip_saved(*d).total * 4 : 0; }
size_t dvgr_clip_workspace_bytes(const dvgr_clip_dims* d) { return (d && !check_clip(d)) ? (size_t)clip_ws(*d).total * 4 : 0; }

int dvgr_lstm_forward(const dvgr_clip_dims* dp, const float* params, const dvgr_lstm_io* io, int flags, void* saved_v,
                      size_t saved_bytes, void* ws_v, size_t ws_bytes, int train, uint64_t seed, void* stream) {
  CLIP_TRY(check_clip(dp));
  if (!params || !io || !io->x || (!io->out_final && !io->out_seq) || !saved_v || !ws_v) return DVGR_E_NULL;
  if (saved_bytes < dvgr_clip_saved_bytes(dp) || ws_bytes < dvgr_clip_workspace_bytes(dp)) return DVGR_E_WORKSPACE;
  if (!al16(params) || !al16(io->x) || !al16(io->out_final) || !al16(io->out_seq) || !al16(saved_v) || !al16(ws_v)) return DVGR_E_ALIGN;
  const dvgr_clip_dims d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const ClipParams P = clip_params(d);
  const ClipSaved SV = clip_saved(d);
  const ClipWs W = clip_ws(d);
  float* sv = (float*)saved_v;
  float* ws = (float*)ws_v;
  const int S = d.S, T = d.T, V = d.V, H = d.H, prec = d.dtype;
  const int64_t TS = (int64_t)T * S;
  const int64_t nW = 8ll * H * V + 8ll * H * H;
  uint16_t* w_p = (uint16_t*)(ws + W.w_p);
  uint16_t* e_p = (uint16_t*)(sv + SV.e_p);
  uint16_t* h_p = (uint16_t*)(sv + SV.h_p);
  const int64_t e_lo = TS * V, h_lo = TS * 2 * H;
  const int embed = (flags & DVGR_LSTM_EMBED_INPUT) ? 1 : 0;

  CLIP_TRY(split_weights(st, params, P, w_p, nW));
  DVGR_LAUNCH(k_clip_bias, dim3(blocks_for(8 * H)), dim3(256), 0, st, params + P.bih, params + P.bhh, ws + W.bias, 8 * H);
  CLIP_LAUNCH_OK();
  DVGR_LAUNCH(k_clip_embed, dim3(blocks_for(TS * V / 4)), dim3(256), 0, st, io->x, sv + SV.e, e_p, e_lo,
                     clip_key(train && embed, seed, 0), S, T, V, embed);
  CLIP_LAUNCH_OK();
  {  // input contraction of every step and both directions
    GemmBatch b{};
    b.p[b.n++] = problem(opnd(sv + SV.e, V, e_p, e_lo), opnd(params + P.wih, V, w_p, nW), sv + SV.g, 8 * H, ws + W.bias,
                         (int)TS, 8 * H, V, 0);
    CLIP_TRY(gemm_launch(st, KC, KC, b, TAG_GENERIC, prec));
  }
  for (int k = 0; k < T; ++k) {
    const int t[2] = {k, T - 1 - k}, tp[2] = {k - 1, T - k};
    if (k > 0) {
      GemmBatch b{};
      for (int dir = 0; dir < 2; ++dir) {
        const int64_t ho = (int64_t)tp[dir] * S * 2 * H + dir * H;
        const int64_t wo = 8ll * H * V + (int64_t)dir * 4 * H * H;       // offset of Whh[dir] inside [wih | whh]
        b.p[b.n++] = problem(opnd(sv + SV.h + ho, 2 * H, h_p + ho, h_lo), opnd(params + P.wih + wo, H, w_p + wo, nW),
                             sv + SV.g + (int64_t)t[dir] * S * 8 * H + dir * 4 * H, 8 * H, nullptr, S, 4 * H, H, 1);
      }
      CLIP_TRY(gemm_launch(st, KC, KC, b, TAG_GENERIC, prec));
    }
    DVGR_LAUNCH(k_clip_cell_fwd, dim3(blocks_for((int64_t)S * 2 * H / 4)), dim3(256), 0, st, sv + SV.g, sv + SV.c, sv + SV.h,
                       h_p, h_lo, S, H, t[0], t[1], k > 0 ? 1 : 0, io->lens, io->out_seq, T);
    CLIP_LAUNCH_OK();
  }
  if (io->out_final) {
    DVGR_LAUNCH(k_clip_out, dim3(blocks_for((int64_t)S * 2 * H)), dim3(256), 0, st, (const float*)(sv + SV.h + (int64_t)(T - 1) * S * 2 * H),
                       (const float*)(sv + SV.h), io->out_final, (float*)nullptr, (float*)nullptr,
                       clip_key(train && (flags & DVGR_LSTM_DROP_FINAL), seed, 1), S, H);
    CLIP_LAUNCH_OK();
  }
  return 0;
}

int dvgr_lstm_backward(const dvgr_clip_dims* dp, const float* params, const dvgr_lstm_io* io, const dvgr_lstm_grads* gr, int flags,
                       void* saved_v, size_t saved_bytes, void* ws_v, size_t ws_bytes, int train, uint64_t seed, void* stream) {
  CLIP_TRY(check_clip(dp));
  if (!params || !io || !gr || !gr->gparams || (!gr->g_final && !gr->g_seq) || !saved_v || !ws_v) return DVGR_E_NULL;
  if (saved_bytes < dvgr_clip_saved_bytes(dp) || ws_bytes < dvgr_clip_workspace_bytes(dp)) return DVGR_E_WORKSPACE;
  if (!al16(params) || !al16(gr->g_final) || !al16(gr->g_seq) || !al16(gr->g_x) || !al16(gr->gparams) || !al16(saved_v) || !al16(ws_v))
    return DVGR_E_ALIGN;
  const dvgr_clip_dims d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const ClipParams P = clip_params(d);
  const ClipSaved SV = clip_saved(d);
  const ClipWs W = clip_ws(d);
  float* sv = (float*)saved_v;
  float* ws = (float*)ws_v;
  float* gparams = gr->gparams;
  const int S = d.S, T = d.T, V = d.V, H = d.H, prec = d.dtype;
  const int64_t TS = (int64_t)T * S;
  const int64_t nW = 8ll * H * V + 8ll * H * H;
  uint16_t* w_p = (uint16_t*)(ws + W.w_p);
  uint16_t* e_p = (uint16_t*)(sv + SV.e_p);
  uint16_t* h_p = (uint16_t*)(sv + SV.h_p);
  uint16_t* g_p = (uint16_t*)(ws + W.g_p);
  const int64_t e_lo = TS * V, h_lo = TS * 2 * H, g_lo = TS * 8 * H;
  const int embed = (flags & DVGR_LSTM_EMBED_INPUT) ? 1 : 0;

  if (hipMemsetAsync(gparams, 0, (size_t)P.total * 4, st) != hipSuccess) return (int)hipGetLastError();
  CLIP_TRY(split_weights(st, params, P, w_p, nW));
  // dh of the last step = g_final through the output dropout (both halves: the forward direction's last step is frame
  // T-1, the reverse direction's is frame 0, and both are step T-1); dc = 0; nothing carried yet
  DVGR_LAUNCH(k_clip_out, dim3(blocks_for((int64_t)S * 2 * H)), dim3(256), 0, st, gr->g_final, gr->g_final, ws + W.dh, ws + W.dc,
                     io->lens ? ws + W.carry : (float*)nullptr, clip_key(train && (flags & DVGR_LSTM_DROP_FINAL), seed, 1), S, H);
  CLIP_LAUNCH_OK();
  for (int k = T - 1; k >= 0; --k) {
    const int t[2] = {k, T - 1 - k};
    DVGR_LAUNCH(k_clip_cell_bwd, dim3(blocks_for((int64_t)S * 2 * H / 4)), dim3(256), 0, st, sv + SV.g, g_p, g_lo, sv + SV.c,
                       ws + W.dh, ws + W.dc, S, H, t[0], t[1], k > 0 ? 1 : 0, io->lens, ws + W.carry, gr->g_seq, T);
    CLIP_LAUNCH_OK();
    if (k > 0) {     // dh of the previous step = dG[t] Whh
      GemmBatch b{};
      for (int dir = 0; dir < 2; ++dir) {
        const int64_t go = (int64_t)t[dir] * S * 8 * H + dir * 4 * H;
        const int64_t wo = 8ll * H * V + (int64_t)dir * 4 * H * H;
        b.p[b.n] = problem(opnd(sv + SV.g + go, 8 * H, g_p + go, g_lo), opnd(params + P.wih + wo, H, w_p + wo, nW),
                           ws + W.dh + dir * H, 2 * H, nullptr, S, H, 4 * H, 1);
        b.p[b.n++].atomic_ok = 1;          // dh was cleared by the cell kernel above
      }
      CLIP_TRY(gemm_launch(st, KC, IC, b, TAG_GENERIC, prec));
    }
  }
  {  // dWih[both directions] = dG^T e, reduction over all T * S rows
    GemmBatch b{};
    b.p[b.n] = problem(opnd(sv + SV.g, 8 * H, g_p, g_lo), opnd(sv + SV.e, V, e_p, e_lo), gparams + P.wih, V, nullptr, 8 * H, V,
                       (int)TS, 1);
    b.p[b.n++].atomic_ok = 1;            // gparams was cleared above: the launcher may split the T * S reduction
    CLIP_TRY(gemm_launch(st, IC, IC, b, TAG_GENERIC, prec));
  }
  if (T > 1) {  // dWhh[dir] = sum over steps of dG[t]^T h_prev: forward pairs frame t with t-1, reverse pairs t with t+1
    GemmBatch b{};
    for (int dir = 0; dir < 2; ++dir) {
      const int64_t go = (dir == 0 ? (int64_t)S * 8 * H : 0) + dir * 4 * H;
      const int64_t ho = (dir == 0 ? 0 : (int64_t)S * 2 * H) + dir * H;
      b.p[b.n] = problem(opnd(sv + SV.g + go, 8 * H, g_p + go, g_lo), opnd(sv + SV.h + ho, 2 * H, h_p + ho, h_lo),
                         gparams + P.whh + (int64_t)dir * 4 * H * H, H, nullptr, 4 * H, H, (int)((int64_t)(T - 1) * S), 1);
      b.p[b.n++].atomic_ok = 1;
    }
    CLIP_TRY(gemm_launch(st, IC, IC, b, TAG_GENERIC, prec));
  }
  {  // db_ih = db_hh = column sums of dG
    ColsumBatch cb{};
    ColsumJob& j = cb.j[cb.n++];
    j.M0 = sv + SV.g; j.M1 = nullptr; j.w0 = nullptr; j.w1 = nullptr; j.rows0 = (int)TS; j.rows1 = 0; j.cols = 8 * H; j.ld = 8 * H;
    j.out = gparams + P.bih;
    int chunks = (int)(TS / 256);
    cb.chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    CLIP_TRY(colsum(st, cb));
    if (hipMemcpyAsync(gparams + P.bhh, gparams + P.bih, (size_t)8 * H * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return (int)hipGetLastError();
  }
  if (gr->g_x) {  // gradient of the input: dE = dG [Wih_f; Wih_r], back to batch-first (and through the prologue)
    if (hipMemsetAsync(ws + W.de, 0, (size_t)TS * V * 4, st) != hipSuccess) return (int)hipGetLastError();
    GemmBatch b{};
    b.p[b.n] = problem(opnd(sv + SV.g, 8 * H, g_p, g_lo), opnd(params + P.wih, V, w_p, nW), ws + W.de, V, nullptr, (int)TS, V, 8 * H, 1);
    b.p[b.n++].atomic_ok = 1;
    CLIP_TRY(gemm_launch(st, KC, IC, b, TAG_GENERIC, prec));
    DVGR_LAUNCH(k_clip_gx, dim3(blocks_for(TS * V / 4)), dim3(256), 0, st, (const float*)(ws + W.de), (const float*)(sv + SV.e), gr->g_x,
                       clip_key(train && embed, seed, 0), S, T, V, embed);
    CLIP_LAUNCH_OK();
  }
  return 0;
}

int dvgr_clip_forward(const dvgr_clip_dims* dp, const float* params, const float* x, float* out, void* saved_v, size_t saved_bytes,
                      void* ws_v, size_t ws_bytes, int train, uint64_t seed, void* stream) {
  if (!x || !out) return DVGR_E_NULL;
  dvgr_lstm_io io{};
  io.x = x; io.out_final = out;
  return dvgr_lstm_forward(dp, params, &io, DVGR_LSTM_EMBED_INPUT | DVGR_LSTM_DROP_FINAL, saved_v, saved_bytes, ws_v, ws_bytes, train, seed,
                           stream);
}

int dvgr_clip_backward(const dvgr_clip_dims* dp, const float* params, const float* g_out, float* gparams, void* saved_v,
                       size_t saved_bytes, void* ws_v, size_t ws_bytes, int train, uint64_t seed, void* stream) {
  if (!g_out || !gparams) return DVGR_E_NULL;
  dvgr_lstm_io io{};
  dvgr_lstm_grads gr{};
  gr.g_final = g_out; gr.gparams = gparams;
  return dvgr_lstm_backward(dp, params, &io, &gr, DVGR_LSTM_EMBED_INPUT | DVGR_LSTM_DROP_FINAL, saved_v, saved_bytes, ws_v, ws_bytes, train,
                            seed, stream);
}

int dvgr_clip_dbg_keep_mask(uint64_t seed, int site, int64_t n, float* keep, void* stream) {
  if (!keep) return DVGR_E_NULL;
  if (site < 0 || site > 1 || n < 0) return DVGR_E_DIMS;
  if (n == 0) return 0;
  DVGR_LAUNCH(k_clip_keep_mask, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, site_key(seed, kClipSiteBase + (uint32_t)site),
                     n, keep, site == 0 ? kThreshIn : kThreshOut);
  return (int)hipGetLastError();
}

}  // extern "C"
