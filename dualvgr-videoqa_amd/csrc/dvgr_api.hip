// dvgr_api.hip -- C ABI (include/dvgr.h): layouts, argument checks and the stage schedules of the
// forward and backward passes.  Host code only; device work is enqueued on the caller's stream and, for what is off
// the critical path, on an internal side stream that forks from and joins back into it inside each call.
#include <stdlib.h>
#include <string.h>

#include "dvgr_gemm.h"
#include "dvgr_side.h"
#include "dvgr_stages.h"
#include "../../include/dvgr_comm.h"

using namespace dvgr;

namespace dvgr {
#ifdef DVGR_TIMING_AIDS
int fwd_sf_phase_launch(hipStream_t stream, const GemmBatch& b, const SfCombineArgs& sf, int prec);   // experiment, dvgr_gemm.hip
#endif
std::atomic<long long> g_launch_count{0};
size_t gat_lds_bytes_max(int N, int F);
size_t aux_ws_floats(const dvgr_dims& d);
int aux_losses(hipStream_t st, const dvgr_dims& d, const dvgr_io* io, float alpha, float beta, float dep_scale,
               float* loss3, float* g_vis, float* const g[4], float* ws, bool zero_loss3 = true, int first = 0, int count = -1);
int sum_and_ones(hipStream_t st, const float* x, float* g_ones, int64_t n, float* out);
}

#define DVGR_TRY(expr)                 \
  do {                                 \
    const int _e = (expr);             \
    if (_e != 0) return _e;            \
  } while (0)

namespace {

struct FwdWs { int64_t Y, R, wpart, total; };
FwdWs fwd_ws(const dvgr_dims& d) {
  FwdWs w{};
  int64_t o = 0;
  w.Y = o; o += round64((int64_t)d.L * d.B * d.T * d.D);
  w.R = o; o += round64((int64_t)d.L * d.B * d.T);
  w.wpart = o; o += 2 * round64((int64_t)2 * (d.D / 32) * d.B * d.N);     // [modality][view][D / 32][B N]: S4's partial logits for S5
  w.total = o;
  return w;
}

// Buffers written by the critical path of layer l and read by the side stream (weight / bias gradients) exist
// twice, indexed by l & 1: the critical path may run one layer ahead of the side stream.  The query branch runs once,
// after the last layer, for ALL layers: its inputs (d p of every layer) and intermediates are kept per layer.
struct BwdWs {
  int64_t dpo, dzf, dp0, dp1, dz[2], dxtmp[2], dspec[2];
  int64_t dw[2][2], dpre[2][2], dWh[2][2], ds_part[2][2], da_part[2][2], dc_part[2][2];   // [set][modality]
  int64_t dp_cat, dqbar, dr, dy, qdots;                                                    // [L] each
  int64_t p1t;            // [L][2] plane-only slots of D x D: P1^T of every (layer, modality) for the fused GAT-view backward
  int64_t total;
};
BwdWs bwd_ws(const dvgr_dims& d) {
  BwdWs w{};
  const int64_t BN = (int64_t)d.B * d.N, D = d.D, F = d.D / 4, L = d.L;
  int64_t o = 0;
  w.dpo = o; o += round64(BN * D);
  w.dzf = o; o += round64(BN * kMfbPooled);
  w.dp0 = o; o += round64(BN * kMfbHidden);
  w.dp1 = o; o += round64(BN * kMfbHidden);
  for (int m = 0; m < 2; ++m) { w.dz[m] = o; o += round64(2 * BN * D); }
  for (int m = 0; m < 2; ++m) { w.dxtmp[m] = o; o += round64(BN * D); }
  for (int m = 0; m < 2; ++m) { w.dspec[m] = o; o += d.G > 1 ? round64(BN * D) : 0; }   // graph_layers > 1: next graph layer's input grad
  for (int s = 0; s < 2; ++s) {
    for (int m = 0; m < 2; ++m) {
      w.dw[s][m] = o;      o += round64(2 * BN);
      w.dpre[s][m] = o;    o += round64(2 * BN * D);
      w.dWh[s][m] = o;     o += round64(BN * kHeads * F);
      w.ds_part[s][m] = o; o += round64((int64_t)d.B * kHeads * d.N * (d.G > 1 ? d.G : 1));   // one block per graph layer
      w.da_part[s][m] = o; o += round64((int64_t)d.B * kHeads * 2 * F);
      w.dc_part[s][m] = o; o += round64((int64_t)d.B * kHeads);
    }
  }
  w.dp_cat = o; o += round64(L * d.B * 2 * D);
  w.dqbar = o;  o += round64(L * d.B * d.Wd);
  w.dr = o;     o += round64(L * d.B * d.T);
  w.dy = o;     o += round64(L * d.B * d.T * D);
  w.qdots = o;  o += round64(2 * L * d.B * d.T);
  w.p1t = o;    o += 2 * L * round64(D * D);
  w.total = o;
  return w;
}

int64_t ws_floats(const dvgr_dims& d) {
  dvgr_dims dl = d;                      // the losses run over the L * G list entries
  dl.L = d.L * (d.G > 1 ? d.G : 1);
  const int64_t f = fwd_ws(d).total + round64((int64_t)aux_ws_floats(dl)), b = bwd_ws(d).total;   // forward may carry the loss
  return f > b ? f : b;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// bf16 hi/lo planes of GEMM operands (DVGR_BF16 / DVGR_BF16X3).  Every fp32 buffer the library owns
// (parameters, saved, workspace) has a parallel planes region with the same element offsets; tensors owned
// by the caller (inputs, list outputs) get plane-only slots inside `saved`.  attach() resolves the planes of
// both operands of a contraction from their fp32 pointers; a problem whose planes are not found simply runs
// on the converting kernels.
struct PlaneMap {
  struct Range { const float* f; int64_t n; const uint16_t* hi; int64_t lo_off; };
  Range r[64];
  int n = 0;
  bool on = false;
  void add(const float* f, int64_t cnt, const uint16_t* hi, int64_t lo_off) {
    if (on && hi && n < 64) r[n++] = Range{f, cnt, hi, lo_off};
  }
  const uint16_t* find(const float* p, int64_t* lo_off) const {
    for (int i = 0; i < n; ++i)
      if (p >= r[i].f && p < r[i].f + r[i].n) { *lo_off = r[i].lo_off; return r[i].hi + (p - r[i].f); }
    return nullptr;
  }
  void attach1(GemmOperand& o) const {
    int64_t l0 = 0, l1 = 0;
    const uint16_t* h0 = find(o.p0, &l0);
    const uint16_t* h1 = find(o.p1, &l1);
    if (h0 && h1 && l0 == l1) { o.h0 = h0; o.h1 = h1; o.lo_off = l0; }
  }
  void attach(GemmProblem& p) const {
    if (!on) return;
    attach1(p.A);
    attach1(p.B);
  }
  uint16_t* out(const float* p, int64_t* lo_off) const {     // planes of an OUTPUT tensor (nullable)
    return on ? const_cast<uint16_t*>(find(p, lo_off)) : nullptr;
  }
};

// Tensors whose ONLY readers are contractions (the dropped copies xd; backward: dpre, dWh) exist as bf16 planes alone when
// every one of those contractions runs on the planes kernels: bf16 modes, B N a multiple of the K-tile and above the
// batch-sized kernels' range (dvgr_small.hip reads fp32 operands for reductions <= 256 rows).  Forward and backward
// derive the same answer from the dimensions.
inline bool lean_planes(bool planes_on, int BN, int G = 1) { return planes_on && (BN % 32) == 0 && BN > 256 && G == 1; }

// C = act(A[M,K] . B[N,K]^T + bias)                      (forward Linear)
GemmProblem nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
               const float* bias, int M, int N, int K, int act, uint32_t drop_a = 0) {
  GemmProblem p{};
  p.A = operand(A, lda, drop_a); p.B = operand(B, ldb);
  p.C = C; p.ldc = ldc; p.bias = bias; p.M = M; p.N = N; p.K = K; p.kseg = K; p.act = act; p.layout = LAYOUT_NT;
  return p;
}
// C (+)= A[M,K] . B[K,N]                                  (backward data)
GemmProblem nn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
               int M, int N, int K, int accumulate, uint32_t drop_c = 0) {
  GemmProblem p{};
  p.A = operand(A, lda); p.B = operand(B, ldb);
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.kseg = K; p.accumulate = accumulate; p.drop_key_c = drop_c;
  p.layout = LAYOUT_NN;
  return p;
}
// C = A[K,M]^T . B[K,N]                                   (backward weight)
GemmProblem tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
               int M, int N, int K, uint32_t drop_b = 0) {
  GemmProblem p{};
  p.A = operand(A, lda); p.B = operand(B, ldb, drop_b);
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.kseg = K; p.layout = LAYOUT_TN;
  p.atomic_ok = 1;   // a long reduction may be split over workgroups; the matrices of gparams are not pre-zeroed,
  p.zero_on_split = 1;   //   so such a launch zeroes its outputs first
  return p;
}

struct Keys {          // dropout keys of one layer (0 in eval mode)
  uint32_t x[2][2], out[2][2], alpha[2][8];
};
Keys layer_keys(int train, uint64_t seed, int l) {
  Keys k{};
  if (!train) return k;
  for (int m = 0; m < 2; ++m)
    for (int g = 0; g < 2; ++g) {
      k.x[m][g] = site_key(seed, drop_site(l, m, g, 0));
      k.out[m][g] = site_key(seed, drop_site(l, m, g, 5));
      for (int h = 0; h < 4; ++h) k.alpha[m][g * 4 + h] = site_key(seed, drop_site(l, m, g, 1 + h));
    }
  return k;
}

// Brackets one launch site with the caller's hipEvents (dvgr_opts.probe_*).
struct Probe {
  const dvgr_opts& o;
  hipStream_t st;
  bool hit(int site, int layer) const { return o.probe_site == site && o.probe_layer == layer && o.probe_begin && o.probe_end; }
  int begin(int site, int layer) const { return hit(site, layer) ? (int)hipEventRecord((hipEvent_t)o.probe_begin, st) : 0; }
  int end(int site, int layer) const { return hit(site, layer) ? (int)hipEventRecord((hipEvent_t)o.probe_end, st) : 0; }
};

PlaneMap make_plane_map(const dvgr_dims& d, const ParamLayout& P, const SavedLayout& S, const float* params,
                        const void* params_lp, const dvgr_io* io, const float* saved, const float* ws) {
  PlaneMap pm;
  pm.on = d.dtype != DVGR_F32 && params_lp != nullptr;
  if (!pm.on) return pm;
  const int64_t BND = (int64_t)d.B * d.N * d.D, wsf = ws_floats(d);
  const uint16_t* sv_hi = reinterpret_cast<const uint16_t*>(saved + S.total);
  pm.add(params, P.total, reinterpret_cast<const uint16_t*>(params_lp), P.total);
  pm.add(saved, S.total, sv_hi, S.total);
  pm.add(ws, wsf, reinterpret_cast<const uint16_t*>(ws + wsf), wsf);
  pm.add(io->A, BND, sv_hi + S.in_a, S.total);
  pm.add(io->M, BND, sv_hi + S.in_m, S.total);
  pm.add(io->dq, (int64_t)d.B * d.T * d.D, sv_hi + S.in_dq, S.total);
  for (int e = 0; e < d.L * d.G && e < 12; ++e) {      // list entry e = unit layer e / G, graph layer e % G
    const int64_t m0 = S.layer_stride * (e / d.G) + S.mod_base + S.g_stride * (e % d.G), m1 = m0 + S.mod_stride;
    pm.add(io->com_app + e * BND, BND, sv_hi + m0 + S.zc, S.total);
    pm.add(io->com_mot + e * BND, BND, sv_hi + m1 + S.zc, S.total);
    pm.add(io->aq_fusion + e * BND, BND, sv_hi + m0 + S.zs, S.total);
    pm.add(io->mq_fusion + e * BND, BND, sv_hi + m1 + S.zs, S.total);
  }
  return pm;
}

int check_common(const dvgr_dims* d, const void* params, const dvgr_io* io, const void* saved, size_t saved_bytes,
                 const void* ws, size_t ws_bytes) {
  if (!d) return DVGR_E_NULL;
  DVGR_TRY(dvgr_check_dims(d));
  if (!params || !io || !saved || !ws) return DVGR_E_NULL;
  if (!io->A || !io->M || !io->dq || !io->words || !io->qlen || !io->vis || !io->aq_embed || !io->mq_embed ||
      !io->com_app || !io->com_mot || !io->aq_fusion || !io->mq_fusion)
    return DVGR_E_NULL;
  if (saved_bytes < dvgr_saved_bytes(d) || ws_bytes < dvgr_workspace_bytes(d)) return DVGR_E_WORKSPACE;
  const void* ptrs[] = {params, io->A, io->M, io->dq, io->words, io->vis, io->aq_embed, io->mq_embed,
                        io->com_app, io->com_mot, io->aq_fusion, io->mq_fusion, saved, ws};
  for (const void* p : ptrs)
    if (!aligned16(p)) return DVGR_E_ALIGN;
  return 0;
}

}  // namespace

extern "C" {

const char* dvgr_version(void) { return "dvgr-hip 0.1.0 gfx950"; }
int dvgr_streams_place(void* caller_stream) {
  int dev = 0;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= 16) return DVGR_E_DIMS;
  return shared_side_stream(dev, 1, true, (hipStream_t)caller_stream) ? 0 : (int)hipErrorOutOfMemory;   // side, then query
}
int dvgr_streams_init(void) { return dvgr_streams_place(nullptr); }

int dvgr_check_dims(const dvgr_dims* d) {
  if (!d) return DVGR_E_NULL;
  if (d->B < 1 || d->N < 1 || d->T < 1 || d->L < 1) return DVGR_E_DIMS;
  if (d->G < 1 || d->G > 4 || d->L * d->G > 48) return DVGR_E_DIMS;   // graph_layers > 1: see dvgr_unit_forward
  if (d->D < 64 || d->D % 64) return DVGR_E_DIMS;
  if (d->Wd < 4 || d->Wd % 4) return DVGR_E_DIMS;
  if (d->N > 64 || d->T > 256) return DVGR_E_DIMS;
  if (d->dtype != DVGR_F32 && d->dtype != DVGR_BF16 && d->dtype != DVGR_BF16X3) return DVGR_E_DIMS;
  if (gat_lds_bytes_max(d->N, d->D / 4) > 160 * 1024) return DVGR_E_DIMS;
  if ((int64_t)d->B * d->N * d->D * 2 >= (1ll << 32)) return DVGR_E_DIMS;   // 32-bit dropout counters
  return 0;
}

int64_t dvgr_param_count(const dvgr_dims* d) { return d ? param_layout(*d).total : 0; }

int dvgr_param_offset(const dvgr_dims* d, int kind, int layer, int mod, int64_t* offset, int64_t* rows, int64_t* cols) {
  if (!d || !offset || !rows || !cols) return DVGR_E_NULL;
  const ParamLayout p = param_layout(*d);
  const int64_t D = d->D, Wd = d->Wd, F = d->D / 4;
  if (kind < 0 || kind >= DVGR_P_KINDS) return DVGR_E_KIND;
  const bool gat_kind = kind >= DVGR_P_GAT_W && kind <= DVGR_P_GAT_C;              // `layer` = GAT module index, 0 .. L * G - 1
  const int G = d->G > 1 ? d->G : 1;
  if (kind < DVGR_P_MFB_W0 && (layer < 0 || layer >= (gat_kind ? d->L * G : d->L) || mod < 0 || mod > 1)) return DVGR_E_KIND;
  const int64_t mb = gat_kind ? gat_set_base(p, d->L, layer, mod) : p.layer_stride * layer + p.mod_stride * mod;   // layer block
  const int64_t lb = p.q_base + p.q_layer_stride * layer;                          // query block (QueryAttn / QueryPunish)
  const int64_t qmb = lb + p.q_mod_base + p.q_mod_stride * mod;
  switch (kind) {
    case DVGR_P_QA_W1: *offset = lb + p.qa_w1; *rows = D; *cols = D; break;
    case DVGR_P_QA_B1: *offset = lb + p.qa_b1; *rows = 1; *cols = D; break;
    case DVGR_P_QA_W2: *offset = lb + p.qa_w2; *rows = 1; *cols = D; break;
    case DVGR_P_QA_B2: *offset = lb + p.qa_b2; *rows = 1; *cols = 1; break;
    case DVGR_P_WQ: *offset = qmb + p.wq; *rows = D; *cols = Wd; break;
    case DVGR_P_BQ: *offset = qmb + p.bq; *rows = 1; *cols = D; break;
    case DVGR_P_GAT_W: *offset = mb + p.gat_w; *rows = kHeads * F; *cols = D; break;
    case DVGR_P_GAT_B: *offset = mb + p.gat_b; *rows = 1; *cols = kHeads * F; break;
    case DVGR_P_GAT_A: *offset = mb + p.gat_a; *rows = kHeads; *cols = 2 * F; break;
    case DVGR_P_GAT_C: *offset = mb + p.gat_c; *rows = 1; *cols = kHeads; break;
    case DVGR_P_SF_P1: *offset = mb + p.sf_p1; *rows = D; *cols = D; break;
    case DVGR_P_SF_BP: *offset = mb + p.sf_bp; *rows = 1; *cols = D; break;
    case DVGR_P_SF_P2: *offset = mb + p.sf_p2; *rows = 1; *cols = D; break;
    case DVGR_P_MFB_W0: *offset = p.mfb_w0; *rows = kMfbHidden; *cols = D; break;
    case DVGR_P_MFB_B0: *offset = p.mfb_b0; *rows = 1; *cols = kMfbHidden; break;
    case DVGR_P_MFB_W1: *offset = p.mfb_w1; *rows = kMfbHidden; *cols = D; break;
    case DVGR_P_MFB_B1: *offset = p.mfb_b1; *rows = 1; *cols = kMfbHidden; break;
    case DVGR_P_MFB_WO: *offset = p.mfb_wo; *rows = D; *cols = kMfbPooled; break;
    case DVGR_P_MFB_BO: *offset = p.mfb_bo; *rows = 1; *cols = D; break;
    default: return DVGR_E_KIND;
  }
  return 0;
}

int dvgr_param_layer_range(const dvgr_dims* d, int layer, int64_t* first, int64_t* last) {
  if (!d || !first || !last) return DVGR_E_NULL;
  const ParamLayout p = param_layout(*d);
  if (layer < 0 || layer > d->L + 2) return DVGR_E_KIND;
  if (layer == d->L + 2) { *first = p.extra_base; *last = p.total; }       // extra GAT sets (graph_layers > 1; empty otherwise)
  else if (layer == d->L + 1) { *first = p.q_base; *last = p.extra_base; } // query block of all layers
  else if (layer == d->L) { *first = p.mfb_base; *last = p.q_base; }       // MFB block
  else { *first = p.layer_stride * layer; *last = p.layer_stride * (layer + 1); }
  return 0;
}

// fp32 region + bf16 hi/lo planes region (same element offsets): 8 bytes per float slot
size_t dvgr_saved_bytes(const dvgr_dims* d) { return d ? (size_t)saved_layout(*d).total * 8 : 0; }

int dvgr_saved_offset(const dvgr_dims* d, int kind, int layer, int mod, size_t* offset_bytes) {
  if (!d || !offset_bytes) return DVGR_E_NULL;
  const SavedLayout s = saved_layout(*d);
  if (kind < 0 || kind >= DVGR_S_KINDS) return DVGR_E_KIND;
  // per-graph-layer tensors (Wh, u, v, xd): `layer` = unit layer * G + graph layer; everything else: the unit layer
  const int G = d->G > 1 ? d->G : 1;
  const bool per_g = kind == DVGR_S_WH || kind == DVGR_S_U || kind == DVGR_S_V || kind == DVGR_S_XD;
  if ((kind < DVGR_S_MFB_X0 || kind == DVGR_S_XD) && (layer < 0 || layer >= (per_g ? d->L * G : d->L) || mod < 0 || mod > 1)) return DVGR_E_KIND;
  const int unit_layer = per_g ? layer / G : layer;
  const int64_t lb = s.layer_stride * unit_layer, mb = lb + s.mod_base + s.mod_stride * mod + (per_g ? s.g_stride * (layer % G) : 0);
  int64_t o;
  switch (kind) {
    case DVGR_S_QA_D: o = lb + s.qa_d; break;
    case DVGR_S_QA_NRM: o = lb + s.qa_nrm; break;
    case DVGR_S_QA_A: o = lb + s.qa_a; break;
    case DVGR_S_QA_AH: o = lb + s.qa_ah; break;
    case DVGR_S_QA_S: o = lb + s.qa_s; break;
    case DVGR_S_QA_Q: o = lb + s.qa_q; break;
    case DVGR_S_P: o = mb + s.p; break;
    case DVGR_S_S: o = mb + s.s; break;
    case DVGR_S_X: o = mb + s.x; break;
    case DVGR_S_WH: o = mb + s.wh; break;
    case DVGR_S_U: o = mb + s.u; break;
    case DVGR_S_V: o = mb + s.v; break;
    case DVGR_S_TN: o = mb + s.tn; break;
    case DVGR_S_BETA: o = mb + s.beta; break;
    case DVGR_S_MFB_X0: o = s.mfb_x0; break;
    case DVGR_S_MFB_X1: o = s.mfb_x1; break;
    case DVGR_S_MFB_Z: o = s.mfb_z; break;
    case DVGR_S_XD: o = mb + s.xd; break;
    default: return DVGR_E_KIND;
  }
  *offset_bytes = (size_t)o * sizeof(float);
  return 0;
}

size_t dvgr_workspace_bytes(const dvgr_dims* d) {
  if (!d) return 0;
  return (size_t)ws_floats(*d) * 8;      // fp32 region + bf16 hi/lo planes region
}

int dvgr_pack_weights(const dvgr_dims* d, const float* params, void* params_lp, void* stream) {
  if (!d || !params || !params_lp) return DVGR_E_NULL;
  DVGR_TRY(dvgr_check_dims(d));
  SplitBatch sb{};
  const int64_t total = param_layout(*d).total;
  sb.lo_off = total;
  sb.j[sb.n++] = SplitJob{params, reinterpret_cast<uint16_t*>(params_lp), total};
  return split_planes((hipStream_t)stream, sb);
}

size_t dvgr_p1t_elems(const dvgr_dims* d) {
  if (!d || dvgr_check_dims(d) != 0) return 0;
  return (size_t)4 * d->L * d->D * d->D;                  // hi + lo planes of L x 2 matrices of D x D
}

// Layout of the dvgr_pack_lo8 buffer (bytes): [Wcat q: L x 2 x (8F x D)] [P1^T q: L x 2 x (D x D)] [Wcat scales: L x 2 x 8F floats]
// [P1^T scales: L x 2 x D floats]
struct Lo8Layout { int64_t wq, pq, ws, ps, total; };
static Lo8Layout lo8_layout(const dvgr_dims& d) {
  Lo8Layout y{};
  const int64_t D = d.D, HF = (int64_t)kHeads * (d.D / 4), n = 2 * (int64_t)d.L;
  y.wq = 0;
  y.pq = y.wq + n * HF * D;
  y.ws = (y.pq + n * D * D + 255) / 256 * 256;
  y.ps = y.ws + n * HF * 4;
  y.total = y.ps + n * D * 4;
  return y;
}

size_t dvgr_lo8_bytes(const dvgr_dims* d) {
  if (!d || dvgr_check_dims(d) != 0 || d->dtype != DVGR_BF16X3) return 0;
  return (size_t)lo8_layout(*d).total;
}

int dvgr_pack_lo8(const dvgr_dims* d, const void* params_lp, const void* p1t, void* lo8, void* stream) {
  if (!d || !params_lp || !p1t || !lo8) return DVGR_E_NULL;
  DVGR_TRY(dvgr_check_dims(d));
  if (d->dtype != DVGR_BF16X3) return DVGR_E_KIND;
  if (4 * d->L > kMaxLo8Jobs) return DVGR_E_DIMS;
  const ParamLayout P = param_layout(*d);
  const Lo8Layout Y = lo8_layout(*d);
  const int64_t D = d->D, HF = (int64_t)kHeads * (d->D / 4), DD = D * D;
  uint8_t* out = reinterpret_cast<uint8_t*>(lo8);
  Lo8Batch qb{};
  qb.D = d->D;
  for (int l = 0; l < d->L; ++l)
    for (int m = 0; m < 2; ++m) {
      const int64_t i = l * 2 + m;
      qb.j[qb.n++] = Lo8Job{reinterpret_cast<const uint16_t*>(params_lp) + P.total + P.layer_stride * l + P.mod_stride * m + P.gat_w,
                            out + Y.wq + i * HF * D, reinterpret_cast<float*>(out + Y.ws) + i * HF, (int)HF};
      qb.j[qb.n++] = Lo8Job{reinterpret_cast<const uint16_t*>(p1t) + 2 * (int64_t)d->L * DD + i * DD,
                            out + Y.pq + i * DD, reinterpret_cast<float*>(out + Y.ps) + i * D, (int)D};
    }
  return quantize_lo8((hipStream_t)stream, qb);
}

static bool backward_fuses(const dvgr_dims& d, int32_t flags) {
  return d.dtype != DVGR_F32 && d.G == 1 && !(flags & (DVGR_OPT_NO_FUSED_GAT | DVGR_OPT_NO_FUSED_GAT_BWD)) && d.L * 2 <= kMaxTransposeJobs &&
         gat_fused_bwd_supported(d.B, d.N, d.D, d.dtype) &&
         (gat_fused_bwd_preferred(d.B, d.N, d.dtype == DVGR_BF16X3 && (d.D % 64) == 0 &&
                                             !(flags & (DVGR_OPT_KB32 | DVGR_OPT_NO_HALF_GAT_TILES | DVGR_OPT_FUSED_BWD_READS_DZ))) ||
          (flags & DVGR_OPT_FORCE_FUSED_GAT_BWD));
}

int dvgr_backward_uses_p1t(const dvgr_dims* d, int32_t flags) {
  if (!d || dvgr_check_dims(d) != 0) return 0;
  return backward_fuses(*d, flags) ? 1 : 0;
}

int dvgr_pack_p1t(const dvgr_dims* d, const void* params_lp, void* p1t, void* stream) {
  if (!d || !params_lp || !p1t) return DVGR_E_NULL;
  DVGR_TRY(dvgr_check_dims(d));
  if (d->dtype == DVGR_F32) return DVGR_E_KIND;
  if ((d->D % 64) != 0 || 2 * d->L > kMaxTransposeJobs) return DVGR_E_DIMS;
  const ParamLayout P = param_layout(*d);
  const int64_t DD = (int64_t)d->D * d->D;
  TransposeBatch tb{};
  for (int l = 0; l < d->L; ++l)
    for (int m = 0; m < 2; ++m)
      tb.j[tb.n++] = TransposeJob{reinterpret_cast<const uint16_t*>(params_lp) + P.layer_stride * l + P.mod_stride * m + P.sf_p1,
                                  reinterpret_cast<uint16_t*>(p1t) + (int64_t)(l * 2 + m) * DD};
  tb.src_lo = P.total; tb.dst_lo = 2 * (int64_t)d->L * DD; tb.D = d->D;
  return transpose_planes((hipStream_t)stream, tb);
}

// =================================================================================================
// Forward schedule (stages S0..S6 of oracle/unit_manual.py)
// =================================================================================================
int dvgr_unit_forward(const dvgr_dims* dp, const float* params, const void* params_lp, const dvgr_io* io,
                      void* saved_v, size_t saved_bytes, void* ws_v, size_t ws_bytes, const dvgr_opts* opts,
                      void* stream) {
  DVGR_TRY(check_common(dp, params, io, saved_v, saved_bytes, ws_v, ws_bytes));
  const dvgr_dims d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const dvgr_opts o0{};
  const dvgr_opts& O = opts ? *opts : o0;
  const int train = O.train;
  const uint64_t seed = O.seed;
  const int prec = dp->dtype;
  const ParamLayout P = param_layout(d);
  const SavedLayout S = saved_layout(d);
  const FwdWs W = fwd_ws(d);
  float* saved = (float*)saved_v;
  float* ws = (float*)ws_v;
  const int B = d.B, N = d.N, T = d.T, D = d.D, Wd = d.Wd, L = d.L, F = D / 4, HF = kHeads * F;
  const int BN = B * N, BT = B * T;
  const int64_t BND = (int64_t)BN * D;
  auto lp = [&](int l) { return params + P.q_base + P.q_layer_stride * l; };                                        // query block
  auto qp = [&](int l, int m) { return params + P.q_base + P.q_layer_stride * l + P.q_mod_base + P.q_mod_stride * m; };
  auto mp = [&](int l, int m) { return params + P.layer_stride * l + P.mod_stride * m; };                          // layer block
  auto ls = [&](int l) { return saved + S.layer_stride * l; };
  auto ms = [&](int l, int m) { return saved + S.layer_stride * l + S.mod_base + S.mod_stride * m; };

  const PlaneMap PM = make_plane_map(d, P, S, params, params_lp, io, saved, ws);
  int64_t plo = 0;                       // lo-plane offset of the last PM.out() lookup
  const int G = d.G;
  const bool lean = lean_planes(PM.on, BN, G);
  bool loss3_cleared = false;            // the loss scalars were zeroed by the input split kernel (else: a memset of aux_losses)
  (void)stop_event_rode();               // (no armed stop event survives from an earlier call of this thread that failed half-way)
  SideStream* side = (O.flags & DVGR_OPT_SINGLE_STREAM) ? nullptr : side_stream(st);
  hipStream_t hs = side ? side->st : st;
  // the fork behind the input split (the other layers' share of S0 starts there, below): its event rides on the split kernel's dispatch
  hipEvent_t split_ev = nullptr;
  bool split_rode = false;
  if (PM.on) {                           // bf16 hi/lo planes of the caller's inputs
    SplitBatch sb{};
    sb.lo_off = S.total;
    uint16_t* sv_hi = reinterpret_cast<uint16_t*>(saved + S.total);
    // (the plain planes of A / M are read by layer 0's projection and weight gradient in EVAL mode only; in training mode those
    // take the dropped copies written below, so the 2 x (B N D) bf16 store is skipped)
    const bool am_planes = !(train && lean && G == 1);
    sb.j[sb.n++] = SplitJob{io->A, am_planes ? sv_hi + S.in_a : nullptr, BND};
    sb.j[sb.n++] = SplitJob{io->M, am_planes ? sv_hi + S.in_m : nullptr, BND};
    sb.j[sb.n++] = SplitJob{io->dq, sv_hi + S.in_dq, (int64_t)BT * D};
    if (train && lean) { // layer 0's x-site: the planes of the two dropped copies of A and M from the same pass (no k_drop_x)
      const Keys K0 = layer_keys(train, seed, 0);
      for (int m = 0; m < 2; ++m)
        for (int gi = 0; gi < 2; ++gi) {
          sb.j[m].drop_hi[gi] = sv_hi + (S.layer_stride * 0 + S.mod_base + S.mod_stride * m + S.xd) + gi * BND;
          sb.j[m].drop_key[gi] = K0.x[m][gi];
        }
    }
    if (O.loss && O.loss->loss3) { sb.zero3 = O.loss->loss3; loss3_cleared = true; }   // before anything adds into them
    if (side && L > 1 && !(O.flags & (DVGR_OPT_S0_ON_SIDE | DVGR_OPT_S0_ON_MAIN | DVGR_OPT_FORKS_BY_RECORD))) {
      split_ev = side->next();
      if (!split_ev) return (int)hipErrorOutOfMemory;
      arm_stop(split_ev, st);
    }
    const int e_split = split_planes(st, sb);
    split_rode = stop_event_rode();
    if (e_split) return e_split;
  }

  // ---- S0: query side of every layer: the contraction Y_l = dq . W1_l^T, the row kernels, the small Wq contraction (below).
  const Probe fprobe{O, st};
  auto y_gemm = [&](int la, int lb_, hipStream_t on) -> int {   // Y_l = dq W1_l^T + b1_l for the layers [la, lb_)
    for (int l0 = la; l0 < lb_; l0 += kMaxBatch) {
      GemmBatch gb{};
      for (int l = l0; l < lb_ && l < l0 + kMaxBatch; ++l) {
        gb.p[gb.n] = nt(io->dq, D, lp(l) + P.qa_w1, D, ws + W.Y + (int64_t)l * BT * D, D, lp(l) + P.qa_b1, BT, D, D, ACT_NONE);
        PM.attach(gb.p[gb.n++]);
      }
      const Probe pr{O, on};
      DVGR_TRY(pr.begin(DVGR_PROBE_FWD_QA_GEMM, l0));
      DVGR_TRY(gemm_launch(on, KC, KC, gb, TAG_GENERIC, prec));
      DVGR_TRY(pr.end(DVGR_PROBE_FWD_QA_GEMM, l0));
    }
    return 0;
  };
  auto qa_rows = [&](int la, int cnt, hipStream_t on) -> int {  // the row kernels of the layers [la, la + cnt): every per-layer address is base + l * stride
    QaRowsArgs a{};
    a.Y = ws + W.Y + (int64_t)la * BT * D; a.logit = ws + W.R + (int64_t)la * BT;
    a.params = params + P.q_base + (int64_t)la * P.q_layer_stride; a.off_w2 = P.qa_w2; a.off_b2 = P.qa_b2; a.layer_stride = P.q_layer_stride;
    a.words = io->words; a.qlen = io->qlen; a.saved = saved + (int64_t)la * S.layer_stride; a.s_layer_stride = S.layer_stride;
    a.s_d = S.qa_d; a.s_nrm = S.qa_nrm; a.s_a = S.qa_a; a.s_ah = S.qa_ah; a.s_S = S.qa_s; a.s_q = S.qa_q;
    a.B = B; a.T = T; a.D = D; a.Wd = Wd; a.L = cnt;
    return qa_rows_fwd(on, a);
  };
  auto wq_gemm = [&](int la, int lb_, hipStream_t on) -> int {  // p = Wq q + bq of the layers [la, lb_), both modalities
    GemmBatch gb{};
    for (int l = la; l < lb_; ++l)
      for (int m = 0; m < 2; ++m) {
        gb.p[gb.n++] = nt(ls(l) + S.qa_q, Wd, qp(l, m) + P.wq, Wd, ms(l, m) + S.p, D, qp(l, m) + P.bq, B, D, Wd, ACT_NONE);
        if (gb.n == kMaxBatch || (l == lb_ - 1 && m == 1)) { DVGR_TRY(gemm_launch(on, KC, KC, gb, TAG_GENERIC, prec)); gb.n = 0; }
      }
    return 0;
  };
  // Where S0 runs (two streams):
  //   DVGR_OPT_S0_ON_SIDE   the form of rounds 1-5: the contraction of all layers on the main stream, rows + Wq on the side stream
  //                         under layer 0's separate projection, joined in front of the gate (layer 0 = two launches)
  //   DVGR_OPT_S0_ON_MAIN   everything on the main stream in front of layer 0, which then takes the fused kernel (no fork / join)
  //   default (round 6)     LAYER 0's share on the main stream in front of the fused layer-0 kernel; the other layers' share on the
  //                         side stream, forked right behind the input split and needed by layer 0's S5 (the gate of layer 1)
  const bool s0_side = side && (O.flags & DVGR_OPT_S0_ON_SIDE);
  const bool s0_all_main = side && !s0_side && ((O.flags & DVGR_OPT_S0_ON_MAIN) || L == 1);
  const bool s0_main = side && !s0_side;                     // layer 0's gate exists before layer 0 starts: it takes the fused kernel
  hipEvent_t s0_rest_ev = nullptr;
  if (!side || s0_all_main) {
    DVGR_TRY(y_gemm(0, L, st));
    DVGR_TRY(qa_rows(0, L, st));
    DVGR_TRY(wq_gemm(0, L, st));
  } else if (s0_side) {
    DVGR_TRY(y_gemm(0, L, st));
    DVGR_TRY(side->order(st, hs));
    DVGR_TRY(qa_rows(0, L, hs));
    DVGR_TRY(wq_gemm(0, L, hs));
  } else {
    if (split_ev) DVGR_TRY(side->order_armed(st, hs, split_ev, split_rode));   // (behind the input split: the planes of dq exist)
    else DVGR_TRY(side->order(st, hs));
    DVGR_TRY(y_gemm(1, L, hs));
    DVGR_TRY(qa_rows(1, L - 1, hs));
    DVGR_TRY(wq_gemm(1, L, hs));
    s0_rest_ev = side->next();
    if (!s0_rest_ev) return (int)hipErrorOutOfMemory;
    { const hipError_t e_ = hipEventRecord(s0_rest_ev, hs); if (e_ != hipSuccess) return (int)e_; }
    DVGR_TRY(y_gemm(0, 1, st));
    DVGR_TRY(qa_rows(0, 1, st));
    DVGR_TRY(wq_gemm(0, 1, st));
  }

  // ---- layers ---------------------------------------------------------------------------------------
  // graph_layers G > 1 (model/models.py:150-160; no shipped config): unit layer l runs G GAT steps; step j uses GAT module
  // [l + j] of each family (SHARED with every other (l', j') of the same sum -- gat_set_base), reads the previous step's
  // SPECIFIC output (step 0: the layer input) and the layer's one punish gate, and appends its two outputs to the lists
  // (entry l * G + j); the last step's pair feeds the SF attention.  The G > 1 path is the plain one: fp32 + planes
  // copies everywhere (no plane-only tensors), single stream in the backward.
  auto gp_ = [&](int k, int m) { return params + gat_set_base(P, L, k, m); };                  // GAT module k
  // S2 + S3 as one launch (dvgr_gatf.hip) where the shape allows it.  Layer 0 beside the side stream keeps two launches: its
  // gate needs the query chain S0, which runs on the side stream UNDER the separate projection.
  const bool fuse_ok = PM.on && !(O.flags & DVGR_OPT_NO_FUSED_GAT) && gat_fused_supported(B, N, D, prec);
  auto fuse_gat = [&](int l, int j) { return fuse_ok && !(side && !s0_main && l == 0 && j == 0); };
  auto gs_ = [&](int l, int m, int j) { return ms(l, m) + S.g_stride * j; };                   // saved block of graph layer j
  int aux_done = 0;                      // unit layers whose list entries have gone to the auxiliary-loss kernels
#ifdef DVGR_TIMING_AIDS
  static const int aux_group_env = []() { const char* e = getenv("DVGR_AUX_GROUP"); return e ? atoi(e) : 0; }();   // A/B timing aid
  const int aux_group = aux_group_env > 0 ? aux_group_env : 1;
#else
  constexpr int aux_group = 1;           // unit layers per loss group (see below)
#endif
  for (int l = 0; l < L; ++l) {
    hipEvent_t aux_ev = nullptr;         // the loss group's fork, when it rides on the layer's fused GAT-view launch
    bool aux_rode = false;
    const float* Xin[2] = {l == 0 ? io->A : ms(l - 1, 0) + S.x, l == 0 ? io->M : ms(l - 1, 1) + S.x};
    float* outc[2] = {nullptr, nullptr};
    float* outs[2] = {nullptr, nullptr};
    for (int j = 0; j < G; ++j) {
    const int e = l * G + j, k = l + j;      // list entry / dropout site index, GAT module index
    const Keys K = layer_keys(train, seed, e);
    const float* Xj[2] = {j == 0 ? Xin[0] : outs[0], j == 0 ? Xin[1] : outs[1]};      // (outs still holds step j - 1's)
    outc[0] = io->com_app + e * BND; outc[1] = io->com_mot + e * BND;
    outs[0] = io->aq_fusion + e * BND; outs[1] = io->mq_fusion + e * BND;
    // S2: Wh = drop_x(X) . Wcat^T + bcat.  Training: the two GATs of a modality see differently dropped
    // copies of X, materialised once (saved: the backward needs them for dWcat) so that the contraction
    // itself carries no mask and can run on the LDS-DMA ring kernel.
    // the dropped copies: layers > 0 get them from the previous layer's S5 kernel; layer 0 from the split kernel above
    // (bf16 modes: planes only, nobody reads them as fp32) or, in DVGR_F32, from k_drop_x; graph layers j > 0: k_drop_x
    if (train && ((l == 0 && !lean) || j > 0))
      DVGR_TRY(drop_x(st, Xj[0], Xj[1], gs_(l, 0, j) + S.xd, gs_(l, 1, j) + S.xd, K.x, BND, PM.out(gs_(l, 0, j) + S.xd, &plo),
                      PM.out(gs_(l, 1, j) + S.xd, &plo), plo));
    if (fuse_gat(l, j)) {
      // S2 + S3 as one launch (dvgr_gatf.hip).  The gate s must exist: layer 0 takes this branch only without a side stream,
      // i.e. with S0 in front of it on this stream; later layers got theirs from the previous layer's S5 kernel.
      if (l == 0 && j == 0)
        DVGR_TRY(punish_fwd(st, Xin[0], Xin[1], ms(l, 0) + S.p, ms(l, 1) + S.p, ms(l, 0) + S.s, ms(l, 1) + S.s, B, N, D));
      GatFusedArgs fa{};
      GatAttnArgs& a = fa.at;
      int64_t xlo = -1, wlo = -1, lo1 = 0;
      bool ok = true;
      for (int m = 0; m < 2 && ok; ++m) {
        a.s[m] = ms(l, m) + S.s;
        a.gat_a[m] = gp_(k, m) + P.gat_a; a.gat_c[m] = gp_(k, m) + P.gat_c;
        a.u[m] = gs_(l, m, j) + S.u; a.v[m] = gs_(l, m, j) + S.v;
        a.out_c[m] = outc[m]; a.out_s[m] = outs[m];
        for (int h = 0; h < 8; ++h) a.key_alpha[m][h] = K.alpha[m][h];
        for (int g = 0; g < 2; ++g) a.key_out[m][g] = K.out[m][g];
        a.zp_c[m] = PM.out(outc[m], &plo);
        a.zp_s[m] = PM.out(outs[m], &plo);
        for (int g = 0; g < 2; ++g) {
          fa.xh[m][g] = PM.find(train ? gs_(l, m, j) + S.xd + g * BND : Xj[m], &lo1);
          ok = ok && fa.xh[m][g] && (xlo < 0 || xlo == lo1);
          xlo = lo1;
        }
        fa.wh[m] = PM.find(gp_(k, m) + P.gat_w, &lo1);
        ok = ok && fa.wh[m] && (wlo < 0 || wlo == lo1);
        wlo = lo1;
        fa.bias[m] = gp_(k, m) + P.gat_b;
        fa.Wh_out[m] = gs_(l, m, j) + S.wh;
        if (O.lo8 && prec == DVGR_BF16X3 && !(O.flags & DVGR_OPT_NO_LO8) && k < L) {      // (extra GAT sets, graph_layers > 1: bf16 lo planes)
          const Lo8Layout Y = lo8_layout(d);
          const uint8_t* base = reinterpret_cast<const uint8_t*>(O.lo8);
          fa.wq[m] = base + Y.wq + (int64_t)(k * 2 + m) * HF * D;
          fa.wscale[m] = reinterpret_cast<const float*>(base + Y.ws) + (int64_t)(k * 2 + m) * HF;
        }
      }
      if (!ok) return DVGR_E_DIMS;          // (cannot happen: fuse_gat() implies planes of every operand)
      a.z_lo_off = plo;
      fa.kb64 = (O.flags & DVGR_OPT_KB32) ? 0 : 1;
      fa.no_half = (O.flags & DVGR_OPT_NO_HALF_GAT_TILES) ? 1 : 0;
      fa.x_lo_off = xlo; fa.w_lo_off = wlo;
      a.B = B; a.N = N; a.D = D;
      DVGR_TRY(fprobe.begin(DVGR_PROBE_FWD_GAT_GEMM, l));
      // the loss group's fork follows this launch (the layer's last graph step): its event rides on the launch
      // (not where the caller's probe times this launch: a stop event on the dispatch lengthens what the probe's end event sees by
      // the signal's path -- 44.9 instead of 39.9 us at C2 while rocprofv3 reads the same kernel time -- and the probe is the bench
      // line's roofline figure)
      if (side && O.loss && j == G - 1 && ((l + 1) % aux_group == 0 || l == L - 1) && !fprobe.hit(DVGR_PROBE_FWD_GAT_GEMM, l) &&
          !(O.flags & DVGR_OPT_FORKS_BY_RECORD)) {
        aux_ev = side->next();
        if (!aux_ev) return (int)hipErrorOutOfMemory;
        arm_stop(aux_ev, st);
      }
      const int e_gat = gat_fused_fwd(st, fa, prec);
      aux_rode = stop_event_rode();
      if (e_gat) return e_gat;
      DVGR_TRY(fprobe.end(DVGR_PROBE_FWD_GAT_GEMM, l));
    } else {
      GemmBatch gb{};
      for (int m = 0; m < 2; ++m) {
        if (!train) {
          gb.p[gb.n] = nt(Xj[m], D, gp_(k, m) + P.gat_w, D, gs_(l, m, j) + S.wh, HF, gp_(k, m) + P.gat_b, BN, HF, D, ACT_NONE);
          PM.attach(gb.p[gb.n++]);
        } else {
          for (int g = 0; g < 2; ++g) {
            gb.p[gb.n] = nt(gs_(l, m, j) + S.xd + g * BND, D, gp_(k, m) + P.gat_w + (int64_t)g * 4 * F * D, D,
                            gs_(l, m, j) + S.wh + g * 4 * F, HF, gp_(k, m) + P.gat_b + g * 4 * F, BN, 4 * F, D, ACT_NONE);
            PM.attach(gb.p[gb.n++]);
          }
        }
      }
      DVGR_TRY(fprobe.begin(DVGR_PROBE_FWD_GAT_GEMM, l));
      DVGR_TRY(gemm_launch(st, KC, KC, gb, TAG_FWD_GAT, prec));
      DVGR_TRY(fprobe.end(DVGR_PROBE_FWD_GAT_GEMM, l));
    }
    // S1 (needs p from S0: the side stream joins here)
    if (l == 0 && j == 0 && side && !s0_main) DVGR_TRY(side->order(hs, st));
    if (l == 0 && j == 0 && !fuse_gat(l, j))   // layers > 0: gate computed by the previous layer's S5 kernel
      DVGR_TRY(punish_fwd(st, Xin[0], Xin[1], ms(l, 0) + S.p, ms(l, 1) + S.p, ms(l, 0) + S.s, ms(l, 1) + S.s, B, N, D));
    // S3
    if (!fuse_gat(l, j)) {
      GatAttnArgs a{};
      for (int m = 0; m < 2; ++m) {
        a.Wh[m] = gs_(l, m, j) + S.wh; a.s[m] = ms(l, m) + S.s;
        a.gat_a[m] = gp_(k, m) + P.gat_a; a.gat_c[m] = gp_(k, m) + P.gat_c;
        a.u[m] = gs_(l, m, j) + S.u; a.v[m] = gs_(l, m, j) + S.v;
        a.out_c[m] = outc[m]; a.out_s[m] = outs[m];
        for (int h = 0; h < 8; ++h) a.key_alpha[m][h] = K.alpha[m][h];
        for (int g = 0; g < 2; ++g) a.key_out[m][g] = K.out[m][g];
        a.zp_c[m] = PM.out(outc[m], &plo);
        a.zp_s[m] = PM.out(outs[m], &plo);
      }
      a.z_lo_off = plo;
      a.B = B; a.N = N; a.D = D;
      DVGR_TRY(fprobe.begin(DVGR_PROBE_FWD_GAT_ATTN, l));
      DVGR_TRY(gat_attn_fwd(st, a));
      DVGR_TRY(fprobe.end(DVGR_PROBE_FWD_GAT_ATTN, l));
    }
    }  // graph layers
    // The auxiliary losses and their gradients (VALU / HBM work on one (sample, list entry) at a time, means over all
    // entries) go to the side stream in groups of unit layers as the entries come into existence, each group under the
    // contractions that follow.  Round 3 measured no difference between group sizes (the forward's tail was the main stream's);
    // with S6 as one launch and sum(vis) inside the out contraction the tail is the loss chain (80 us at C2 for all three
    // layers at once, starting 12 us behind the last GAT kernel): one layer per group (experiments_r04.tsv, aux_groups_*:
    // C2 1.0065 ms against 1.0085 for one group of three; C5 9.177 against 9.205 in threes and 9.237 for all twelve at once).
    if (O.loss && ((l + 1) % aux_group == 0 || l == L - 1)) {
      const dvgr_loss_spec& Q = *O.loss;
      if (!Q.loss3 || !Q.g_com_app || !Q.g_com_mot || !Q.g_aq_fusion || !Q.g_mq_fusion) return DVGR_E_NULL;
      if (side && aux_ev) DVGR_TRY(side->order_armed(st, hs, aux_ev, aux_rode));
      else if (side) DVGR_TRY(side->order(st, hs));
      dvgr_io lio = *io;
      lio.vis = nullptr;                   // sum(vis) follows at the very end
      float* const gl[4] = {Q.g_com_app, Q.g_com_mot, Q.g_aq_fusion, Q.g_mq_fusion};
      dvgr_dims dl = d;                    // the losses run over the list ENTRIES: L * G of them (train.py:148-154: len(list))
      dl.L = L * G; dl.G = 1;
      const int e0 = aux_done * G, e1 = (l + 1) * G;
      DVGR_TRY(aux_losses(hs, dl, &lio, Q.alpha, Q.beta, Q.dep_scale, Q.loss3, nullptr, gl, ws + fwd_ws(d).total,
                          !loss3_cleared && aux_done == 0, e0, e1 - e0));
      aux_done = l + 1;
    }
    // S4: tn_v = tanh(z_v . P1^T + bP)
    GemmBatch gb4{};
    for (int m = 0; m < 2; ++m) {
      gb4.p[gb4.n] = nt(outc[m], D, mp(l, m) + P.sf_p1, D, ms(l, m) + S.tn, D, mp(l, m) + P.sf_bp, BN, D, D, ACT_TANH);
      PM.attach(gb4.p[gb4.n++]);
      gb4.p[gb4.n] = nt(outs[m], D, mp(l, m) + P.sf_p1, D, ms(l, m) + S.tn + BND, D, mp(l, m) + P.sf_bp, BN, D, D, ACT_TANH);
      PM.attach(gb4.p[gb4.n++]);
    }
    // S5
    SfCombineArgs a5{};
    for (int m = 0; m < 2; ++m) {
      a5.X[m] = Xin[m]; a5.z0[m] = outc[m]; a5.z1[m] = outs[m]; a5.tn[m] = ms(l, m) + S.tn;
      a5.p2[m] = mp(l, m) + P.sf_p2; a5.beta[m] = ms(l, m) + S.beta; a5.Xout[m] = ms(l, m) + S.x;
      a5.o[m] = (l == L - 1) ? (m == 0 ? io->aq_embed : io->mq_embed) : nullptr;
      // planes of X': read by the next layer's projection and weight gradient in EVAL mode and by the MFB after the last layer;
      // in training mode those read the dropped copies instead, so layers < L - 1 skip the 2 x (B N D) bf16 store
      a5.xp[m] = (train && G == 1 && l < L - 1) ? nullptr : PM.out(ms(l, m) + S.x, &plo);
    }
    a5.x_lo_off = plo;
    a5.rows = BN; a5.D = D; a5.N = N;
    if (l + 1 < L) {                     // S1 and the x-site dropout of layer l + 1, on the row while it is in registers
      const Keys Kn = layer_keys(train, seed, (l + 1) * G);
      for (int m = 0; m < 2; ++m) {
        a5.next_p[m] = ms(l + 1, m) + S.p; a5.next_s[m] = ms(l + 1, m) + S.s;
        a5.next_xd[m] = (train && !lean) ? ms(l + 1, m) + S.xd : nullptr;       // fp32 copies only where they are read
        a5.next_xd_p[m] = train ? PM.out(ms(l + 1, m) + S.xd, &plo) : nullptr;
        for (int gi = 0; gi < 2; ++gi) a5.next_key[m][gi] = Kn.x[m][gi];
      }
      a5.next_xd_lo_off = plo;
    }
    // S4 hands S5 its logits tn_v . p2 as partial dots per 32-column block from its epilogue where the launch runs on whole
    // tiles of the planes kernels: S5 then does not read the 2 x (B N D) tanh outputs back (they stay saved for the backward)
    {
      const int64_t wstride = round64((int64_t)2 * (D / 32) * BN);
      for (int i = 0; i < gb4.n; ++i) {
        const int m = i >> 1, v = i & 1;
        gb4.p[i].rowdot_vec = mp(l, m) + P.sf_p2;
        gb4.p[i].rowdot_out = ws + W.wpart + m * wstride + (int64_t)v * (D / 32) * BN;
        gb4.p[i].rowdot_ld = BN;
      }
      if ((D % 32) == 0 && !(O.flags & (DVGR_OPT_NO_FUSED_GAT | DVGR_OPT_S5_READS_TN)) && gemm_rowdot_supported(KC, KC, gb4, prec)) {
        for (int m = 0; m < 2; ++m) a5.wpart[m] = ws + W.wpart + m * wstride;
      } else {
        for (int i = 0; i < gb4.n; ++i) { gb4.p[i].rowdot_vec = nullptr; gb4.p[i].rowdot_out = nullptr; }
      }
    }
    bool phased = false;
#ifdef DVGR_TIMING_AIDS
    static const bool phase_on = []() { const char* e = getenv("DVGR_PHASE"); return e && e[0] == '1'; }();
    if (phase_on) {                      // experiment: S4 and S5 as two phases of one launch behind a grid barrier
      const int e = fwd_sf_phase_launch(st, gb4, a5, prec);
      if (e == 0) phased = true;
      else if (e != DVGR_E_DIMS) return e;
    }
#endif
    if (!phased) {
      DVGR_TRY(fprobe.begin(DVGR_PROBE_FWD_SF_GEMM, l));
      DVGR_TRY(gemm_launch(st, KC, KC, gb4, TAG_FWD_SF, prec));
      DVGR_TRY(fprobe.end(DVGR_PROBE_FWD_SF_GEMM, l));
      if (l == 0 && s0_rest_ev) {          // S5 of layer 0 computes layer 1's gate from p_1: the other layers' share of S0
        const hipError_t e_ = hipStreamWaitEvent(st, s0_rest_ev, 0);
        if (e_ != hipSuccess) return (int)e_;
      }
      DVGR_TRY(sf_combine_fwd(st, a5));
    }
  }

  // ---- S6: MFB ------------------------------------------------------------------------------------------
  bool vis_sum_fused = false;
  {
    // linear0 / linear1 + ELU + the pooled product as ONE launch on the planes path (k_mfb_fwd, dvgr_gemm.hip); otherwise the
    // two-problem contraction followed by k_mfb_pool_fwd
    MfbFwdArgs fa{};
    int64_t lx0 = 0, lx1 = 0, lw0 = 0, lw1 = 0;
    bool fused6 = PM.on && !(O.flags & DVGR_OPT_NO_FUSED_GAT) && BN > 128 && mfb_fwd_supported(BN, D, kMfbHidden, prec);
    if (fused6) {
      fa.xa = PM.find(ms(L - 1, 0) + S.x, &lx0); fa.xm = PM.find(ms(L - 1, 1) + S.x, &lx1);
      fa.w0 = PM.find(params + P.mfb_w0, &lw0); fa.w1 = PM.find(params + P.mfb_w1, &lw1);
      fused6 = fa.xa && fa.xm && fa.w0 && fa.w1 && lx0 == lx1 && lw0 == lw1;
    }
    if (fused6) {
      fa.x_lo = lx0; fa.w_lo = lw0;
      fa.b0 = params + P.mfb_b0; fa.b1 = params + P.mfb_b1;
      fa.x0 = saved + S.mfb_x0; fa.x1 = saved + S.mfb_x1; fa.z = saved + S.mfb_z;
      fa.zp = PM.out(saved + S.mfb_z, &plo); fa.z_lo = plo;
      fa.rows = BN; fa.K = D; fa.hidden = kMfbHidden;
      DVGR_TRY(fprobe.begin(DVGR_PROBE_FWD_MFB_GEMM, 0));
      DVGR_TRY(mfb_fwd_launch(st, fa, prec));
      DVGR_TRY(fprobe.end(DVGR_PROBE_FWD_MFB_GEMM, 0));
    } else {
    GemmBatch gb{};
    gb.p[gb.n] = nt(ms(L - 1, 0) + S.x, D, params + P.mfb_w0, D, saved + S.mfb_x0, kMfbHidden, params + P.mfb_b0, BN, kMfbHidden, D, ACT_ELU);
    PM.attach(gb.p[gb.n++]);
    gb.p[gb.n] = nt(ms(L - 1, 1) + S.x, D, params + P.mfb_w1, D, saved + S.mfb_x1, kMfbHidden, params + P.mfb_b1, BN, kMfbHidden, D, ACT_ELU);
    PM.attach(gb.p[gb.n++]);
    DVGR_TRY(fprobe.begin(DVGR_PROBE_FWD_MFB_GEMM, 0));
    DVGR_TRY(gemm_launch(st, KC, KC, gb, TAG_GENERIC, prec));
    DVGR_TRY(fprobe.end(DVGR_PROBE_FWD_MFB_GEMM, 0));
    uint16_t* zfp = PM.out(saved + S.mfb_z, &plo);
    DVGR_TRY(mfb_pool_fwd(st, saved + S.mfb_x0, saved + S.mfb_x1, saved + S.mfb_z, BN, zfp, plo));
    }
    GemmBatch go{};
    go.p[go.n] = nt(saved + S.mfb_z, kMfbPooled, params + P.mfb_wo, kMfbPooled, io->vis, D, params + P.mfb_bo, BN, D, kMfbPooled, ACT_ELU);
    PM.attach(go.p[go.n++]);
    // sum(vis) and d sum(vis) / d vis = 1 of the scope-U loss ride in this contraction's epilogue where it runs on the planes
    // kernels (one float atomic per workgroup into loss3[0], cleared at the head of the forward): no k_sum_and_ones launch
    if (O.loss && O.loss->g_vis && loss3_cleared) {
      go.p[0].sum_out = O.loss->loss3;
      go.p[0].ones_out = O.loss->g_vis;
      if (gemm_sum_supported(KC, KC, go, prec)) vis_sum_fused = true;
      else { go.p[0].sum_out = nullptr; go.p[0].ones_out = nullptr; }
    }
    DVGR_TRY(gemm_launch(st, KC, KC, go, TAG_GENERIC, prec));
  }
  if (O.loss) {
    if (side) DVGR_TRY(side->order(hs, st));
    if (O.loss->g_vis && !vis_sum_fused) DVGR_TRY(sum_and_ones(st, io->vis, O.loss->g_vis, BND, O.loss->loss3));
  }
  return 0;
}

// =================================================================================================
// Backward schedule
// =================================================================================================
int dvgr_unit_backward(const dvgr_dims* dp, const float* params, const void* params_lp, const dvgr_io* io,
                       const dvgr_grads* g, const void* saved_v, size_t saved_bytes, void* ws_v, size_t ws_bytes,
                       const dvgr_opts* opts, void* stream) {
  const dvgr_opts o0{};
  const dvgr_opts& O = opts ? *opts : o0;
  const int train = O.train;
  const uint64_t seed = O.seed;
  void* const* layer_done_events = O.layer_done_events;
  DVGR_TRY(check_common(dp, params, io, saved_v, saved_bytes, ws_v, ws_bytes));
  if (!g || !g->gA || !g->gM || !g->gdq || !g->gwords || !g->gparams) return DVGR_E_NULL;
  if (!aligned16(g->gA) || !aligned16(g->gM) || !aligned16(g->gdq) || !aligned16(g->gwords) || !aligned16(g->gparams))
    return DVGR_E_ALIGN;
  const dvgr_dims d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const Probe probe{O, st};
  const int prec = dp->dtype;
  const ParamLayout P = param_layout(d);
  const SavedLayout S = saved_layout(d);
  const BwdWs W = bwd_ws(d);
  const float* saved = (const float*)saved_v;
  float* ws = (float*)ws_v;
  float* gp = g->gparams;
  const int B = d.B, N = d.N, T = d.T, D = d.D, Wd = d.Wd, L = d.L, F = D / 4, HF = kHeads * F;
  const int BN = B * N, BT = B * T;
  const int64_t BND = (int64_t)BN * D;
  auto lp = [&](int l) { return params + P.q_base + P.q_layer_stride * l; };                                       // query block
  auto qp = [&](int l, int m) { return params + P.q_base + P.q_layer_stride * l + P.q_mod_base + P.q_mod_stride * m; };
  auto mp = [&](int l, int m) { return params + P.layer_stride * l + P.mod_stride * m; };                          // layer block
  auto glp = [&](int l) { return gp + P.q_base + P.q_layer_stride * l; };
  auto gqp = [&](int l, int m) { return gp + P.q_base + P.q_layer_stride * l + P.q_mod_base + P.q_mod_stride * m; };
  auto gmp = [&](int l, int m) { return gp + P.layer_stride * l + P.mod_stride * m; };
  auto ls = [&](int l) { return saved + S.layer_stride * l; };
  auto ms = [&](int l, int m) { return saved + S.layer_stride * l + S.mod_base + S.mod_stride * m; };
  float* dXacc[2] = {g->gA, g->gM};     // running gradient w.r.t. the current layer boundary
  const PlaneMap PM = make_plane_map(d, P, S, params, params_lp, io, saved, ws);
  int64_t plo = 0;
  const int G = d.G;
  const bool lean = lean_planes(PM.on, BN, G);   // dpre / dWh / xd exist as planes only (see lean_planes)
  if (G > 1 && (layer_done_events || O.comm)) return DVGR_E_DIMS;   // shared GAT modules: a block is final only at the very end
  auto gpp_ = [&](int k, int m) { return params + gat_set_base(P, L, k, m); };                 // GAT module k: parameters,
  auto ggp_ = [&](int k, int m) { return gp + gat_set_base(P, L, k, m); };                     //   gradients
  auto gs_ = [&](int l, int m, int j) { return ms(l, m) + S.g_stride * j; };                   // saved block of graph layer j

  // every element of gparams is written: the matrices by their contractions, everything between them (bias and vector
  // gradients, which contractions and column sums ADD into -- GemmProblem::rowsum -- and the alignment padding) is zeroed
  // here in one launch; gdq is the sum over layers of partial tiles added atomically, so it starts from zero as well.
  // (Two hipMemsetAsync over the whole buffers took 17 us at the head of the backward's critical path.)
  {
    ZeroList z{};
    auto add = [&](float* p, int64_t n) -> int {
      if (n <= 0) return 0;
      if (z.count == kMaxZeroSegs) { DVGR_TRY(zero_ranges(st, z)); z.count = 0; }
      z.p[z.count] = p; z.n[z.count++] = n;
      return 0;
    };
    DVGR_TRY(add(g->gdq, (int64_t)BT * D));
    for (int l = 0; l < L; ++l)
      for (int m = 0; m < 2; ++m) {
        DVGR_TRY(add(gmp(l, m) + P.gat_b, P.sf_p1 - P.gat_b));
        DVGR_TRY(add(gmp(l, m) + P.sf_bp, P.mod_stride - P.sf_bp));
      }
    for (int k = L; k < L * G; ++k)              // extra GAT sets: the vectors of the used ones, everything of the unused ones
      for (int m = 0; m < 2; ++m) {
        const bool used = k <= L + G - 2;
        DVGR_TRY(add(ggp_(k, m) + (used ? P.gat_b : 0), P.extra_mod_stride - (used ? P.gat_b : 0)));
      }
    DVGR_TRY(add(gp + P.mfb_b0, P.mfb_w1 - P.mfb_b0));
    DVGR_TRY(add(gp + P.mfb_b1, P.mfb_wo - P.mfb_b1));
    DVGR_TRY(add(gp + P.mfb_bo, P.q_base - P.mfb_bo));
    for (int l = 0; l < L; ++l) {
      DVGR_TRY(add(glp(l) + P.qa_b1, P.q_mod_base - P.qa_b1));
      for (int m = 0; m < 2; ++m) DVGR_TRY(add(gqp(l, m) + P.bq, P.q_mod_stride - P.bq));
    }
    // the last batch of ranges goes out together with the first element-wise kernel of the backward (S6: the ELU derivative)
    uint16_t* dpop = PM.out(ws + W.dpo, &plo);
    DVGR_TRY(bwd_head(st, z, g->g_vis, io->vis, ws + W.dpo, BND, dpop, plo));
  }

  // Two streams: `st` carries the critical path (the dX chain), `hs` every weight / bias gradient.
  // DVGR_OPT_SINGLE_STREAM (or a missing side stream) folds everything onto `st`.
  SideStream* side = ((O.flags & DVGR_OPT_SINGLE_STREAM) || 4 * L + 16 > 256 || G > 1) ? nullptr : side_stream(st);
  hipStream_t hs = side ? side->st : st;
  // query branch (see dvgr_side.h).  Small batches are bound by the host's enqueue rate, where a third stream only adds
  // event calls (C1, B N = 160: 0.43 ms per step without it, 0.48 ms with it).
  const bool use_qs = !(O.flags & DVGR_OPT_NO_QUERY_STREAM) && BND >= (int64_t)512 * 768;
  hipStream_t qs = side ? side->query(use_qs) : st;
  (void)stop_event_rode();               // (no armed stop event survives from an earlier call of this thread that failed half-way)
  // the per-layer fork behind the fused backward launch (layer 0: behind its d p launch) rides on that launch's dispatch
  hipEvent_t fork_ev = nullptr;
  bool fork_rode = false;
  auto arm_fork = [&]() -> int {
    fork_ev = nullptr; fork_rode = false;
    if (!side || (O.flags & DVGR_OPT_FORKS_BY_RECORD)) return 0;
    fork_ev = side->next();
    if (!fork_ev) return (int)hipErrorOutOfMemory;
    arm_stop(fork_ev, st);
    return 0;
  };
  auto to_side = [&]() -> int {                                               // hs continues after st's work so far
    if (!side) return 0;
    if (fork_ev) { const int r = side->order_armed(st, hs, fork_ev, fork_rode); fork_ev = nullptr; return r; }
    return side->order(st, hs);
  };
  auto to_main = [&]() -> int { return side ? side->order(hs, st) : 0; };     // st continues after hs's work so far
  auto to_side_and_query = [&]() -> int {                                     // one record on st, both other streams wait
    if (!side) return 0;
    hipEvent_t e = fork_ev ? fork_ev : side->next();
    if (!e) return (int)hipErrorOutOfMemory;
    hipError_t r = (fork_ev && fork_rode) ? hipSuccess : hipEventRecord(e, st);
    fork_ev = nullptr;
    if (r == hipSuccess) r = hipStreamWaitEvent(hs, e, 0);
    if (r == hipSuccess && qs != hs) r = hipStreamWaitEvent(qs, e, 0);
    return (int)r;
  };
  auto to_query = [&]() -> int { return side ? side->order(st, qs) : 0; };
  // (every fork costs the main stream a ~6 us bubble -- the event's barrier packet -- so there is one per layer, not one
  // per dependency: the side stream's work of a layer is enqueued behind the layer's last producer it needs)
  auto layer_done = [&](int index, hipStream_t on) -> int {
    hipEvent_t ev = (layer_done_events && layer_done_events[index]) ? (hipEvent_t)layer_done_events[index] : nullptr;
    if (!ev && !O.comm) return 0;
    if (!ev) {                               // native all-reduce without caller events: one of the library's own
      ev = side ? side->next() : nullptr;
      if (!ev) {
        static thread_local hipEvent_t own[16] = {};
        hipEvent_t& slot = own[index & 15];
        if (!slot && hipEventCreateWithFlags(&slot, kEventFlags) != hipSuccess) return (int)hipErrorOutOfMemory;
        ev = slot;
      }
    }
    const hipError_t e = hipEventRecord(ev, on);
    if (e != hipSuccess) return (int)e;
    if (O.comm) {                            // the block's all-reduce enters the communication stream right here
      int64_t first = 0, last = 0;
      DVGR_TRY(dvgr_param_layer_range(dp, index, &first, &last));
      DVGR_TRY(dvgr_comm_allreduce_avg((dvgr_comm*)O.comm, gp + first, last - first, (void*)ev, (void*)on));
    }
    if (O.layer_done_cb && layer_done_events && layer_done_events[index]) O.layer_done_cb(index, O.layer_done_user);
    return 0;
  };
  // A weight gradient dW = dY^T . X carries its bias gradient dY^T . 1 along (GemmProblem::rowsum) where the engine
  // supports it for the batch; otherwise the bias gets a column-sum job.
  struct BiasOf { const float* dY; int rows; int cols; int64_t ld; float* out; };
  auto launch_dw = [&](hipStream_t on, GemmBatch& gb, const BiasOf* bias, int tag, ColsumBatch& cb) -> int {
    const bool fused = gemm_rowsum_supported(IC, IC, gb, prec);
    for (int i = 0; i < gb.n; ++i) {
      if (!bias[i].out) continue;
      if (fused) gb.p[i].rowsum = bias[i].out;
      else cb.j[cb.n++] = ColsumJob{bias[i].dY, nullptr, nullptr, nullptr, bias[i].rows, 0, bias[i].cols, bias[i].ld, bias[i].out};
    }
    return gemm_launch(on, IC, IC, gb, tag, prec);
  };

  // Data-gradient (NN) and weight-gradient (TN) problems that become ready together go out as ONE launch where the engine
  // can mix layouts (gemm_launch_mixed); otherwise one launch per kind.  bias[i] belongs to gb.p[i] (TN problems only).
  auto launch_mixed = [&](hipStream_t on, GemmBatch& gb, const BiasOf* bias, int tag, ColsumBatch& cb) -> int {
    if (gemm_mixed_supported(gb, prec)) {
      for (int i = 0; i < gb.n; ++i)
        if (bias[i].out) gb.p[i].rowsum = bias[i].out;
      // longest tiles first: workgroups are dispatched in problem order, and a tile's run time grows with its reduction
      // length (dP1: 2 B N rows, dWcat: B N, dX: D or 2 D) -- a long tile that starts last would finish alone on its
      // compute unit (measured: 97 us with the dX tiles first against the 60-70 us the total work fills)
      GemmBatch sorted{};
      bool used[kMaxBatch] = {};
      for (int k = 0; k < gb.n; ++k) {
        int best = -1;
        for (int i = 0; i < gb.n; ++i)
          if (!used[i] && (best < 0 || gb.p[i].K > gb.p[best].K)) best = i;
        used[best] = true;
        sorted.p[sorted.n++] = gb.p[best];
      }
      return gemm_launch_mixed(on, sorted, tag, prec);
    }
    GemmBatch gn{}, gt{};
    BiasOf bt[kMaxBatch] = {};
    for (int i = 0; i < gb.n; ++i) {
      if (gb.p[i].layout == LAYOUT_TN) { bt[gt.n] = bias[i]; gt.p[gt.n++] = gb.p[i]; }
      else gn.p[gn.n++] = gb.p[i];
    }
    if (gn.n) DVGR_TRY(gemm_launch(on, KC, IC, gn, tag, prec));
    if (gt.n) DVGR_TRY(launch_dw(on, gt, bt, tag == TAG_BWD_GAT_DX ? TAG_BWD_GAT_DW : tag, cb));
    return 0;
  };

  // S4 backward (data) + S3 backward as ONE launch per layer (dvgr_gatb.hip) where the shape allows it: the dz contraction
  // on a tile of whole samples x one head's columns with the attention backward as its epilogue.  Its B operand is P1^T
  // (rows = the head's columns, reduction index contiguous): the planes of every layer's P1 are transposed into plane-only
  // workspace slots by one small launch, off the critical path (side stream; 14 MB at C2).
  const bool fuse_bwd = PM.on && backward_fuses(d, O.flags);
  // ... and forms the part of dz that k_sf_bwd would write (beta_v do + the list gradient) itself, from the tensors k_sf_bwd reads
  // anyway: 2 x (B N D) fp32 fewer to store and to read back, and the list gradients are read once instead of twice
  const bool fuse_dz = fuse_bwd && !(O.flags & DVGR_OPT_FUSED_BWD_READS_DZ);
  hipEvent_t p1t_done = nullptr;
  int64_t p1t_lo = 0;
  auto p1t_hi = [&](int l, int m) -> const uint16_t* {
    if (O.p1t) {                                   // prepared by the caller (dvgr_pack_p1t): nothing to transpose in this call
      p1t_lo = 2 * (int64_t)L * D * D;
      return reinterpret_cast<const uint16_t*>(O.p1t) + (int64_t)(l * 2 + m) * D * D;
    }
    return PM.out(ws + W.p1t + (int64_t)(l * 2 + m) * round64((int64_t)D * D), &p1t_lo);
  };
  auto p1_transposed = [&](hipStream_t on) -> int {
    if (!fuse_bwd || O.p1t) return 0;
    TransposeBatch tb{};
    int64_t slo = 0;
    for (int l = 0; l < L; ++l)
      for (int m = 0; m < 2; ++m) {
        const uint16_t* src = PM.find(mp(l, m) + P.sf_p1, &slo);
        uint16_t* dst = const_cast<uint16_t*>(p1t_hi(l, m));
        if (!src || !dst) return DVGR_E_DIMS;        // (cannot happen: PM.on implies planes of the parameters and of the workspace)
        tb.j[tb.n++] = TransposeJob{src, dst};
      }
    tb.src_lo = slo; tb.dst_lo = p1t_lo; tb.D = D;
    DVGR_TRY(transpose_planes(on, tb));
    if (side && on != st) {
      p1t_done = side->next();
      if (!p1t_done) return (int)hipErrorOutOfMemory;
      const hipError_t e = hipEventRecord(p1t_done, on);
      if (e != hipSuccess) return (int)e;
    }
    return 0;
  };

  // ---- S6 backward (its ELU derivative went out with the zero list above) ---------------------------------
  {
    GemmBatch gd{};   // dz = dpo . Wo
    gd.p[gd.n] = nn(ws + W.dpo, D, params + P.mfb_wo, kMfbPooled, ws + W.dzf, kMfbPooled, BN, kMfbPooled, D, 0);
    PM.attach(gd.p[gd.n++]);
    DVGR_TRY(gemm_launch(st, KC, IC, gd, TAG_GENERIC, prec));
  }
  {
    uint16_t* q0 = PM.out(ws + W.dp0, &plo);
    uint16_t* q1 = PM.out(ws + W.dp1, &plo);
    // (two streams: the fork of the MFB weight gradients follows this launch and rides on it)
    hipEvent_t mfb_ev = nullptr;
    if (side && !(O.flags & DVGR_OPT_FORKS_BY_RECORD)) {
      mfb_ev = side->next();
      if (!mfb_ev) return (int)hipErrorOutOfMemory;
      arm_stop(mfb_ev, st);
    }
    const int e_pool = mfb_pool_bwd(st, ws + W.dzf, saved + S.mfb_x0, saved + S.mfb_x1, ws + W.dp0, ws + W.dp1, BN, q0, q1, plo);
    const bool mfb_rode = stop_event_rode();
    if (e_pool) return e_pool;
    // dA = dp0 . W0, dM = dp1 . W1 and the three MFB weight gradients (+ their biases) in ONE launch
    GemmBatch gb{};
    BiasOf bias[kMaxBatch] = {};
    gb.p[gb.n] = nn(ws + W.dp0, kMfbHidden, params + P.mfb_w0, D, dXacc[0], D, BN, D, kMfbHidden, 0);
    PM.attach(gb.p[gb.n++]);
    gb.p[gb.n] = nn(ws + W.dp1, kMfbHidden, params + P.mfb_w1, D, dXacc[1], D, BN, D, kMfbHidden, 0);
    PM.attach(gb.p[gb.n++]);
    bias[gb.n] = BiasOf{ws + W.dpo, BN, D, D, gp + P.mfb_bo};
    gb.p[gb.n] = tn(ws + W.dpo, D, saved + S.mfb_z, kMfbPooled, gp + P.mfb_wo, kMfbPooled, D, kMfbPooled, BN);
    PM.attach(gb.p[gb.n++]);
    bias[gb.n] = BiasOf{ws + W.dp0, BN, kMfbHidden, kMfbHidden, gp + P.mfb_b0};
    gb.p[gb.n] = tn(ws + W.dp0, kMfbHidden, ms(L - 1, 0) + S.x, D, gp + P.mfb_w0, D, kMfbHidden, D, BN);
    PM.attach(gb.p[gb.n++]);
    bias[gb.n] = BiasOf{ws + W.dp1, BN, kMfbHidden, kMfbHidden, gp + P.mfb_b1};
    gb.p[gb.n] = tn(ws + W.dp1, kMfbHidden, ms(L - 1, 1) + S.x, D, gp + P.mfb_w1, D, kMfbHidden, D, BN);
    PM.attach(gb.p[gb.n++]);
    ColsumBatch cb{};
    if (side) {
      // The three weight gradients are leaves (40 K-steps per tile against 16 for dA / dM): side stream, behind a fork, under
      // the last layer's S5 backward; dA / dM alone are 120 tiles and take the 64-row kernel.  (One mixed launch on the
      // main stream: 1.115 against 1.100 ms.  Small batches: the batch-sized kernels take the weight gradients.)
      GemmBatch gn{}, gt{};
      BiasOf bt[kMaxBatch] = {};
      for (int i = 0; i < gb.n; ++i) {
        if (gb.p[i].layout == LAYOUT_TN) { bt[gt.n] = bias[i]; gt.p[gt.n++] = gb.p[i]; }
        else gn.p[gn.n++] = gb.p[i];
      }
      if (mfb_ev) DVGR_TRY(side->order_armed(st, hs, mfb_ev, mfb_rode));
      else DVGR_TRY(side->order(st, hs));
      DVGR_TRY(p1_transposed(hs));           // (first thing on the side stream: the last layer's fused kernel waits for it)
      DVGR_TRY(gemm_launch(st, KC, IC, gn, TAG_GENERIC, prec));
      DVGR_TRY(launch_dw(hs, gt, bt, TAG_GENERIC, cb));
      if (cb.n) { cb.chunks = 16; DVGR_TRY(colsum(hs, cb)); }
      DVGR_TRY(layer_done(L, hs));
    } else {
      DVGR_TRY(p1_transposed(st));
      DVGR_TRY(launch_mixed(st, gb, bias, TAG_GENERIC, cb));
      if (cb.n) { cb.chunks = 16; DVGR_TRY(colsum(st, cb)); }
      DVGR_TRY(layer_done(L, st));
    }
  }

  // ---- query branch (S1 tail and S0 backward) of the layers [lo, hi), batched over them: d q-bar, the row kernels and
  // d(dq) form a dependent chain (query_chain); d Wq, d W1 and the bias / vector gradients of the query block are its
  // leaves (query_leaves).  It runs in TWO batches: layers >= 1 as soon as layer 1's d p exists (under layer 0's
  // critical path), layer 0 right after its attention backward (its d p comes from a punish pass of its own).
  auto query_chain = [&](int lo, int hi, hipStream_t on, int first) -> int {
    GemmBatch gd{};   // dqbar[l] = [dp_app | dp_mot] . [Wq_app ; Wq_mot]
    for (int l = lo; l < hi; ++l) {
      GemmProblem p = nn(ws + W.dp_cat + (int64_t)l * B * 2 * D, 2 * D, qp(l, 0) + P.wq, Wd, ws + W.dqbar + (int64_t)l * B * Wd, Wd,
                         B, Wd, 2 * D, 0);
      p.kseg = D;
      p.B.p1 = qp(l, 1) + P.wq;
      p.atomic_ok = 1;                       // 3 output tiles x 48 K-tiles: split K, partials added atomically
      p.accumulate = 1;
      gd.p[gd.n++] = p;
      if (gd.n == kMaxBatch || l == hi - 1) { DVGR_TRY(gemm_launch(on, KC, IC, gd, TAG_GENERIC, prec)); gd.n = 0; }
    }
    QaBwdArgs a{};
    a.dqbar = ws + W.dqbar + (int64_t)lo * B * Wd; a.words = io->words; a.qlen = io->qlen;
    a.params = params + P.q_base + P.q_layer_stride * lo; a.off_w2 = P.qa_w2; a.layer_stride = P.q_layer_stride;
    a.saved = saved + S.layer_stride * lo; a.s_layer_stride = S.layer_stride;
    a.s_d = S.qa_d; a.s_nrm = S.qa_nrm; a.s_a = S.qa_a; a.s_ah = S.qa_ah; a.s_S = S.qa_s;
    a.gwords = g->gwords; a.dr = ws + W.dr + (int64_t)lo * BT; a.dy = ws + W.dy + (int64_t)lo * BT * D;
    a.rowdots = ws + W.qdots + 2 * (int64_t)lo * BT;
    a.B = B; a.T = T; a.D = D; a.Wd = Wd; a.L = hi - lo;
    a.first = first;                         // 1: this batch initialises gwords, 0: it adds to it (same stream: ordered)
    a.dy_p = PM.out(ws + W.dy + (int64_t)lo * BT * D, &plo);
    a.lo_off = plo;
    return qa_bwd(on, a);
  };
  auto query_gemms = [&](int lo, int hi, hipStream_t on) -> int {
    // d(dq) += dy[l] . W1[l] (several problems into ONE output: tiles are added atomically, gdq is zeroed) and
    // dW1[l] = dy[l]^T . dq (+ d b1 = dy[l]^T . 1), all layers, one launch per 6 layers
    const int per = kMaxBatch / 2;
    for (int l0 = lo; l0 < hi; l0 += per) {
      GemmBatch gb{};
      BiasOf bias[kMaxBatch] = {};
      const int l1 = (l0 + per) < hi ? (l0 + per) : hi;
      for (int l = l0; l < l1; ++l) {
        gb.p[gb.n] = nn(ws + W.dy + (int64_t)l * BT * D, D, lp(l) + P.qa_w1, D, g->gdq, D, BT, D, D, 1);
        PM.attach(gb.p[gb.n]);
        gb.p[gb.n].atomic_ok = 1;
        gb.p[gb.n++].atomic_out = (L > 1);
      }
      for (int l = l0; l < l1; ++l) {
        gb.p[gb.n] = tn(ws + W.dy + (int64_t)l * BT * D, D, io->dq, D, glp(l) + P.qa_w1, D, D, D, BT);
        PM.attach(gb.p[gb.n]);
        bias[gb.n++] = BiasOf{ws + W.dy + (int64_t)l * BT * D, BT, D, D, glp(l) + P.qa_b1};
      }
      ColsumBatch cb2{};
      DVGR_TRY(launch_mixed(on, gb, bias, TAG_GENERIC, cb2));
      if (cb2.n) { cb2.chunks = 16; DVGR_TRY(colsum(on, cb2)); }
    }
    return 0;
  };
  // The leaves come in two kinds: d Wq / d bq need only d p (on_p: any stream ordered behind the punish pass that wrote it);
  // d w2 / d b2 need d r, which the chain's row kernels write (on_r: behind the chain).
  // (kinds: bit 0 = the d p leaves, bit 1 = the d r leaves -- a stream handle cannot say "none": the caller's may be the null stream)
  auto query_leaves = [&](int lo, int hi, hipStream_t on_p, hipStream_t on_r, int kinds = 3) -> int {
    ColsumBatch cb{};
    auto flush_cb = [&](hipStream_t on) -> int {
      if (!cb.n) return 0;
      cb.chunks = 16;
      const int e = colsum(on, cb);
      cb.n = 0;
      return e;
    };
    GemmBatch gb{};   // dWq[l][m] = dp[l][m]^T . q[l]
    for (int l = lo; l < hi && (kinds & 1); ++l)
      for (int m = 0; m < 2; ++m) {
        const float* dpc = ws + W.dp_cat + (int64_t)l * B * 2 * D + m * D;
        gb.p[gb.n++] = tn(dpc, 2 * D, ls(l) + S.qa_q, Wd, gqp(l, m) + P.wq, Wd, D, Wd, B);
        if (gb.n == kMaxBatch || (l == hi - 1 && m == 1)) { DVGR_TRY(gemm_launch(on_p, IC, IC, gb, TAG_GENERIC, prec)); gb.n = 0; }
        cb.j[cb.n++] = ColsumJob{dpc, nullptr, nullptr, nullptr, B, 0, D, 2 * D, gqp(l, m) + P.bq};
        if (cb.n == kMaxColsumJobs) DVGR_TRY(flush_cb(on_p));
      }
    if ((kinds & 1) && (on_r != on_p || !(kinds & 2))) DVGR_TRY(flush_cb(on_p));
    if (!(kinds & 2)) return 0;
    for (int l = lo; l < hi; ++l) {
      cb.j[cb.n++] = ColsumJob{ls(l) + S.qa_d, nullptr, ws + W.dr + (int64_t)l * BT, nullptr, BT, 0, D, D, glp(l) + P.qa_w2};
      if (cb.n == kMaxColsumJobs) DVGR_TRY(flush_cb(on_r));
      cb.j[cb.n++] = ColsumJob{ws + W.dr + (int64_t)l * BT, nullptr, nullptr, nullptr, BT, 0, 1, 1, glp(l) + P.qa_b2};
      if (cb.n == kMaxColsumJobs) DVGR_TRY(flush_cb(on_r));
    }
    return flush_cb(on_r);
  };

  // The whole branch of the layers [lo, hi) on the query stream, in dependency order.  Its kernels are small (a few dozen
  // tiles, latency bound), so they run beside the main stream's contractions instead of after them: queued behind the side
  // stream's weight-gradient launches and with its two contractions at the very end of the main stream the branch cost a
  // ~170 us tail after the last dX contraction (rocprofv3 timeline, profiles/r02_v1); on its own stream it ends under
  // layer 0's weight-gradient launch.
  // Order on the query stream: chain and contractions of the layers >= 1 (issued at layer 1, they finish in the gap
  // before layer 0's d p exists), then layer 0's chain and contractions, then the leaves of ALL layers in one small
  // contraction launch and one column-sum launch.
  hipEvent_t leaves_deferred = nullptr;      // set: the query block's leaves go to the side stream at the end of layer 0 (event: chain done)
  hipEvent_t chain0_done = nullptr;          // small batches: layer 0's chain is done (the main stream takes its contractions)
  auto query_branch = [&](int lo, int hi, int first) -> int {
    DVGR_TRY(query_chain(lo, hi, qs, first));
    if (lo == 0 && side && !use_qs && !(O.flags & DVGR_OPT_NO_QUERY_STREAM)) {
      // small batch, no query stream: the branch would end as a serial tail on the side stream after the main stream has
      // gone idle; its two contractions go to the main stream instead (issued at the end), the leaves stay here
      chain0_done = side->next();
      if (!chain0_done) return (int)hipErrorOutOfMemory;
      const hipError_t e = hipEventRecord(chain0_done, qs);
      if (e != hipSuccess) return (int)e;
      return query_leaves(0, L, qs, qs);
    }
    if (lo == 0 && side && qs != hs && !(O.flags & DVGR_OPT_LEAVES_ON_QUERY)) {
      // a query stream of its own: the d Wq / d bq leaves of all layers do not wait for layer 0's chain -- they go to the SIDE
      // stream (ordered behind the same punish pass; enqueued at the end of layer 0, behind its column sums, where that stream
      // goes idle) and run beside the chain instead of ending the step as the last two launches of the query stream (the
      // step's tail was chain + contractions + leaves = 116 us) ...
      // ... and the d w2 / d b2 leaves need the chain but not its two contractions: they follow on the side stream behind an
      // event recorded here, instead of ending the step behind the contractions
      leaves_deferred = side->next();
      if (!leaves_deferred) return (int)hipErrorOutOfMemory;
      const hipError_t e = hipEventRecord(leaves_deferred, qs);
      if (e != hipSuccess) return (int)e;
      return query_gemms(lo, hi, qs);
    }
    DVGR_TRY(query_gemms(lo, hi, qs));
    return lo == 0 ? query_leaves(0, L, qs, qs) : 0;
  };

  // ---- layers, last to first -----------------------------------------------------------------------
  hipEvent_t side_done[2] = {nullptr, nullptr};   // side stream finished the layer that used buffer set 0 / 1
  const bool fold_punish = G == 1 && !(O.flags & DVGR_OPT_NO_FOLDED_PUNISH);
  const bool dp_side = side && fold_punish && !(O.flags & DVGR_OPT_DP_ON_MAIN);
  int fold_layer = -1;                             // the layer whose dX half of the S1 backward the next k_sf_bwd completes
  for (int l = L - 1; l >= 0; --l) {
    const int set = l & 1;
    const int e_last = l * G + G - 1;                 // list entry of the layer's last graph layer: the pair the SF attention saw
    const Keys K = layer_keys(train, seed, l * G);    // graph layer 0 (the part below the graph-layer loop handles j = 0)
    const float* Xin[2] = {l == 0 ? io->A : ms(l - 1, 0) + S.x, l == 0 ? io->M : ms(l - 1, 1) + S.x};
    const float* outc[2] = {io->com_app + e_last * BND, io->com_mot + e_last * BND};
    const float* outs[2] = {io->aq_fusion + e_last * BND, io->mq_fusion + e_last * BND};
    if (side && side_done[set]) {      // layer l + 2 shared this buffer set
      const hipError_t e = hipStreamWaitEvent(st, side_done[set], 0);
      if (e != hipSuccess) return (int)e;
    }
    // S5/S4 backward, row part
    {
      SfBwdArgs a{};
      for (int m = 0; m < 2; ++m) {
        a.dX[m] = dXacc[m];
        a.g_embed[m] = (l == L - 1) ? (m == 0 ? g->g_aq_embed : g->g_mq_embed) : nullptr;
        a.z0[m] = outc[m]; a.z1[m] = outs[m]; a.tn[m] = ms(l, m) + S.tn; a.beta[m] = ms(l, m) + S.beta;
        a.p2[m] = mp(l, m) + P.sf_p2;
        const float* gc = m == 0 ? g->g_com_app : g->g_com_mot;
        const float* gs = m == 0 ? g->g_aq_fusion : g->g_mq_fusion;
        a.g_z0[m] = gc ? gc + e_last * BND : nullptr;
        a.g_z1[m] = gs ? gs + e_last * BND : nullptr;
        a.dw[m] = ws + W.dw[set][m]; a.dpre[m] = lean ? nullptr : ws + W.dpre[set][m];
        a.dz[m] = fuse_dz ? nullptr : ws + W.dz[m];         // (fuse_dz: the fused GAT-view backward forms dz's pre-part itself)
        a.dpre_p[m] = PM.out(ws + W.dpre[set][m], &plo);
        if (fold_layer == l + 1) {         // the layer above left the dX half of its S1 backward to this pass (SfBwdArgs::fin_*)
          const int us = (l + 1) & 1;
          a.fin_dX[m] = dXacc[m];
          a.fin_add[m] = train ? ws + W.dxtmp[m] : nullptr;
          a.fin_ds_part[m] = ws + W.ds_part[us][m];
          a.fin_s[m] = ms(l + 1, m) + S.s;
          a.fin_p[m] = ms(l + 1, m) + S.p;
        }
      }
      a.fin_N = N;
      a.lo_off = plo;
      a.rows = BN; a.D = D;
      DVGR_TRY(sf_bwd(st, a));
    }
    if (fuse_bwd) {
      // dz_v += dpre_v . P1 and the attention backward of graph layer 0 in one launch (dvgr_gatb.hip)
      if (p1t_done) {                        // the transposed planes come from the side stream
        const hipError_t e = hipStreamWaitEvent(st, p1t_done, 0);
        if (e != hipSuccess) return (int)e;
        p1t_done = nullptr;
      }
      GatFusedBwdArgs fa{};
      GatAttnBwdArgs& a = fa.at;
      int64_t alo = -1, lo1 = 0;
      bool ok = true;
      for (int m = 0; m < 2 && ok; ++m) {
        a.dz[m] = fuse_dz ? nullptr : ws + W.dz[m];
        if (fuse_dz) {
          const float* gc = m == 0 ? g->g_com_app : g->g_com_mot;
          const float* gsx = m == 0 ? g->g_aq_fusion : g->g_mq_fusion;
          fa.dx[m] = dXacc[m];
          fa.g_embed[m] = (l == L - 1) ? (m == 0 ? g->g_aq_embed : g->g_mq_embed) : nullptr;
          fa.g_z[m][0] = gc ? gc + (int64_t)l * BND : nullptr;
          fa.g_z[m][1] = gsx ? gsx + (int64_t)l * BND : nullptr;
          fa.beta[m] = ms(l, m) + S.beta;
        }
        a.out_c[m] = (m == 0 ? io->com_app : io->com_mot) + (int64_t)l * BND;
        a.out_s[m] = (m == 0 ? io->aq_fusion : io->mq_fusion) + (int64_t)l * BND;
        a.Wh[m] = ms(l, m) + S.wh; a.s[m] = ms(l, m) + S.s; a.u[m] = ms(l, m) + S.u; a.v[m] = ms(l, m) + S.v;
        a.gat_a[m] = mp(l, m) + P.gat_a; a.gat_c[m] = mp(l, m) + P.gat_c;
        a.dWh[m] = lean ? nullptr : ws + W.dWh[set][m];
        a.ds_part[m] = ws + W.ds_part[set][m];
        a.da_part[m] = ws + W.da_part[set][m];
        a.dc_part[m] = ws + W.dc_part[set][m];
        for (int h = 0; h < 8; ++h) a.key_alpha[m][h] = K.alpha[m][h];
        for (int gi = 0; gi < 2; ++gi) a.key_out[m][gi] = K.out[m][gi];
        a.dWh_p[m] = PM.out(ws + W.dWh[set][m], &plo);
        fa.ah[m] = PM.find(ws + W.dpre[set][m], &lo1);
        ok = ok && fa.ah[m] && (alo < 0 || alo == lo1);
        alo = lo1;
        fa.bh[m] = p1t_hi(l, m);
        ok = ok && fa.bh[m];
        if (O.lo8 && O.p1t && prec == DVGR_BF16X3 && !(O.flags & DVGR_OPT_NO_LO8)) {      // (the pack was made from O.p1t's planes)
          const Lo8Layout Y = lo8_layout(d);
          const uint8_t* base = reinterpret_cast<const uint8_t*>(O.lo8);
          fa.bq[m] = base + Y.pq + (int64_t)(l * 2 + m) * D * D;
          fa.bscale[m] = reinterpret_cast<const float*>(base + Y.ps) + (int64_t)(l * 2 + m) * D;
        }
      }
      if (!ok) return DVGR_E_DIMS;           // (cannot happen: fuse_bwd implies planes of every operand)
      a.lo_off = plo;
      fa.kb64 = (O.flags & DVGR_OPT_KB32) ? 0 : 1;
      fa.no_half = (O.flags & DVGR_OPT_NO_HALF_GAT_TILES) ? 1 : 0;
      fa.a_lo_off = alo; fa.b_lo_off = p1t_lo;
      a.B = B; a.N = N; a.D = D;
      DVGR_TRY(probe.begin(DVGR_PROBE_BWD_SF_DX, l));
      const bool arm_here = l > 0 && !probe.hit(DVGR_PROBE_BWD_SF_DX, l);   // (layer 0: its d p launch follows on this stream, the fork rides on that one; a probed launch: see the forward)
      if (arm_here) DVGR_TRY(arm_fork());
      const int e_gatb = gat_fused_bwd(st, fa, prec);
      if (arm_here) fork_rode = stop_event_rode();
      if (e_gatb) return e_gatb;
      DVGR_TRY(probe.end(DVGR_PROBE_BWD_SF_DX, l));
    } else
    {
      GemmBatch gd{};   // dz_v += dpre_v . P1
      for (int m = 0; m < 2; ++m) {
        gd.p[gd.n] = nn(ws + W.dpre[set][m], D, mp(l, m) + P.sf_p1, D, ws + W.dz[m], D, 2 * BN, D, D, 1);
        PM.attach(gd.p[gd.n++]);
      }
      DVGR_TRY(probe.begin(DVGR_PROBE_BWD_SF_DX, l));
      DVGR_TRY(gemm_launch(st, KC, IC, gd, TAG_BWD_SF_DX, prec));
      DVGR_TRY(probe.end(DVGR_PROBE_BWD_SF_DX, l));
    }
    // S3 backward of graph layer j (list entry l * G + j, GAT module l + j)
    auto attn_bwd = [&](int j) -> int {
      const int e = l * G + j, k = l + j;
      const Keys Kj = layer_keys(train, seed, e);
      GatAttnBwdArgs a{};
      for (int m = 0; m < 2; ++m) {
        a.dz[m] = ws + W.dz[m];
        a.out_c[m] = (m == 0 ? io->com_app : io->com_mot) + e * BND;
        a.out_s[m] = (m == 0 ? io->aq_fusion : io->mq_fusion) + e * BND;
        a.Wh[m] = gs_(l, m, j) + S.wh; a.s[m] = ms(l, m) + S.s; a.u[m] = gs_(l, m, j) + S.u; a.v[m] = gs_(l, m, j) + S.v;
        a.gat_a[m] = gpp_(k, m) + P.gat_a; a.gat_c[m] = gpp_(k, m) + P.gat_c;
        a.dWh[m] = lean ? nullptr : ws + W.dWh[set][m];
        a.ds_part[m] = ws + W.ds_part[set][m] + (int64_t)j * B * kHeads * N;      // one block per graph layer, summed by the punish pass
        a.da_part[m] = ws + W.da_part[set][m];
        a.dc_part[m] = ws + W.dc_part[set][m];
        for (int h = 0; h < 8; ++h) a.key_alpha[m][h] = Kj.alpha[m][h];
        for (int gi = 0; gi < 2; ++gi) a.key_out[m][gi] = Kj.out[m][gi];
        a.dWh_p[m] = PM.out(ws + W.dWh[set][m], &plo);
      }
      a.lo_off = plo;
      a.B = B; a.N = N; a.D = D;
      DVGR_TRY(probe.begin(DVGR_PROBE_BWD_GAT_ATTN, l));
      DVGR_TRY(gat_attn_bwd(st, a));
      return probe.end(DVGR_PROBE_BWD_GAT_ATTN, l);
    };
    // graph_layers > 1 (single stream, fp32 + planes everywhere): graph layers G - 1 .. 1 in full -- attention backward,
    // the gradient of the graph layer's INPUT (= the previous graph layer's specific output) and the weight gradients of
    // GAT module l + j, which other (layer, graph layer) pairs share: the first pair to touch a module overwrites, the
    // later ones accumulate (bias / a / c gradients are added atomically into zeroed ranges either way).
    for (int j = G - 1; j >= 1; --j) {
      const int e = l * G + j, k = l + j;
      const Keys Kj = layer_keys(train, seed, e);
      if (j < G - 1) {                     // dz of this graph layer: its list gradients + what came down from graph layer j + 1
        DzBuildArgs a{};
        for (int m = 0; m < 2; ++m) {
          const float* gc = m == 0 ? g->g_com_app : g->g_com_mot;
          const float* gsx = m == 0 ? g->g_aq_fusion : g->g_mq_fusion;
          a.g_c[m] = gc ? gc + e * BND : nullptr;
          a.g_s[m] = gsx ? gsx + e * BND : nullptr;
          a.in0[m] = ws + W.dspec[m];
          a.in1[m] = train ? ws + W.dxtmp[m] : nullptr;
          a.dz[m] = ws + W.dz[m];
        }
        a.n = BND;
        DVGR_TRY(dz_build(st, a));
      }
      DVGR_TRY(attn_bwd(j));
      {                                    // the small vector gradients of module k
        ColsumBatch cbj{};
        for (int m = 0; m < 2; ++m) {
          cbj.j[cbj.n++] = ColsumJob{ws + W.da_part[set][m], nullptr, nullptr, nullptr, B, 0, kHeads * 2 * F, kHeads * 2 * F, ggp_(k, m) + P.gat_a};
          cbj.j[cbj.n++] = ColsumJob{ws + W.dc_part[set][m], nullptr, nullptr, nullptr, B, 0, kHeads, kHeads, ggp_(k, m) + P.gat_c};
        }
        cbj.chunks = 16;
        DVGR_TRY(colsum(st, cbj));
      }
      {                                    // gradient of this graph layer's input: both GATs' partials, fresh buffers
        GemmBatch gd{};
        if (!train) {
          for (int m = 0; m < 2; ++m) {
            gd.p[gd.n] = nn(ws + W.dWh[set][m], HF, gpp_(k, m) + P.gat_w, D, ws + W.dspec[m], D, BN, D, HF, 0);
            PM.attach(gd.p[gd.n++]);
          }
        } else {
          for (int gi = 0; gi < 2; ++gi)
            for (int m = 0; m < 2; ++m) {
              gd.p[gd.n] = nn(ws + W.dWh[set][m] + gi * 4 * F, HF, gpp_(k, m) + P.gat_w + (int64_t)gi * 4 * F * D, D,
                              gi == 0 ? ws + W.dspec[m] : ws + W.dxtmp[m], D, BN, D, 4 * F, 0, Kj.x[m][gi]);
              PM.attach(gd.p[gd.n++]);
            }
        }
        DVGR_TRY(gemm_launch(st, KC, IC, gd, TAG_BWD_GAT_DX, prec));
      }
      {                                    // dWcat of module k (+ d bcat)
        GemmBatch gb{};
        BiasOf bias[kMaxBatch] = {};
        const int acc = (l < L - 1) ? 1 : 0;             // layer l + 1 already wrote module k (as its graph layer j - 1)
        const float* Xj[2] = {io->aq_fusion + (e - 1) * BND, io->mq_fusion + (e - 1) * BND};   // the graph layer's input
        for (int m = 0; m < 2; ++m) {
          if (!train) {
            gb.p[gb.n] = tn(ws + W.dWh[set][m], HF, Xj[m], D, ggp_(k, m) + P.gat_w, D, HF, D, BN);
            gb.p[gb.n].accumulate = acc;
            PM.attach(gb.p[gb.n]);
            bias[gb.n++] = BiasOf{ws + W.dWh[set][m], BN, HF, HF, ggp_(k, m) + P.gat_b};
          } else {
            for (int gi = 0; gi < 2; ++gi) {
              gb.p[gb.n] = tn(ws + W.dWh[set][m] + gi * 4 * F, HF, gs_(l, m, j) + S.xd + gi * BND, D,
                              ggp_(k, m) + P.gat_w + (int64_t)gi * 4 * F * D, D, 4 * F, D, BN);
              gb.p[gb.n].accumulate = acc;
              PM.attach(gb.p[gb.n]);
              bias[gb.n++] = BiasOf{ws + W.dWh[set][m] + gi * 4 * F, BN, 4 * F, HF, ggp_(k, m) + P.gat_b + gi * 4 * F};
            }
          }
        }
        ColsumBatch cbw{};
        DVGR_TRY(launch_dw(st, gb, bias, TAG_BWD_GAT_DW, cbw));
        if (cbw.n) { cbw.chunks = 16; DVGR_TRY(colsum(st, cbw)); }
      }
    }
    if (G > 1) {                           // graph layer 0's dz: its list gradients + the gradient of graph layer 1's input
      DzBuildArgs a{};
      const int e0 = l * G;
      for (int m = 0; m < 2; ++m) {
        const float* gc = m == 0 ? g->g_com_app : g->g_com_mot;
        const float* gsx = m == 0 ? g->g_aq_fusion : g->g_mq_fusion;
        a.g_c[m] = gc ? gc + e0 * BND : nullptr;
        a.g_s[m] = gsx ? gsx + e0 * BND : nullptr;
        a.in0[m] = ws + W.dspec[m];
        a.in1[m] = train ? ws + W.dxtmp[m] : nullptr;
        a.dz[m] = ws + W.dz[m];
      }
      a.n = BND;
      DVGR_TRY(dz_build(st, a));
    }
    if (!fuse_bwd) DVGR_TRY(attn_bwd(0));
    auto punish = [&](int parts, hipStream_t on) -> int {   // S1 backward: bit 0 = d p (kept per layer for the query branch), bit 1 = dX
      PunishBwdArgs a{};
      for (int m = 0; m < 2; ++m) {
        a.ds_part[m] = ws + W.ds_part[set][m]; a.s[m] = ms(l, m) + S.s; a.p[m] = ms(l, m) + S.p;
        a.X[m] = Xin[m]; a.dX[m] = dXacc[m];
        a.add[m] = (train && (parts & 2)) ? ws + W.dxtmp[m] : nullptr;
      }
      a.dp_cat = ws + W.dp_cat + (int64_t)l * B * 2 * D; a.B = B; a.N = N; a.D = D; a.parts = parts;
      a.zero = (parts & 1) ? ws + W.dqbar + (int64_t)l * B * Wd : nullptr; a.zero_cols = Wd;   // dq-bar accumulates split-K partials
      a.copies = G;
      return punish_bwd(on, a);
    };
    if (l == 0) {                               // layer 0 is the last one: its query chain must not wait for the dX chain (both other streams read its d p: it stays in front of their fork)
      DVGR_TRY(arm_fork());
      const int e_p = punish(1, st);
      fork_rode = stop_event_rode();
      if (e_p) return e_p;
    }
    // dw, the attention's small partials and (layer 0) d p are final: the side stream takes the column sums of the small
    // vector gradients (p2, a, c) and layer 0's query chain.  The side stream carries NO large contraction: two
    // contractions at once only slow each other down (both are bound by the compute units' global -> LDS path; measured:
    // a weight-gradient launch of 35 us alone took 95-105 us beside the critical path's kernels and slowed those by a
    // third), so every 240-tile contraction stays on the main stream and the side stream fills in latency-bound kernels.
    if (l == 0) DVGR_TRY(to_side_and_query()); else DVGR_TRY(to_side());
    // d p of a layer >= 1 has one reader, the query branch on the query / side stream: its launch goes to the side stream as well
    // (round 6; nothing on the caller's stream waits for it -- the dX half of the S1 backward is the next layer's k_sf_bwd)
    if (l >= 1 && dp_side) DVGR_TRY(punish(1, hs));
    ColsumBatch cb{};
    for (int m = 0; m < 2; ++m) {
      cb.j[cb.n++] = ColsumJob{ms(l, m) + S.tn, nullptr, ws + W.dw[set][m], nullptr, 2 * BN, 0, D, D, gmp(l, m) + P.sf_p2};
      cb.j[cb.n++] = ColsumJob{ws + W.da_part[set][m], nullptr, nullptr, nullptr, B, 0, kHeads * 2 * F, kHeads * 2 * F, gmp(l, m) + P.gat_a};
      cb.j[cb.n++] = ColsumJob{ws + W.dc_part[set][m], nullptr, nullptr, nullptr, B, 0, kHeads, kHeads, gmp(l, m) + P.gat_c};
    }
    // Layer 0 with the query branch folded into the side stream (a reducer is running, DVGR_OPT_NO_QUERY_STREAM): the
    // layer's column sums go out BEFORE the branch and get an event of their own, so that the layer block's event -- and
    // with it the block's all-reduce -- does not wait for the query branch queued behind them (it did: both last buckets
    // became ready together, ~100 us after the main stream's last kernel).
    hipEvent_t colsums_done = nullptr;
    if (l == 0 && side && qs == hs) {
      cb.chunks = 16;
      DVGR_TRY(colsum(hs, cb));
      cb.n = 0;
      colsums_done = side->next();
      if (!colsums_done) return (int)hipErrorOutOfMemory;
      const hipError_t e3 = hipEventRecord(colsums_done, hs);
      if (e3 != hipSuccess) return (int)e3;
    }
    if (l == 0) DVGR_TRY(query_branch(0, 1, L == 1 ? 1 : 0));
    // S2 backward on the critical path: dX += drop_x(dWh . Wcat)
    {
      GemmBatch gd{};
      if (!train) {
        for (int m = 0; m < 2; ++m) {
          gd.p[gd.n] = nn(ws + W.dWh[set][m], HF, mp(l, m) + P.gat_w, D, dXacc[m], D, BN, D, HF, 1);
          PM.attach(gd.p[gd.n++]);
        }
      } else {
        // the two GATs dropped x with different masks: the common GAT's partial accumulates into dX, the specific GAT's
        // goes to a scratch tensor in the same launch (two tiles may not read-modify-write one output), and the punish
        // backward, which rewrites every element of dX anyway, folds it in
        for (int gi = 0; gi < 2; ++gi)
          for (int m = 0; m < 2; ++m) {
            gd.p[gd.n] = nn(ws + W.dWh[set][m] + gi * 4 * F, HF, mp(l, m) + P.gat_w + (int64_t)gi * 4 * F * D, D,
                            gi == 0 ? dXacc[m] : ws + W.dxtmp[m], D, BN, D, 4 * F, gi == 0 ? 1 : 0, K.x[m][gi]);
            PM.attach(gd.p[gd.n++]);
          }
      }
      DVGR_TRY(probe.begin(DVGR_PROBE_BWD_GAT_DX, l));
      DVGR_TRY(gemm_launch(st, KC, IC, gd, TAG_BWD_GAT_DX, prec));
      DVGR_TRY(probe.end(DVGR_PROBE_BWD_GAT_DX, l));
    }
    // S1 backward: the dX update (folding the specific GAT's partial of the training-mode S2 backward in) and, for the
    // layers > 0, d p in the same launch
    // Layers >= 1 (graph_layers 1): only d p here; the dX half -- dX += dtau p + the specific GAT's partial -- is done by the NEXT
    // layer's k_sf_bwd on the rows it reads anyway (round 6: one kernel of ~20 us fewer on the critical path per layer >= 1).
    // Layer 0's dX is the call's output: its dX half stays a launch of its own.
    if (l >= 1 && fold_punish) { if (!dp_side) DVGR_TRY(punish(1, st)); fold_layer = l; }
    else DVGR_TRY(punish(l == 0 ? 2 : 3, st));   // (layer 0's dX half on the side stream beside its weight-gradient launch: C2 +2.8 %, C3 -1.0 %, C4 -0.4 %: not kept)
    if (l == 1) {                           // every d p of the layers >= 1 exists: their query branch, under layer 0
      // (with d p on the side stream this fork of the caller's stream orders nothing the branch needs -- but without it the branch
      // starts one contraction earlier, beside the dX launch, and C2 is 1.4 % slower: it stays)
      DVGR_TRY(to_query());
      if (dp_side && qs != hs) DVGR_TRY(side->order(hs, qs));     // (the d p of the layers >= 1 were written on the side stream)
      DVGR_TRY(query_branch(1, L, 1));
    }
    // every weight gradient of the layer as ONE launch: dP1 = sum_v dpre_v^T . z_v (two row segments) and
    // dWcat = dWh^T . drop_x(X), both modalities, with the bias gradients d bP and d bcat as row sums of their dY operands
    {
      GemmBatch gb{};
      BiasOf bias[kMaxBatch] = {};
      for (int m = 0; m < 2; ++m) {
        GemmProblem p = tn(ws + W.dpre[set][m], D, outc[m], D, gmp(l, m) + P.sf_p1, D, D, D, 2 * BN);
        p.kseg = BN;
        p.A.p1 = ws + W.dpre[set][m] + BND;
        p.B.p1 = outs[m];
        PM.attach(p);
        bias[gb.n] = BiasOf{ws + W.dpre[set][m], 2 * BN, D, D, gmp(l, m) + P.sf_bp};
        gb.p[gb.n++] = p;
      }
      for (int m = 0; m < 2; ++m) {
        if (!train) {
          gb.p[gb.n] = tn(ws + W.dWh[set][m], HF, Xin[m], D, gmp(l, m) + P.gat_w, D, HF, D, BN);
          PM.attach(gb.p[gb.n]);
          bias[gb.n++] = BiasOf{ws + W.dWh[set][m], BN, HF, HF, gmp(l, m) + P.gat_b};
        } else {
          for (int gi = 0; gi < 2; ++gi) {
            gb.p[gb.n] = tn(ws + W.dWh[set][m] + gi * 4 * F, HF, ms(l, m) + S.xd + gi * BND, D,
                            gmp(l, m) + P.gat_w + (int64_t)gi * 4 * F * D, D, 4 * F, D, BN);
            PM.attach(gb.p[gb.n]);
            bias[gb.n++] = BiasOf{ws + W.dWh[set][m] + gi * 4 * F, BN, 4 * F, HF, gmp(l, m) + P.gat_b + gi * 4 * F};
          }
        }
      }
      // Where: the launch is unbalanced (the dP1 tiles reduce over 2 B N rows, the dWcat tiles over B N: 72 tiles run twice
      // as long as the other 144), so beside the next layer's critical-path kernels it fills compute units that would idle
      // behind its short tiles -- the layers > 0 put it on the side stream (measured: 1.245 ms per step against 1.272 with
      // it on the main stream, and 1.27 merged with the dX launch); layer 0 has no next layer and keeps it on the main
      // stream, which is idle by then.
#ifdef DVGR_TIMING_AIDS
      static const bool tn_main = []() { const char* e = getenv("DVGR_TN_MAIN"); return e && e[0] == '1'; }();   // A/B aid
#else
      constexpr bool tn_main = false;
#endif
      hipStream_t dws = (side && l > 0 && !tn_main) ? hs : st;      // (beside a reducer as well: on the main stream 1.37-1.40 against 1.32 ms)      // (layer 0's on the side stream too: no difference, 1.179 ms both)
      const Probe dprobe{O, dws};
      // (cannot happen: lean_planes() implies the planes kernels; under it dpre / dWh have no fp32 copy for a column sum)
      if (lean && !gemm_rowsum_supported(IC, IC, gb, prec)) return DVGR_E_DIMS;
      DVGR_TRY(dprobe.begin(DVGR_PROBE_BWD_GAT_DW, l));
      DVGR_TRY(launch_dw(dws, gb, bias, TAG_BWD_GAT_DW, cb));     // (biases the engine cannot carry join the column sums)
      DVGR_TRY(dprobe.end(DVGR_PROBE_BWD_GAT_DW, l));
      if (cb.n) { cb.chunks = 16; DVGR_TRY(colsum(hs, cb)); }
      if (side) {
        side_done[set] = side->next();
        if (!side_done[set]) return (int)hipErrorOutOfMemory;
        const hipError_t e2 = hipEventRecord(side_done[set], hs);
        if (e2 != hipSuccess) return (int)e2;
      }
      if ((layer_done_events && layer_done_events[l]) || O.comm) {
        if (dws == hs || !side) {
          DVGR_TRY(layer_done(l, hs));                   // every gradient of block l was written on the side stream
        } else if (colsums_done && !cb.n) {              // (layer 0 beside a reducer: only the column sums, see above)
          const hipError_t e4 = hipStreamWaitEvent(st, colsums_done, 0);
          if (e4 != hipSuccess) return (int)e4;
          DVGR_TRY(layer_done(l, st));
        } else {                                         // block l is written on both streams: the event is recorded on the
          DVGR_TRY(to_main());                           // main stream behind the side stream's column sums
          DVGR_TRY(layer_done(l, st));
        }
      }
    }
  }

  if (leaves_deferred) {
    DVGR_TRY(query_leaves(0, L, hs, hs, 1));
    const hipError_t e = hipStreamWaitEvent(hs, leaves_deferred, 0);
    if (e != hipSuccess) return (int)e;
    DVGR_TRY(query_leaves(0, L, hs, hs, 2));
  }
  if (chain0_done) {
    const hipError_t e = hipStreamWaitEvent(st, chain0_done, 0);
    if (e != hipSuccess) return (int)e;
    DVGR_TRY(query_gemms(0, 1, st));
  }
  // ---- the caller's stream sees every gradient: it waits for the side stream and the query stream
  DVGR_TRY(to_main());
  if (side && qs != hs) DVGR_TRY(side->order(qs, st));
  DVGR_TRY(layer_done(L + 1, st));        // the query block
  return 0;
}


int dvgr_scope_u_loss(const dvgr_dims* dp, const dvgr_io* io, float alpha, float beta, float dep_scale,
                      float* loss3, float* g_vis, float* g_com_app, float* g_com_mot, float* g_aq_fusion,
                      float* g_mq_fusion, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dp || !io || !loss3 || !workspace || !g_com_app || !g_com_mot || !g_aq_fusion || !g_mq_fusion) return DVGR_E_NULL;
  if (!io->com_app || !io->com_mot || !io->aq_fusion || !io->mq_fusion) return DVGR_E_NULL;
  DVGR_TRY(dvgr_check_dims(dp));
  if (workspace_bytes < aux_ws_floats(*dp) * sizeof(float)) return DVGR_E_WORKSPACE;
  float* const g[4] = {g_com_app, g_com_mot, g_aq_fusion, g_mq_fusion};
  return aux_losses((hipStream_t)stream, *dp, io, alpha, beta, dep_scale, loss3, g_vis, g, reinterpret_cast<float*>(workspace));
}

}  // extern "C"

// =================================================================================================
// Test hooks (include/dvgr_debug.h)
// =================================================================================================
#include "../../include/dvgr_debug.h"

namespace dvgr { int keep_mask_launch(hipStream_t st, uint32_t key, int64_t n, float* keep); }

extern "C" {

int dvgr_dbg_gemm(int mode, int M, int N, int K, int kseg, const float* A, const float* A2, int64_t lda,
                  const float* B, const float* B2, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                  int accumulate, uint32_t drop_key_a, uint32_t drop_key_b, uint32_t drop_key_c, int prec,
                  void* stream) {
  if (!A || !B || !C) return DVGR_E_NULL;
  GemmBatch gb{};
  GemmProblem p{};
  p.A = operand(A, lda, drop_key_a);
  p.B = operand(B, ldb, drop_key_b);
  if (A2) p.A.p1 = A2;
  if (B2) p.B.p1 = B2;
  p.C = C; p.ldc = ldc; p.bias = bias; p.M = M; p.N = N; p.K = K; p.kseg = (kseg > 0 && kseg < K) ? kseg : K;
  p.act = act & 15; p.dbg = act >> 4; p.accumulate = accumulate; p.drop_key_c = drop_key_c;
  p.atomic_ok = accumulate;   // an accumulating output may be produced by split-K partial tiles
  gb.p[gb.n++] = p;
  const int la = mode == 2 ? IC : KC, lb = mode == 0 ? KC : IC;
  return gemm_launch((hipStream_t)stream, la, lb, gb, TAG_GENERIC, prec);
}

int dvgr_dbg_split(const float* src, void* hi, int64_t lo_off, int64_t n, void* stream) {
  if (!src || !hi) return DVGR_E_NULL;
  SplitBatch sb{};
  sb.lo_off = lo_off;
  sb.j[sb.n++] = SplitJob{src, reinterpret_cast<uint16_t*>(hi), n};
  return split_planes((hipStream_t)stream, sb);
}

int dvgr_dbg_gemm_planes(int mode, int M, int N, int K, const float* A, const void* A_hi, int64_t lda, int64_t a_lo_off,
                         const float* B, const void* B_hi, int64_t ldb, int64_t b_lo_off, float* C, int64_t ldc, int act,
                         int accumulate, int prec, void* stream) {
  if (!A || !B || !C || !A_hi || !B_hi) return DVGR_E_NULL;
  GemmBatch gb{};
  GemmProblem p{};
  p.A = operand(A, lda);
  p.B = operand(B, ldb);
  p.A.h0 = p.A.h1 = reinterpret_cast<const uint16_t*>(A_hi); p.A.lo_off = a_lo_off;
  p.B.h0 = p.B.h1 = reinterpret_cast<const uint16_t*>(B_hi); p.B.lo_off = b_lo_off;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.kseg = K;
  p.act = act & 15; p.dbg = act >> 4; p.accumulate = accumulate; p.atomic_ok = accumulate;
  gb.p[gb.n++] = p;
  const int la = mode == 2 ? IC : KC, lb = mode == 0 ? KC : IC;
  return gemm_launch((hipStream_t)stream, la, lb, gb, TAG_GENERIC, prec);
}

// timing aid (tools/gemm_shapes.py): `count` copies of one pre-split problem in ONE launch (outputs count * M rows apart),
// as the unit's launch sites batch the modalities / views; kseg < K selects the two-segment reduction (segment 1 = A2 / B2)
int dvgr_dbg_gemm_planes_batch(int mode, int M, int N, int K, int kseg, const float* A, const void* A_hi, int64_t lda,
                               int64_t a_lo_off, const float* B, const void* B_hi, int64_t ldb, int64_t b_lo_off, float* C,
                               int64_t ldc, int act, int accumulate, int atomic_ok, int prec, int count, int tag, float* rowsum,
                               int same_output, void* stream) {
  if (!A || !B || !C || !A_hi || !B_hi) return DVGR_E_NULL;
  if (count < 1 || count > kMaxBatch) return DVGR_E_DIMS;
  GemmBatch gb{};
  for (int i = 0; i < count; ++i) {
    GemmProblem p{};
    p.A = operand(A, lda);
    p.B = operand(B, ldb);
    p.A.h0 = p.A.h1 = reinterpret_cast<const uint16_t*>(A_hi); p.A.lo_off = a_lo_off;
    p.B.h0 = p.B.h1 = reinterpret_cast<const uint16_t*>(B_hi); p.B.lo_off = b_lo_off;
    p.C = same_output ? C : C + (int64_t)i * M * ldc;
    p.atomic_out = same_output;            // every problem adds its tiles into ONE pre-zeroed output
    p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.kseg = (kseg > 0 && kseg < K) ? kseg : K;
    if (p.kseg < K) {      // second segment: the same rows again (timing only)
      if (mode == 2) { p.A.p1 = A; p.A.h1 = p.A.h0; }
      p.B.p1 = B; p.B.h1 = p.B.h0;
    }
    p.act = act & 15; p.dbg = act >> 4; p.accumulate = accumulate; p.atomic_ok = atomic_ok;
    gb.p[gb.n++] = p;
  }
  const int la = mode == 2 ? IC : KC, lb = mode == 0 ? KC : IC;
  if (rowsum) {
    if (!gemm_rowsum_supported(la, lb, gb, prec)) return DVGR_E_DIMS;
    for (int i = 0; i < count; ++i) gb.p[i].rowsum = rowsum + (int64_t)i * M;
  }
  return gemm_launch((hipStream_t)stream, la, lb, gb, tag, prec);
}

int dvgr_dbg_colsum(const float* M, const float* w, int rows, int cols, int64_t ld, float* out, void* stream) {
  if (!M || !out) return DVGR_E_NULL;
  ColsumBatch cb{};
  cb.j[cb.n++] = ColsumJob{M, nullptr, w, nullptr, rows, 0, cols, ld, out};
  return colsum((hipStream_t)stream, cb);
}

int dvgr_dbg_keep_mask(uint32_t key, int64_t n, float* keep, void* stream) {
  if (!keep) return DVGR_E_NULL;
  return keep_mask_launch((hipStream_t)stream, key, n, keep);
}

int dvgr_dbg_event_create(void** event) {
  if (!event) return DVGR_E_NULL;
  hipEvent_t e = nullptr;
  const hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
  *event = (void*)e;
  return (int)r;
}
int dvgr_dbg_event_elapsed_ms(void* begin, void* end, float* ms) {
  if (!begin || !end || !ms) return DVGR_E_NULL;
  hipError_t r = hipEventSynchronize((hipEvent_t)end);
  if (r == hipSuccess) r = hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end);
  return (int)r;
}
int dvgr_dbg_event_destroy(void* event) { return event ? (int)hipEventDestroy((hipEvent_t)event) : 0; }

long long dvgr_dbg_launch_count(void) { return dvgr::g_launch_count.load(std::memory_order_relaxed); }

int dvgr_dbg_xmap_tags(int mask) { return dvgr::gemm_xmap_tags(mask); }

int dvgr_dbg_plan_xcd_blocks(int n, const int* M, const int* N, const int* K, int bm, int bn, int max_per_xcd, int* seg_out) {
  return dvgr::gemm_plan_xcd_blocks(n, M, N, K, bm, bn, max_per_xcd, seg_out);
}

uint32_t dvgr_dbg_site_key(uint64_t seed, int layer, int mod, int gat, int which) {
  return site_key(seed, drop_site(layer, mod, gat, which));
}

}  // extern "C"
