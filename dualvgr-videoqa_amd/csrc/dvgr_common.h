// dvgr_common.h -- layouts (host) and device helpers shared by every translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#ifdef DVGR_TIMING_AIDS
#include <stdlib.h>
#include <string.h>
#endif

#include "../../include/dvgr.h"

namespace dvgr {

// Every kernel launch of the library goes through DVGR_LAUNCH: one relaxed atomic increment per launch, read back by
// dvgr_dbg_launch_count() (include/dvgr_debug.h) -- bench.py reports launches per step from it.
extern std::atomic<long long> g_launch_count;
#ifdef DVGR_TIMING_AIDS
// knock-out timing aid (tools/knockout.sh): DVGR_SKIP=<substring>[,<substring>...] -- launches whose kernel expression contains
// one of them are not enqueued (results are garbage; the step's time without a kernel is that kernel's cost IN the step)
inline bool launch_skipped(const char* what) {
  static const char* env = getenv("DVGR_SKIP");
  if (!env || !*env) return false;
  const char* p = env;
  while (*p) {
    const char* q = p;
    while (*q && *q != ',') ++q;
    const size_t n = (size_t)(q - p);
    if (n) for (const char* w = what; *w; ++w) if (!strncmp(w, p, n)) return true;
    p = *q ? q + 1 : q;
  }
  return false;
}
#define DVGR_LAUNCH_FIRST(first, ...) #first
#define DVGR_LAUNCH(...)                                                    \
  do {                                                                      \
    if (::dvgr::launch_skipped(DVGR_LAUNCH_FIRST(__VA_ARGS__))) break;      \
    ::dvgr::g_launch_count.fetch_add(1, std::memory_order_relaxed);         \
    hipLaunchKernelGGL(__VA_ARGS__);                                        \
  } while (0)
#else
#define DVGR_LAUNCH(...)                                                    \
  do {                                                                      \
    ::dvgr::g_launch_count.fetch_add(1, std::memory_order_relaxed);         \
    hipLaunchKernelGGL(__VA_ARGS__);                                        \
  } while (0)
#endif

// A fork right behind ONE kernel (the side stream continues after it) need not put an event-record packet between that kernel and
// the next one of its stream: the event can ride on the kernel's own dispatch (hipExtLaunchKernelGGL's stopEvent).  Measured
// (tools/micro/fork_cost.hip, profiles/r06_fork_cost.txt): +1.7 us per fork for the forking stream instead of +2.6.  Protocol, one
// thread: arm_stop(event, stream) -> call a launch function whose launch is a DVGR_LAUNCH_FORKABLE on that stream ->
// stop_event_rode() says whether the event went out with it (else the caller records it the ordinary way: the launch was a
// different one, was skipped, or the function launched nothing).
struct ArmedStop { hipEvent_t ev; hipStream_t st; bool used; };
inline thread_local ArmedStop g_armed_stop{nullptr, nullptr, false};
inline void arm_stop(hipEvent_t e, hipStream_t s) { g_armed_stop = ArmedStop{e, s, false}; }
inline bool stop_event_rode() { const bool u = g_armed_stop.used; g_armed_stop = ArmedStop{nullptr, nullptr, false}; return u; }
// (the arguments must have the kernel's parameter types exactly: hipExtLaunchKernelGGL deduces them from both sides)
#ifdef DVGR_TIMING_AIDS
#define DVGR_LAUNCH_FORKABLE(kern, grid, block, lds, stream, ...)                                                        \
  do {                                                                                                                   \
    if (::dvgr::launch_skipped(#kern)) break;                                                                            \
    ::dvgr::g_launch_count.fetch_add(1, std::memory_order_relaxed);                                                      \
    if (::dvgr::g_armed_stop.ev && !::dvgr::g_armed_stop.used && ::dvgr::g_armed_stop.st == (stream)) {                  \
      hipExtLaunchKernelGGL(kern, grid, block, lds, stream, nullptr, ::dvgr::g_armed_stop.ev, 0, __VA_ARGS__);                   \
      ::dvgr::g_armed_stop.used = true;                                                                                  \
    } else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                                      \
  } while (0)
#else
#define DVGR_LAUNCH_FORKABLE(kern, grid, block, lds, stream, ...)                                                        \
  do {                                                                                                                   \
    ::dvgr::g_launch_count.fetch_add(1, std::memory_order_relaxed);                                                      \
    if (::dvgr::g_armed_stop.ev && !::dvgr::g_armed_stop.used && ::dvgr::g_armed_stop.st == (stream)) {                  \
      hipExtLaunchKernelGGL(kern, grid, block, lds, stream, nullptr, ::dvgr::g_armed_stop.ev, 0, __VA_ARGS__);                   \
      ::dvgr::g_armed_stop.used = true;                                                                                  \
    } else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                                      \
  } while (0)
#endif

constexpr int kHeads = 8;          // 4 common-GAT heads + 4 specific-GAT heads per modality
constexpr int kMfbHidden = 512;    // mm_dim * factor   (model/fusions/fusions.py:384-395 defaults)
constexpr int kMfbPooled = 256;    // mm_dim
constexpr float kDropP = 0.15f;    // model/models.py:95-100
constexpr float kKeepScale = 1.0f / (1.0f - 0.15f);
constexpr uint32_t kDropThresh16 = 9830;  // floor(0.15 * 65536): keep iff 16-bit field >= this
constexpr float kSlope = 0.01f;    // LeakyReLU slope, model/models.py:95-100

// ---------------------------------------------------------------------------------------------
// Flat parameter layout.  Every tensor starts at a multiple of 64 floats.
//   [ layer 0 | layer 1 | ... | layer L-1 | MFB | query block | extra GAT sets (graph_layers > 1 only) ]
//   layer block  = per modality: Wcat, bcat, a, c (the two punish-GATs, 8 heads), P1, bP, p2 (SF attention)
//   query block  = per layer: QueryAttn W1, b1, w2, b2; per modality: QueryPunish Wq, bq
//   extra GAT set = per modality: Wcat, bcat, a, c of GAT module index L, L+1, ..., L*G-1.  The reference builds L*G
//                  modules per family and addresses module [i + j] in unit layer i, graph layer j (model/models.py:150-160),
//                  so modules 0 .. L+G-2 are used (module k by every (i, j) with i + j = k: they SHARE it) and the rest only
//                  exist in the state dict.  Module k < L lives in layer block k; the others in the extra region.
// The query-side parameters sit in a block of their own because their gradients become final together, at the very end
// of the backward (the query branch of ALL layers runs as one batch of launches once every layer's d p is known), while
// a layer block is final as soon as that layer's weight-gradient contractions are: the blocks are the buckets of the
// data-parallel gradient all-reduce (dvgr_param_layer_range).
// ---------------------------------------------------------------------------------------------
struct ParamLayout {
  int64_t layer_stride;            // floats per layer block
  int64_t mod_stride;              // floats per modality block inside a layer block
  int64_t gat_w, gat_b, gat_a, gat_c, sf_p1, sf_bp, sf_p2;          // offsets inside a modality block
  int64_t mfb_base, mfb_w0, mfb_b0, mfb_w1, mfb_b1, mfb_wo, mfb_bo;  // absolute offsets
  int64_t q_base, q_layer_stride, q_mod_base, q_mod_stride;          // query block
  int64_t qa_w1, qa_b1, qa_w2, qa_b2;                                // offsets inside a query layer block
  int64_t wq, bq;                                                    // offsets inside a query modality block
  int64_t extra_base, extra_stride, extra_mod_stride;                // extra GAT sets (module indices >= L), G > 1
  int64_t total;
};

__host__ __device__ inline int64_t round64(int64_t x) { return (x + 63) / 64 * 64; }

inline ParamLayout param_layout(const dvgr_dims& d) {
  ParamLayout p{};
  const int64_t D = d.D, Wd = d.Wd, F = d.D / 4;
  int64_t m = 0;
  p.gat_w = m; m += round64(kHeads * F * D);
  p.gat_b = m; m += round64(kHeads * F);
  p.gat_a = m; m += round64(kHeads * 2 * F);
  p.gat_c = m; m += 64;
  p.sf_p1 = m; m += round64(D * D);
  p.sf_bp = m; m += round64(D);
  p.sf_p2 = m; m += round64(D);
  p.mod_stride = m;
  p.layer_stride = 2 * m;
  p.mfb_base = p.layer_stride * d.L;
  int64_t q = p.mfb_base;
  p.mfb_w0 = q; q += round64((int64_t)kMfbHidden * D);
  p.mfb_b0 = q; q += round64(kMfbHidden);
  p.mfb_w1 = q; q += round64((int64_t)kMfbHidden * D);
  p.mfb_b1 = q; q += round64(kMfbHidden);
  p.mfb_wo = q; q += round64(D * (int64_t)kMfbPooled);
  p.mfb_bo = q; q += round64(D);
  p.q_base = q;
  int64_t o = 0;
  p.qa_w1 = o; o += round64(D * D);
  p.qa_b1 = o; o += round64(D);
  p.qa_w2 = o; o += round64(D);
  p.qa_b2 = o; o += 64;
  p.q_mod_base = o;
  int64_t w = 0;
  p.wq = w; w += round64(D * Wd);
  p.bq = w; w += round64(D);
  p.q_mod_stride = w;
  p.q_layer_stride = o + 2 * w;
  p.extra_base = p.q_base + p.q_layer_stride * d.L;
  p.extra_mod_stride = p.sf_p1;                 // Wcat, bcat, a, c: the GAT part of a modality block
  p.extra_stride = 2 * p.extra_mod_stride;
  const int G = d.G > 1 ? d.G : 1;
  p.total = p.extra_base + p.extra_stride * (int64_t)d.L * (G - 1);
  return p;
}
// float offset of GAT module `k` (0 <= k < L*G), modality `mod`: the base its gat_w / gat_b / gat_a / gat_c offsets add to
inline int64_t gat_set_base(const ParamLayout& p, int L, int k, int mod) {
  return k < L ? p.layer_stride * k + p.mod_stride * mod : p.extra_base + p.extra_stride * (k - L) + p.extra_mod_stride * mod;
}

// ---------------------------------------------------------------------------------------------
// Saved-for-backward layout (float offsets).
// ---------------------------------------------------------------------------------------------
struct SavedLayout {
  int64_t layer_stride, mod_stride, mod_base;
  int64_t qa_d, qa_nrm, qa_a, qa_ah, qa_s, qa_q;              // inside a layer block
  int64_t p, s, x, wh, u, v, tn, beta, xd, zc, zs;            // inside a modality block (wh .. zs: graph layer 0)
  int64_t g_stride;                                           // graph layer j > 0: the same six offsets + j * g_stride
  int64_t mfb_base, mfb_x0, mfb_x1, mfb_z, in_a, in_m, in_dq; // absolute
  int64_t total;                                              // floats; the bf16 planes region follows (2 x total x 2 bytes)
};

inline SavedLayout saved_layout(const dvgr_dims& d) {
  SavedLayout s{};
  const int64_t B = d.B, N = d.N, T = d.T, D = d.D, Wd = d.Wd, F = d.D / 4;
  int64_t o = 0;
  s.qa_d = o;   o += round64(B * T * D);
  s.qa_nrm = o; o += round64(B * T);
  s.qa_a = o;   o += round64(B * T);
  s.qa_ah = o;  o += round64(B * T);
  s.qa_s = o;   o += round64(B);
  s.qa_q = o;   o += round64(B * Wd);
  s.mod_base = o;
  int64_t m = 0;
  s.p = m;    m += round64(B * D);
  s.s = m;    m += round64(B * N);
  s.x = m;    m += round64(B * N * D);
  s.tn = m;   m += round64(2 * B * N * D);
  s.beta = m; m += round64(B * N * 2);
  const int64_t g0 = m;                      // per graph layer (G of them; every reference config has G = 1):
  s.wh = m;   m += round64(B * N * kHeads * F);
  s.u = m;    m += round64(B * kHeads * N);
  s.v = m;    m += round64(B * kHeads * N);
  s.xd = m;   m += round64(2 * B * N * D);   // training mode: x * keep/(1-p) for the common / specific GAT
  s.zc = m;   m += round64(B * N * D);       // plane-only slots: bf16 hi/lo of the common / specific GAT outputs
  s.zs = m;   m += round64(B * N * D);       //   (their fp32 values live in the caller's list tensors)
  s.g_stride = m - g0;
  m += s.g_stride * ((d.G > 1 ? d.G : 1) - 1);
  s.mod_stride = m;
  s.layer_stride = o + 2 * m;
  s.mfb_base = s.layer_stride * d.L;
  int64_t q = s.mfb_base;
  s.mfb_x0 = q; q += round64(B * N * kMfbHidden);
  s.mfb_x1 = q; q += round64(B * N * kMfbHidden);
  s.mfb_z = q;  q += round64(B * N * kMfbPooled);
  s.in_a = q;   q += round64(B * N * D);     // plane-only slots: bf16 hi/lo of the inputs A, M, dq
  s.in_m = q;   q += round64(B * N * D);
  s.in_dq = q;  q += round64(B * T * D);
  s.total = q;
  return s;
}

// ---------------------------------------------------------------------------------------------
// Dropout: counter-based, stateless.  site = ((layer*2 + mod)*2 + gat)*6 + {0: x, 1..4: alpha of
// head 0..3, 5: out}; `key` = site_key(seed, site) is computed on the host and passed by value.
// keep(key, idx): one 32-bit hash serves two consecutive elements (16-bit fields).
// dualvgr-videoqa_amd/dropout.py replicates this bit for bit in numpy.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
inline uint32_t site_key(uint64_t seed, uint32_t site) {
  uint32_t k = mix32(site + 0x9E3779B1u);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k ^ (uint32_t)seed);
  return k | 1u;  // never 0: key 0 means "no dropout at this site"
}
inline uint32_t drop_site(int layer, int mod, int gat, int which) {
  return (uint32_t)(((layer * 2 + mod) * 2 + gat) * 6 + which);
}
// keep iff the element's 16-bit field >= thresh (= floor(p * 65536))
__device__ __forceinline__ bool drop_keep_t(uint32_t key, uint32_t idx, uint32_t thresh) {
  const uint32_t r = mix32(((idx >> 1) * 0x9E3779B1u) ^ key);
  const uint32_t f = (idx & 1u) ? (r >> 16) : (r & 0xFFFFu);
  return f >= thresh;
}
__device__ __forceinline__ bool drop_keep(uint32_t key, uint32_t idx) { return drop_keep_t(key, idx, kDropThresh16); }
// scaled keep factor: 1/(1-p) or 0; key == 0 -> 1 (eval mode)
__device__ __forceinline__ float drop_factor(uint32_t key, uint32_t idx) {
  if (key == 0u) return 1.0f;
  return drop_keep(key, idx) ? kKeepScale : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Small device helpers
// ---------------------------------------------------------------------------------------------
// Wave-wide reductions on the DPP cross-lane path (VALU speed; the ds_bpermute butterfly that __shfl_xor compiles
// to costs ~100 cycles per step).  Quads, then rows of 16, then row_bcast 15 / 31 carry the partials up to lane 63;
// v_readlane broadcasts the result to every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xb1, 0xf>(0.f, v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4e, 0xf>(0.f, v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x114, 0xf>(0.f, v);   // row_shr 4
  v += dpp_move<0x118, 0xf>(0.f, v);   // row_shr 8
  v += dpp_move<0x142, 0xa>(0.f, v);   // row_bcast 15 -> rows 1, 3
  v += dpp_move<0x143, 0xc>(0.f, v);   // row_bcast 31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_move<0xb1, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x4e, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x114, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x118, 0xf>(ninf, v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(ninf, v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float elu_f(float x) { return x > 0.0f ? x : expm1f(x); }
// ELU on the hardware exponential (absolute error ~1e-7: the attention epilogues evaluate one per output element on a
// VALU-bound path, where the library expm1f costs ~4 x as many instructions)
__device__ __forceinline__ float elu_fast(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }
// keep factors (1/(1-p) or 0; all 1 for key == 0) of FOUR consecutive elements e0 .. e0+3, e0 a multiple of 4: two hashes
// (a hash serves the two elements of an even / odd pair) instead of the four drop_factor() would compute
__device__ __forceinline__ void drop_factor4(uint32_t key, uint32_t e0, float (&f)[4]) {
  if (key == 0u) { f[0] = f[1] = f[2] = f[3] = 1.0f; return; }
  const uint32_t h0 = mix32(((e0 >> 1) * 0x9E3779B1u) ^ key);
  const uint32_t h1 = mix32((((e0 >> 1) + 1u) * 0x9E3779B1u) ^ key);
  f[0] = (h0 & 0xFFFFu) >= kDropThresh16 ? kKeepScale : 0.f;
  f[1] = (h0 >> 16) >= kDropThresh16 ? kKeepScale : 0.f;
  f[2] = (h1 & 0xFFFFu) >= kDropThresh16 ? kKeepScale : 0.f;
  f[3] = (h1 >> 16) >= kDropThresh16 ? kKeepScale : 0.f;
}
// d ELU(P)/dP from the output value O = ELU(P)
__device__ __forceinline__ float elu_grad_from_out(float o) { return o > 0.0f ? 1.0f : o + 1.0f; }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------
// bf16 "planes": hi = bf16_rne(x), lo = bf16_rne(x - hi).  x ~ hi + lo to 2^-16 relative, and
// a.b ~ hi.hi + hi.lo + lo.hi is what DVGR_BF16X3 feeds the bf16 MFMA.  Producers of GEMM operands
// store the planes next to (or instead of) the fp32 values so the contraction kernels convert nothing.
// ---------------------------------------------------------------------------------------------
// Streaming stores of kernel outputs (A/B: -DDVGR_NT_STORES).  An XCD's L2 is not coherent with the other seven, so whatever
// a kernel leaves dirty is written back at its end, in one burst the next kernel waits behind (MI355X_MICROARCH.md, row
// `boundary`: + bytes / 6 TB/s); outputs are consumed by workgroups on arbitrary XCDs, i.e. from memory anyway.
#ifdef DVGR_NT_STORES
typedef float dvgr_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t dvgr_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st4(float4* p, float4 v) {
  __builtin_nontemporal_store(dvgr_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<dvgr_f4v*>(p));
}
__device__ __forceinline__ void st2u(uint2* p, uint2 v) {
  __builtin_nontemporal_store(dvgr_u2v{v.x, v.y}, reinterpret_cast<dvgr_u2v*>(p));
}
#else
__device__ __forceinline__ void st4(float4* p, float4 v) { *p = v; }
__device__ __forceinline__ void st2u(uint2* p, uint2 v) { *p = v; }
#endif

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)a, (__bf16)b};                 // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo_as_f32(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_as_f32(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }
// 4 consecutive elements starting at element index idx (multiple of 4); hi == nullptr: no-op
__device__ __forceinline__ void split_store4(uint16_t* hi, int64_t lo_off, int64_t idx, float4 v) {
  if (!hi) return;
  const uint32_t h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
  st2u(reinterpret_cast<uint2*>(hi + idx), make_uint2(h01, h23));
  const uint32_t l01 = pack_bf16(v.x - bf16_lo_as_f32(h01), v.y - bf16_hi_as_f32(h01));
  const uint32_t l23 = pack_bf16(v.z - bf16_lo_as_f32(h23), v.w - bf16_hi_as_f32(h23));
  st2u(reinterpret_cast<uint2*>(hi + lo_off + idx), make_uint2(l01, l23));
}
__device__ __forceinline__ void split_store1(uint16_t* hi, int64_t lo_off, int64_t idx, float x) {
  if (!hi) return;
  const uint32_t h = pack_bf16(x, 0.f);
  hi[idx] = (uint16_t)h;
  hi[lo_off + idx] = (uint16_t)pack_bf16(x - bf16_lo_as_f32(h), 0.f);
}

}  // namespace dvgr
