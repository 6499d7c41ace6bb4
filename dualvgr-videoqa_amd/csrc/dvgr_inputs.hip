// dvgr_inputs.hip -- include/dvgr_inputs.h: word features (embedding gather + dropout + tanh, and the scatter-add backward)
// and the motion projection (one Linear on the unit's contraction engine).
#include <hip/hip_runtime.h>

#include "../../include/dvgr.h"
#include "../../include/dvgr_inputs.h"
#include "dvgr_gemm.h"
#include "dvgr_stages.h"

namespace dvgr {
namespace {

constexpr uint32_t kWordsSite = 300000;      // dropout site of the word features, clear of the unit's, the head's, the encoders'

// One wave per token position: 16-byte gathers of the table row, the element's keep flag from the counter hash.
__global__ __launch_bounds__(256) void k_words_fwd(const float* table, int64_t vocab, int wd, const int64_t* tokens, int64_t n,
                                                   uint32_t key, float* words, int32_t* bad) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  int64_t tok = tokens[row];
  if (tok < 0 || tok >= vocab) {
    if (lane == 0 && bad) *bad = 1;
    tok = 0;
  }
  const float4* src = reinterpret_cast<const float4*>(table + tok * wd);
  float4* dst = reinterpret_cast<float4*>(words + row * wd);
  for (int c = lane; c < wd / 4; c += 64) {
    const float4 v = src[c];
    const uint32_t e = (uint32_t)(row * wd + 4 * c);
    float4 y;
    y.x = tanhf(v.x * drop_factor(key, e + 0));
    y.y = tanhf(v.y * drop_factor(key, e + 1));
    y.z = tanhf(v.z * drop_factor(key, e + 2));
    y.w = tanhf(v.w * drop_factor(key, e + 3));
    dst[c] = y;
  }
}

__global__ __launch_bounds__(256) void k_words_bwd(const float* g_words, const float* words, const int64_t* tokens, int64_t n,
                                                   int64_t vocab, int wd, uint32_t key, float* g_table) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  int64_t tok = tokens[row];
  if (tok < 0 || tok >= vocab) tok = 0;
  const float4* g4 = reinterpret_cast<const float4*>(g_words + row * wd);
  const float4* y4 = reinterpret_cast<const float4*>(words + row * wd);
  float* dst = g_table + tok * wd;
  for (int c = lane; c < wd / 4; c += 64) {
    const float4 g = g4[c], y = y4[c];
    const uint32_t e = (uint32_t)(row * wd + 4 * c);
    atomicAdd(dst + 4 * c + 0, g.x * (1.f - y.x * y.x) * drop_factor(key, e + 0));
    atomicAdd(dst + 4 * c + 1, g.y * (1.f - y.y * y.y) * drop_factor(key, e + 1));
    atomicAdd(dst + 4 * c + 2, g.z * (1.f - y.z * y.z) * drop_factor(key, e + 2));
    atomicAdd(dst + 4 * c + 3, g.w * (1.f - y.w * y.w) * drop_factor(key, e + 3));
  }
}

__global__ __launch_bounds__(256) void k_words_keep(uint32_t key, int64_t count, float* keep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) keep[i] = drop_keep(key, (uint32_t)i) ? 1.f : 0.f;
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// y[r][:] = b: lets a few-tile, long-K projection run split over K (partial tiles ADD into y), see dvgr_linear_forward
__global__ __launch_bounds__(256) void k_fill_bias(float* y, const float* b, int64_t rows, int N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // float4 index
  const int n4 = N / 4;
  if (i >= rows * n4) return;
  reinterpret_cast<float4*>(y)[i] = reinterpret_cast<const float4*>(b)[i % n4];
}

}  // namespace
}  // namespace dvgr

using namespace dvgr;

extern "C" {

int dvgr_words_forward(const float* table, int64_t vocab, int wd, const int64_t* tokens, int64_t n, int train, uint64_t seed,
                       float* words, int32_t* bad_tokens, void* stream) {
  if (!table || !tokens || !words) return DVGR_E_NULL;
  if (vocab < 1 || wd < 4 || (wd & 3) || n < 0 || n * (int64_t)wd >= (1ll << 32)) return DVGR_E_DIMS;   // 32-bit dropout counters
  if (!aligned16(table) || !aligned16(words)) return DVGR_E_ALIGN;
  if (n == 0) return 0;
  const uint32_t key = train ? site_key(seed, kWordsSite) : 0u;
  DVGR_LAUNCH(k_words_fwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, table, vocab, wd, tokens, n, key,
                     words, bad_tokens);
  return (int)hipGetLastError();
}

int dvgr_words_backward(const float* g_words, const float* words, const int64_t* tokens, int64_t n, int64_t vocab, int wd, int train,
                        uint64_t seed, float* g_table, void* stream) {
  if (!g_words || !words || !tokens || !g_table) return DVGR_E_NULL;
  if (vocab < 1 || wd < 4 || (wd & 3) || n < 0 || n * (int64_t)wd >= (1ll << 32)) return DVGR_E_DIMS;
  if (!aligned16(g_words) || !aligned16(words)) return DVGR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(g_table, 0, (size_t)vocab * wd * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  const uint32_t key = train ? site_key(seed, kWordsSite) : 0u;
  DVGR_LAUNCH(k_words_bwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, g_words, words, tokens, n, vocab, wd, key, g_table);
  return (int)hipGetLastError();
}

int dvgr_words_dbg_keep_mask(uint64_t seed, int64_t count, float* keep, void* stream) {
  if (!keep) return DVGR_E_NULL;
  if (count < 0 || count >= (1ll << 32)) return DVGR_E_DIMS;
  if (count == 0) return 0;
  DVGR_LAUNCH(k_words_keep, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, site_key(seed, kWordsSite), count,
                     keep);
  return (int)hipGetLastError();
}

int dvgr_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int dtype, void* stream) {
  if (!x || !w || !y) return DVGR_E_NULL;
  if (M < 1 || N < 4 || K < 4 || (N & 3) || (K & 3)) return DVGR_E_DIMS;
  if (dtype != DVGR_F32 && dtype != DVGR_BF16 && dtype != DVGR_BF16X3) return DVGR_E_DIMS;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y)) return DVGR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  GemmBatch gb{};
  GemmProblem p{};
  p.A = operand(x, K); p.B = operand(w, K);
  p.C = y; p.ldc = N; p.bias = b; p.M = M; p.N = N; p.K = K; p.kseg = K; p.act = ACT_NONE; p.layout = LAYOUT_NT;
  // The motion projection at C2 is 60 output tiles with a 2048-long reduction: a quarter of the chip for 64 K-tiles.  With
  // the bias written first and the contraction ACCUMULATING into y the launcher may split K over workgroups (float atomics).
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles <= 128 && K >= 1024 && (K % 32) == 0) {
    if (b) {
      if (!aligned16(b)) return DVGR_E_ALIGN;
      const int64_t n4 = (int64_t)M * (N / 4);
      DVGR_LAUNCH(k_fill_bias, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, y, b, (int64_t)M, N);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
      p.bias = nullptr; p.accumulate = 1; p.atomic_ok = 1;
    } else {
      p.atomic_ok = 1; p.zero_on_split = 1;
    }
  }
  gb.p[gb.n++] = p;
  return gemm_launch(st, KC, KC, gb, TAG_GENERIC, dtype);
}

int dvgr_linear_backward(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int M, int N, int K, int dtype,
                         void* stream) {
  if (!x || !w || !gy || !gw) return DVGR_E_NULL;
  if (M < 4 || N < 4 || K < 4 || (N & 3) || (K & 3)) return DVGR_E_DIMS;
  if (dtype != DVGR_F32 && dtype != DVGR_BF16 && dtype != DVGR_BF16X3) return DVGR_E_DIMS;
  if (!aligned16(x) || !aligned16(w) || !aligned16(gy) || !aligned16(gw) || (gx && !aligned16(gx))) return DVGR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  {                                            // gw = gy^T . x: reduction over the M rows
    GemmBatch b{};
    GemmProblem p{};
    p.A = operand(gy, N); p.B = operand(x, K);
    p.C = gw; p.ldc = K; p.M = N; p.N = K; p.K = M; p.kseg = M; p.layout = LAYOUT_TN;
    p.atomic_ok = 1; p.zero_on_split = 1;      // a long reduction may be split over workgroups; the launch then zeroes gw first
    b.p[b.n++] = p;
    const int e = gemm_launch(st, IC, IC, b, TAG_GENERIC, dtype);
    if (e) return e;
  }
  if (gb) {
    const hipError_t e0 = hipMemsetAsync(gb, 0, (size_t)N * sizeof(float), st);
    if (e0 != hipSuccess) return (int)e0;
    ColsumBatch cb{};
    cb.j[cb.n++] = ColsumJob{gy, nullptr, nullptr, nullptr, M, 0, N, N, gb};
    cb.chunks = 16;
    const int e = colsum(st, cb);
    if (e) return e;
  }
  if (gx) {
    GemmBatch b{};
    GemmProblem p{};
    p.A = operand(gy, N); p.B = operand(w, K);
    p.C = gx; p.ldc = K; p.M = M; p.N = K; p.K = N; p.kseg = N; p.layout = LAYOUT_NN;
    b.p[b.n++] = p;
    const int e = gemm_launch(st, KC, IC, b, TAG_GENERIC, dtype);
    if (e) return e;
  }
  return 0;
}

}  // extern "C"
