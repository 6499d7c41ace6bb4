// dvgr_optim.hip -- gradient-norm clipping + Adam over a list of flat segments (include/dvgr_optim.h).
// HBM-bound element-wise work: 16 B per lane, grid-stride inside a segment, blockIdx.y = segment of this launch's table.
#include <math.h>

#include "../../include/dvgr_optim.h"
#include "dvgr_common.h"

namespace dvgr {

constexpr int kSegsPerLaunch = 40;         // segment table passed by value (kernarg): 40 * 56 B
struct SegTable {
  dvgr_optim_seg s[kSegsPerLaunch];
};

// One atomicAdd per workgroup into ONE address: the grid is capped at kSqnormBlocks workgroups per segment (strided loop, four
// 16-byte loads in flight per thread) -- with a workgroup per 1024 elements (30 k same-address atomics for the whole model's
// 30 M gradients) the launch took 229 us, i.e. 0.5 TB/s: the atomics, not the 120 MB, were the time.
constexpr int kSqnormBlocks = 512;
__global__ __launch_bounds__(256) void k_optim_sqnorm(const SegTable tab, float* __restrict__ out) {
  const dvgr_optim_seg S = tab.s[blockIdx.y];
  const int64_t n = S.n;
  const int64_t start = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4, stride = (int64_t)gridDim.x * 256 * 4;
  if ((int64_t)blockIdx.x * 1024 >= n) return;          // uniform per workgroup
  float acc = 0.f;
  const bool vec = ((((uintptr_t)S.grad) & 15) == 0);
  int64_t i = start;
  if (vec) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; i + 3 * stride + 4 <= n; i += 4 * stride) {
      const float4 g0 = *reinterpret_cast<const float4*>(S.grad + i), g1 = *reinterpret_cast<const float4*>(S.grad + i + stride);
      const float4 g2 = *reinterpret_cast<const float4*>(S.grad + i + 2 * stride), g3 = *reinterpret_cast<const float4*>(S.grad + i + 3 * stride);
      a0 += g0.x * g0.x + g0.y * g0.y + g0.z * g0.z + g0.w * g0.w;
      a1 += g1.x * g1.x + g1.y * g1.y + g1.z * g1.z + g1.w * g1.w;
      a2 += g2.x * g2.x + g2.y * g2.y + g2.z * g2.z + g2.w * g2.w;
      a3 += g3.x * g3.x + g3.y * g3.y + g3.z * g3.z + g3.w * g3.w;
    }
    acc = (a0 + a1) + (a2 + a3);
  }
  for (; i < n; i += stride) {
    if (vec && i + 4 <= n) {
      const float4 g = *reinterpret_cast<const float4*>(S.grad + i);
      acc += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
    } else {
      for (int64_t e = i; e < n && e < i + 4; ++e) acc += S.grad[e] * S.grad[e];
    }
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

struct AdamArgs {
  const float* sqnorm;
  float max_norm, lr_t, b1, b2, bc2_sqrt, eps;
};

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, float coef, const AdamArgs& a) {
  g *= coef;
  m = m + (g - m) * (1.0f - a.b1);                       // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.b2 + (1.0f - a.b2) * g * g;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps; // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  p -= a.lr_t * (m / denom);                             // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(256) void k_optim_adam(const SegTable tab, const AdamArgs a) {
  const dvgr_optim_seg S = tab.s[blockIdx.y];
  const int64_t n = S.n;
  if ((int64_t)blockIdx.x * 1024 >= n) return;
  float coef = 1.0f;
  if (a.sqnorm && a.max_norm > 0.f) {
    const float c = a.max_norm / (sqrtf(*a.sqnorm) + 1e-6f);
    coef = c < 1.0f ? c : 1.0f;                          // torch.clamp(clip_coef, max=1.0)
  }
  const int64_t start = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4, stride = (int64_t)gridDim.x * 256 * 4;
  const bool vec = (((uintptr_t)S.param | (uintptr_t)S.grad | (uintptr_t)S.exp_avg | (uintptr_t)S.exp_avg_sq) & 15) == 0;
  for (int64_t i = start; i < n; i += stride) {
    if (vec && i + 4 <= n) {
      float4 p = *reinterpret_cast<float4*>(S.param + i), g = *reinterpret_cast<float4*>(S.grad + i);
      float4 m = *reinterpret_cast<float4*>(S.exp_avg + i), v = *reinterpret_cast<float4*>(S.exp_avg_sq + i);
      adam1(p.x, g.x, m.x, v.x, coef, a);
      adam1(p.y, g.y, m.y, v.y, coef, a);
      adam1(p.z, g.z, m.z, v.z, coef, a);
      adam1(p.w, g.w, m.w, v.w, coef, a);
      *reinterpret_cast<float4*>(S.param + i) = p;
      *reinterpret_cast<float4*>(S.grad + i) = g;
      *reinterpret_cast<float4*>(S.exp_avg + i) = m;
      *reinterpret_cast<float4*>(S.exp_avg_sq + i) = v;
      split_store4(reinterpret_cast<uint16_t*>(S.planes), S.lo_off, i, p);
    } else {
      for (int64_t e = i; e < n && e < i + 4; ++e) {
        adam1(S.param[e], S.grad[e], S.exp_avg[e], S.exp_avg_sq[e], coef, a);
        split_store1(reinterpret_cast<uint16_t*>(S.planes), S.lo_off, e, S.param[e]);
      }
    }
  }
}

static int check_segs(const dvgr_optim_seg* segs, int nseg, bool need_state) {
  if (!segs) return DVGR_E_NULL;
  if (nseg < 0 || nseg > DVGR_OPTIM_MAX_SEGS) return DVGR_E_DIMS;
  for (int i = 0; i < nseg; ++i) {
    if (segs[i].n < 0) return DVGR_E_DIMS;
    if (segs[i].n == 0) continue;
    if (!segs[i].grad) return DVGR_E_NULL;
    if (need_state && (!segs[i].param || !segs[i].exp_avg || !segs[i].exp_avg_sq)) return DVGR_E_NULL;
    if (((uintptr_t)segs[i].grad) & 3) return DVGR_E_ALIGN;
    if (need_state && segs[i].planes && ((((uintptr_t)segs[i].planes) & 7) || (segs[i].lo_off & 3))) return DVGR_E_ALIGN;
  }
  return 0;
}

// grid.x of one launch: enough 1024-element blocks for its largest segment, capped (grid-stride covers the rest)
static dim3 table_grid(const SegTable& t, int cnt, int64_t cap = 2048) {
  int64_t nmax = 1;
  for (int i = 0; i < cnt; ++i) nmax = t.s[i].n > nmax ? t.s[i].n : nmax;
  int64_t bx = (nmax + 1023) / 1024;
  if (bx > cap) bx = cap;
  return dim3((unsigned)bx, (unsigned)cnt, 1);
}

}  // namespace dvgr

using namespace dvgr;

extern "C" {

int dvgr_optim_grad_sqnorm(const dvgr_optim_seg* segs, int nseg, float* sqnorm, void* stream) {
  if (!sqnorm) return DVGR_E_NULL;
  const int e0 = check_segs(segs, nseg, false);
  if (e0) return e0;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(sqnorm, 0, sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  for (int base = 0; base < nseg; base += kSegsPerLaunch) {
    SegTable t{};
    const int cnt = (nseg - base) < kSegsPerLaunch ? (nseg - base) : kSegsPerLaunch;
    for (int i = 0; i < cnt; ++i) t.s[i] = segs[base + i];
    DVGR_LAUNCH(k_optim_sqnorm, table_grid(t, cnt, kSqnormBlocks), dim3(256), 0, st, t, sqnorm);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return (int)le;
  }
  return 0;
}

int dvgr_optim_adam_step(const dvgr_optim_seg* segs, int nseg, const float* sqnorm, float max_norm, float lr, float beta1,
                         float beta2, float eps, int64_t step, void* stream) {
  const int e0 = check_segs(segs, nseg, true);
  if (e0) return e0;
  if (step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return DVGR_E_DIMS;
  hipStream_t st = (hipStream_t)stream;
  AdamArgs a{};
  a.sqnorm = sqnorm; a.max_norm = max_norm; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.lr_t = (float)((double)lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  for (int base = 0; base < nseg; base += kSegsPerLaunch) {
    SegTable t{};
    const int cnt = (nseg - base) < kSegsPerLaunch ? (nseg - base) : kSegsPerLaunch;
    for (int i = 0; i < cnt; ++i) t.s[i] = segs[base + i];
    DVGR_LAUNCH(k_optim_adam, table_grid(t, cnt), dim3(256), 0, st, t, a);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return (int)le;
  }
  return 0;
}

}  // extern "C"
