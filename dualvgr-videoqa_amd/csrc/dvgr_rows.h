// dvgr_rows.h -- row bodies shared by a stand-alone row kernel and by a multi-phase launch (dvgr_gemm.hip: the S4 -> S5 phase
// experiment of round 3): one wave per row.
#pragma once
#include "dvgr_stages.h"

namespace dvgr {

// S5 forward of one (modality, row): model/Attention.py:20-23 + model/models.py:168-169 (and the next layer's gate / x-site copies)
__device__ __forceinline__ void sf_combine_row(const SfCombineArgs& a, int mod, int row, int lane) {
  const int D4 = a.D / 4;
  const float4* t0 = reinterpret_cast<const float4*>(a.tn[mod] + (int64_t)row * a.D);
  const float4* t1 = reinterpret_cast<const float4*>(a.tn[mod] + ((int64_t)a.rows + row) * a.D);
  const float4* p2 = reinterpret_cast<const float4*>(a.p2[mod]);
  float w0 = 0.f, w1 = 0.f;
  if (a.wpart[mod]) {                                        // the S4 epilogue's partial dots, one per 32-column block: fixed order
    const int nb = a.D / 32;
    const float* q0 = a.wpart[mod] + row;
    const float* q1 = q0 + (int64_t)nb * a.rows;
    for (int k = lane; k < nb; k += 64) { w0 += q0[(int64_t)k * a.rows]; w1 += q1[(int64_t)k * a.rows]; }
  } else {
    for (int k = lane; k < D4; k += 64) {
      const float4 u = t0[k], v = t1[k], p = p2[k];
      w0 += u.x * p.x + u.y * p.y + u.z * p.z + u.w * p.w;
      w1 += v.x * p.x + v.y * p.y + v.z * p.z + v.w * p.w;
    }
  }
  w0 = wave_sum(w0); w1 = wave_sum(w1);
  const float mx = fmaxf(w0, w1);
  const float e0 = expf(w0 - mx), e1 = expf(w1 - mx);
  const float b0 = e0 / (e0 + e1), b1 = e1 / (e0 + e1);
  if (lane == 0) { a.beta[mod][row * 2] = b0; a.beta[mod][row * 2 + 1] = b1; }
  const float4* z0 = reinterpret_cast<const float4*>(a.z0[mod] + (int64_t)row * a.D);
  const float4* z1 = reinterpret_cast<const float4*>(a.z1[mod] + (int64_t)row * a.D);
  const float4* x = reinterpret_cast<const float4*>(a.X[mod] + (int64_t)row * a.D);
  float4* xo = reinterpret_cast<float4*>(a.Xout[mod] + (int64_t)row * a.D);
  float4* oo = a.o[mod] ? reinterpret_cast<float4*>(a.o[mod] + (int64_t)row * a.D) : nullptr;
  const float4* np = a.next_p[mod] ? reinterpret_cast<const float4*>(a.next_p[mod] + (int64_t)(row / a.N) * a.D) : nullptr;
  float* nxd = a.next_xd[mod];                              // fp32 copies: only where a reader wants them (DVGR_F32)
  const bool want_xd = nxd != nullptr || a.next_xd_p[mod] != nullptr;
  const int64_t n4 = (int64_t)a.rows * D4;                  // float4s of one (B*N, D) tensor
  float gate = 0.f;
  for (int k = lane; k < D4; k += 64) {
    const float4 u = z0[k], v = z1[k], xx = x[k];
    const float4 o = make_float4(b0 * u.x + b1 * v.x, b0 * u.y + b1 * v.y, b0 * u.z + b1 * v.z, b0 * u.w + b1 * v.w);
    if (oo) st4(oo + k, o);
    const float4 xn = make_float4(xx.x + o.x, xx.y + o.y, xx.z + o.z, xx.w + o.w);
    st4(xo + k, xn);
    split_store4(a.xp[mod], a.x_lo_off, (int64_t)row * a.D + 4 * k, xn);
    if (np) {                                               // next layer: gate logit and the two dropped copies of X'
      const float4 pv = np[k];
      gate += xn.x * pv.x + xn.y * pv.y + xn.z * pv.z + xn.w * pv.w;
      if (want_xd) {
        const int64_t i = (int64_t)row * D4 + k;            // float4 index, as k_drop_x
        const uint32_t e = (uint32_t)(i * 4);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const uint32_t key = a.next_key[mod][g];
          const uint32_t h0 = mix32(((e >> 1) * 0x9E3779B1u) ^ key);
          const uint32_t h1 = mix32((((e >> 1) + 1u) * 0x9E3779B1u) ^ key);
          float4 y;
          y.x = xn.x * (((h0 & 0xFFFFu) >= kDropThresh16) ? kKeepScale : 0.f);
          y.y = xn.y * (((h0 >> 16) >= kDropThresh16) ? kKeepScale : 0.f);
          y.z = xn.z * (((h1 & 0xFFFFu) >= kDropThresh16) ? kKeepScale : 0.f);
          y.w = xn.w * (((h1 >> 16) >= kDropThresh16) ? kKeepScale : 0.f);
          if (nxd) st4(reinterpret_cast<float4*>(nxd) + (int64_t)g * n4 + i, y);
          split_store4(a.next_xd_p[mod], a.next_xd_lo_off, ((int64_t)g * n4 + i) * 4, y);
        }
      }
    }
  }
  if (np) {
    gate = wave_sum(gate);
    if (lane == 0) a.next_s[mod][row] = sigmoid_f(gate);
  }
}


}  // namespace dvgr
