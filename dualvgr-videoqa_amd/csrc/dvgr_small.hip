// dvgr_small.hip -- batch-sized contractions.  A 128x128-tile kernel gives a problem with M = B (= 64) output rows six to
// eight workgroups that each walk the whole reduction; these kernels cut such problems into 64x16 (NT, NN) or 64x64 (TN)
// output tiles so that the weight matrix is streamed by 50-300 workgroups, and run them on the exact fp32 MFMA
// (v_mfma_f32_16x16x4_f32), so every precision mode gets fp32-grade results here.
//   NT  C[M][N] = act(A[M][K] . W[N][K]^T + bias)      M <= 128   (p = q Wq^T, question_proj, classifier Linears)
//   NN  C[M][N] (+)= A[M][K] . B[K][N]                  M <= 128   (dq-bar = dp_cat [Wq_a; Wq_m], dX of the classifier)
//   TN  C[M][N] = A[K][M]^T . B[K][N]                   K <= 256   (dWq = dp^T q, dW of the classifier: K = batch)
// Fragment convention of the 16x16x4 MFMA: lane l supplies A[row l%16][k l/16] and B[k l/16][col l%16]; acc[r] is
// C[row 4 (l/16) + r][col l%16].  Where the reduction index is contiguous a lane loads a float4 = its k-slot of four
// consecutive k-steps (any assignment of k values to steps works as long as A and B use the same one).
#include "dvgr_gemm.h"

namespace dvgr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float small_act(float v, int act) {
  if (act == ACT_TANH) return tanhf(v);
  if (act == ACT_ELU) return elu_f(v);
  return v;
}

// ---- NT: workgroup = 64 rows x 16 columns, the four waves split K ------------------------------------------------
__global__ __launch_bounds__(256) void k_small_nt(const GemmBatch batch) {
  __shared__ float red[4][4][256];
  const GemmProblem& P = batch.p[blockIdx.z];
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
  if (n0 >= P.N || m0 >= P.M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int K = P.K;
  const int kw = ((K + 63) / 64) * 16;                     // per-wave share of K, a multiple of 16
  const int kb = wave * kw, ke = (kb + kw) < K ? (kb + kw) : K;
  floatx4 acc[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) acc[I] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int wrow = n0 + r16;
  const float* wp = P.B.p0 + (int64_t)(wrow < P.N ? wrow : P.N - 1) * P.B.ld;
  const float* ap[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    const int row = m0 + 16 * I + r16;
    ap[I] = P.A.p0 + (int64_t)(row < P.M ? row : P.M - 1) * P.A.ld;
  }
  // software pipeline, two operand sets: the loads of step i + 1 are in flight while the MFMAs of step i issue
  struct Ops { float4 a[4]; float4 w; };
  auto load = [&](int k0, Ops& o) {
    const int k = k0 + 4 * kq;
    const bool in = k < ke;                                 // K % 4 == 0: a float4 is in or out as a whole
    o.w = (in && wrow < P.N) ? *reinterpret_cast<const float4*>(wp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int I = 0; I < 4; ++I)
      o.a[I] = (in && m0 + 16 * I + r16 < P.M) ? *reinterpret_cast<const float4*>(ap[I] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto mma = [&](const Ops& o) {
#pragma unroll
    for (int I = 0; I < 4; ++I) {
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].x, o.w.x, acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].y, o.w.y, acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].z, o.w.z, acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].w, o.w.w, acc[I], 0, 0, 0);
    }
  };
  Ops o0, o1;
  load(kb, o0);
  for (int k0 = kb; k0 < ke; k0 += 32) {
    load(k0 + 16, o1);
    mma(o0);
    load(k0 + 32, o0);
    mma(o1);
  }
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][I][r * 64 + lane] = acc[I][r];
  __syncthreads();
  const int r = tid >> 6, col = n0 + (lane & 15);
  if (col >= P.N) return;
  const float bias = P.bias ? P.bias[col] : 0.f;
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    const int row = m0 + 16 * I + 4 * (lane >> 4) + r;
    if (row < P.M) {
      const float v = (red[0][I][tid] + red[1][I][tid]) + (red[2][I][tid] + red[3][I][tid]) + bias;
      float* c = P.C + (int64_t)row * P.ldc + col;
      *c = small_act(P.accumulate ? *c + v : v, P.act);
    }
  }
}

// ---- NN: workgroup = 64 rows x 16 columns; B is [K][N] (two row segments allowed).  The four waves split the K range
// of the workgroup; gridDim.y = row blocks x K splits (splits > 1 only for outputs that may be added to atomically) ---
__global__ __launch_bounds__(256) void k_small_nn(const GemmBatch batch, int rowblocks) {
  __shared__ float red[4][4][256];
  const GemmProblem& P = batch.p[blockIdx.z];
  const int ksplit = (int)gridDim.y / rowblocks, ks = (int)blockIdx.y / rowblocks;
  const int n0 = blockIdx.x * 16, m0 = ((int)blockIdx.y % rowblocks) * 64;
  if (n0 >= P.N || m0 >= P.M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int K = P.K;
  const int kper = ((K + 64 * ksplit - 1) / (64 * ksplit)) * 64;   // per-workgroup share of K, multiple of 64
  const int kbeg = ks * kper, kend = (kbeg + kper) < K ? (kbeg + kper) : K;
  if (kbeg >= K) return;
  const int kw = kper / 4;
  const int kb = kbeg + wave * kw, ke = (kb + kw) < kend ? (kb + kw) : kend;
  floatx4 acc[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) acc[I] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int bcol = n0 + r16;
  const int bc = bcol < P.N ? bcol : P.N - 1;
  const float* ap[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    const int row = m0 + 16 * I + r16;
    ap[I] = P.A.p0 + (int64_t)(row < P.M ? row : P.M - 1) * P.A.ld;
  }
  struct Ops { float4 a[4]; float b[4]; };
  auto load = [&](int k0, Ops& o) {
    const int k = k0 + 4 * kq;
    const bool in = k < ke;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = k + e;
      o.b[e] = (in && bcol < P.N) ? ((kk < P.kseg) ? P.B.p0[(int64_t)kk * P.B.ld + bc] : P.B.p1[(int64_t)(kk - P.kseg) * P.B.ld + bc]) : 0.f;
    }
#pragma unroll
    for (int I = 0; I < 4; ++I)
      o.a[I] = (in && m0 + 16 * I + r16 < P.M) ? *reinterpret_cast<const float4*>(ap[I] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto mma = [&](const Ops& o) {
#pragma unroll
    for (int I = 0; I < 4; ++I) {
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].x, o.b[0], acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].y, o.b[1], acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].z, o.b[2], acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[I].w, o.b[3], acc[I], 0, 0, 0);
    }
  };
  Ops o0, o1;
  load(kb, o0);
  for (int k0 = kb; k0 < ke; k0 += 32) {
    load(k0 + 16, o1);
    mma(o0);
    load(k0 + 32, o0);
    mma(o1);
  }
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][I][r * 64 + lane] = acc[I][r];
  __syncthreads();
  const int r = tid >> 6, col = n0 + (lane & 15);
  if (col >= P.N) return;
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    const int row = m0 + 16 * I + 4 * (lane >> 4) + r;
    if (row < P.M) {
      const float v = (red[0][I][tid] + red[1][I][tid]) + (red[2][I][tid] + red[3][I][tid]);
      float* c = P.C + (int64_t)row * P.ldc + col;
      if (ksplit > 1 || P.atomic_out) atomicAdd(c, v);
      else *c = P.accumulate ? *c + v : v;
    }
  }
}

// ---- TN: workgroup = 64 x 64 outputs, wave w owns columns 16 w .. 16 w + 15; the whole (short) K per wave ------------
__global__ __launch_bounds__(256) void k_small_tn(const GemmBatch batch) {
  const GemmProblem& P = batch.p[blockIdx.z];
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  if (n0 >= P.N || m0 >= P.M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  floatx4 acc[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) acc[I] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int bcol = n0 + 16 * wave + r16;
  const int bc = bcol < P.N ? bcol : P.N - 1;
  int am[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) { const int m = m0 + 16 * I + r16; am[I] = m < P.M ? m : P.M - 1; }
  // eight k-steps (32 reduction rows) per operand set, two sets in flight
  struct Ops { float a[8][4]; float b[8]; };
  auto load = [&](int k0, Ops& o) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int k = k0 + 4 * s8 + kq;
      const bool in = k < P.K;
      o.b[s8] = (in && bcol < P.N) ? P.B.p0[(int64_t)k * P.B.ld + bc] : 0.f;
#pragma unroll
      for (int I = 0; I < 4; ++I) o.a[s8][I] = (in && m0 + 16 * I + r16 < P.M) ? P.A.p0[(int64_t)k * P.A.ld + am[I]] : 0.f;
    }
  };
  auto mma = [&](const Ops& o) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
      for (int I = 0; I < 4; ++I) acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[s8][I], o.b[s8], acc[I], 0, 0, 0);
  };
  Ops o0, o1;
  load(0, o0);
  for (int k0 = 0; k0 < P.K; k0 += 64) {
    load(k0 + 32, o1);
    mma(o0);
    load(k0 + 64, o0);
    mma(o1);
  }
  const int col = n0 + 16 * wave + r16;
  if (col >= P.N) return;
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * I + 4 * kq + r;
      if (row < P.M) {
        float* c = P.C + (int64_t)row * P.ldc + col;
        *c = P.accumulate ? *c + acc[I][r] : acc[I][r];
      }
    }
}

// Problems these kernels take: no dropout mask on an operand or on the result, one reduction segment except for NN's B.
bool small_eligible(const GemmBatch& b, int la, int lb) {
  for (int i = 0; i < b.n; ++i) {
    const GemmProblem& p = b.p[i];
    if (p.A.drop_key || p.B.drop_key || p.drop_key_c || p.dbg) return false;
    if (la == KC && lb == KC) { if (p.M > 128 || p.kseg != p.K || (p.K & 3)) return false; }
    else if (la == KC && lb == IC) { if (p.M > 128 || (p.K & 3) || p.bias || p.act != ACT_NONE) return false; }
    else if (la == IC && lb == IC) { if (p.K > 256 || p.kseg != p.K || p.bias || p.act != ACT_NONE) return false; }
    else return false;
  }
  return true;
}

int small_launch(hipStream_t st, int la, int lb, const GemmBatch& b) {
  int mx = 0, nx = 0;
  for (int i = 0; i < b.n; ++i) { mx = b.p[i].M > mx ? b.p[i].M : mx; nx = b.p[i].N > nx ? b.p[i].N : nx; }
  if (la == KC && lb == KC) DVGR_LAUNCH(k_small_nt, dim3((nx + 15) / 16, (mx + 63) / 64, b.n), dim3(256), 0, st, b);
  else if (la == KC && lb == IC) {
    // few output tiles and a long reduction: split K over workgroups when every output may be added to atomically
    const int rowblocks = (mx + 63) / 64, tiles = ((nx + 15) / 16) * rowblocks * b.n;
    bool atomic_ok = true;
    int kmin = 1 << 30;
    for (int i = 0; i < b.n; ++i) { atomic_ok = atomic_ok && b.p[i].atomic_ok && b.p[i].accumulate; kmin = b.p[i].K < kmin ? b.p[i].K : kmin; }
    int ksplit = 1;
    if (atomic_ok && tiles < 128) { ksplit = 256 / tiles; if (ksplit > kmin / 128) ksplit = kmin / 128; if (ksplit < 1) ksplit = 1; }
    DVGR_LAUNCH(k_small_nn, dim3((nx + 15) / 16, rowblocks * ksplit, b.n), dim3(256), 0, st, b, rowblocks);
  }
  else DVGR_LAUNCH(k_small_tn, dim3((nx + 63) / 64, (mx + 63) / 64, b.n), dim3(256), 0, st, b);
  return (int)hipGetLastError();
}

}  // namespace dvgr
