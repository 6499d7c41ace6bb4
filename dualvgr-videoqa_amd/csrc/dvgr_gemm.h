// dvgr_gemm.h -- the one dense-contraction engine of the unit: C (+)= act(A x B + bias).
//
// Every contraction on the hot path is one of three shapes:
//   NT  Y = X . W^T      forward Linear           A = X [m][k] (KC)   B = W  [n][k] (KC)
//   NN  dX = dY . W      backward-data            A = dY[m][k] (KC)   B = W  [k][n] (IC)
//   TN  dW = dY^T . X    backward-weight          A = dY[k][m] (IC)   B = X  [k][n] (IC)
// KC = the reduction index k is contiguous in memory, IC = the tile index is contiguous.
//
// CDNA4 mapping: 256 threads = 4 waves (2 x 2), each wave owns a (BM/2) x (BN/2) block of
// 16x16 accumulator tiles and issues v_mfma_f32_16x16x4_f32 (exact fp32, = fmaf chain).
// Operand tiles are staged HBM -> registers -> LDS (double-buffered, one barrier per K-tile) so
// the dropout keep-mask of the x-site can be applied while the data passes through registers.
// LDS images: KC tile [rows][BK+4] read with ds_read_b128 (lane (r,g) takes k = 4g..4g+3 of row r
// and feeds MFMA j with element j: the k order inside a 16-block is permuted identically for A and
// B, which a sum does not care about); IC tile [BK][cols+4] read with 4 ds_read_b32.
#pragma once
#include "dvgr_common.h"

namespace dvgr {

enum : int { KC = 0, IC = 1 };
enum : int { LAYOUT_NT = 0, LAYOUT_NN = 1, LAYOUT_TN = 2 };   // GemmProblem::layout (gemm_launch_mixed)
enum : int { ACT_NONE = 0, ACT_TANH = 1, ACT_ELU = 2 };

struct GemmOperand {
  const float* p0;     // rows [0, kseg) of the reduction (IC) / the whole matrix (KC)
  const float* p1;     // rows [kseg, K) of the reduction (IC only; used when kseg < K)
  int64_t ld;          // leading dimension (floats); multiple of 4
  uint32_t drop_key;   // 0 = none; keep index = row * ld + col of the source matrix
  // Optional pre-split bf16 "planes" of the same matrix (same shape and ld): hi = bf16(x), lo = bf16(x - hi),
  // written by whoever produced the matrix (split_store in dvgr_common.h).  With planes on both operands the
  // bf16 modes run on gemm_planes_kernel, which moves bf16 straight into LDS and converts nothing.
  const uint16_t* h0;  // hi plane of segment 0 (nullptr: no planes)
  const uint16_t* h1;  // hi plane of segment 1
  int64_t lo_off;      // lo plane = hi plane + lo_off elements
};

struct GemmProblem {
  GemmOperand A, B;
  float* C;
  int64_t ldc;
  const float* bias;   // per output column, nullable
  int M, N, K, kseg;   // kseg == K when the reduction is one segment
  int act;             // ACT_*
  int accumulate;      // C += result instead of C = result
  uint32_t drop_key_c; // keep mask applied to the result (index row * ldc + col), 0 = none
  int dbg;             // profiling aid (tools/): bit 0 skip the epilogue, bit 1 skip the K loop; 0 in production
  int atomic_ok;       // the output is pre-zeroed (or accumulates) and has no bias/act/mask: the launcher may split K
                       // over several workgroups that add their partial tiles with float atomics
  int layout;          // LAYOUT_*: only read by gemm_launch_mixed (gemm_launch takes the layouts as arguments)
  int zero_on_split;   // atomic_ok with an output that is NOT pre-zeroed: a launch that does split K zeroes it first (own launch)
  int atomic_out;      // several problems of one batch ADD into this same (pre-zeroed) output: every tile is added with
                       // float atomics whether or not K is split
  float* ones_out;     // nullable: a tensor of C's shape and leading dimension that receives 1.0 wherever C is written, and
  float* sum_out;      // nullable: *sum_out += the sum of every value stored to C (one float atomic per workgroup) -- the forward's
                       // last contraction carries sum(vis) and d sum(vis) / d vis = 1 of the scope-U loss this way instead of a
                       // k_sum_and_ones launch behind it.  Honoured by the planes kernels only: gemm_sum_supported().
  // nullable, NT on the planes kernels (gemm_rowdot_supported): the partial row dots of the STORED values with a vector over the
  // output columns, one per 32-column block: rowdot_out[(col / 32) * rowdot_ld + row] = sum_{c in block} C[row][c] * rowdot_vec[c]
  // (every block of every row is written exactly once: no atomics, a reader sums the N / 32 partials in a fixed order).  S4 hands
  // S5 its logits w_v = tanh(z_v P1^T + b) . p2 this way instead of S5 reading the 2 x (B N D) tanh outputs back.
  const float* rowdot_vec;
  float* rowdot_out;
  int64_t rowdot_ld;
  float* rowsum;       // nullable, TN only (A = dY[k][m]): rowsum[m] += sum_k A[k][m], i.e. the bias gradient dY^T . 1 that
                       // belongs to the weight gradient C = dY^T . X, taken from the A fragments the kernel already holds
                       // (one extra MFMA against a ones fragment per A fragment, in the workgroups of output column 0).
                       // ADDED atomically: the target must be zeroed beforehand.  Only set it when gemm_rowsum_supported().
};

constexpr int kMaxBatch = 12;
// XCD block map of a planes launch (round 6; set by the launcher, never by callers): XCD x works on at most two rectangular
// blocks of output tiles, each inside ONE problem -- so its 4 MiB L2 streams the operand panels of one or two problems instead
// of a slice of every problem's (the default map cuts every problem's tile list into eight chunks of its own: each L2 then
// fetches ~1/3 of EVERY problem's panels).  Problems of one reduction length form a class; a class of P problems (8 % P == 0)
// gives each problem 8 / P XCDs, its tile grid cut into that many equal rectangles; two classes (the weight-gradient launch:
// dP1 reduces over 2 B N rows, dWcat over B N) give every XCD one block of each, so the XCDs carry equal work.
struct XcdSeg { unsigned char prob, m_lo, m_cnt, n_lo, n_cnt; };
struct GemmBatch {
  GemmProblem p[kMaxBatch];
  int n;
  int xmap;                 // 1: seg[][] is the workgroup -> tile map (grid = 8 x the largest per-XCD tile count, y = z = 1)
  XcdSeg seg[8][2];
};

// Launch-site tags: each tag is a separate instantiation (= its own kernel symbol in a profile).
enum : int { TAG_GENERIC = 0, TAG_FWD_GAT = 1, TAG_FWD_SF = 2, TAG_BWD_GAT_DW = 3, TAG_BWD_GAT_DX = 4,
             TAG_BWD_SF_DW = 5, TAG_BWD_SF_DX = 6 };

// True when gemm_launch will honour GemmProblem::rowsum for this batch (the LDS-DMA ring kernels do; the register-staged
// fallbacks for odd shapes and the batch-sized kernels do not: their callers use a column-sum job instead).
bool gemm_rowsum_supported(int la, int lb, const GemmBatch& batch, int prec);
// True when gemm_launch will honour GemmProblem::ones_out / sum_out for this batch (NT problems on the planes kernels).
bool gemm_sum_supported(int la, int lb, const GemmBatch& batch, int prec);
// True when gemm_launch will honour GemmProblem::rowdot_out for this batch (NT on the planes kernels, every tile whole).
bool gemm_rowdot_supported(int la, int lb, const GemmBatch& batch, int prec);

// Problems of different layouts (GemmProblem::layout) in ONE launch where every problem qualifies for the planes kernel
// (bf16 modes, pre-split operands, K % 32 == 0, not batch-sized); otherwise one gemm_launch per layout.  rowsum is honoured
// when gemm_mixed_supported().
bool gemm_mixed_supported(const GemmBatch& batch, int prec);
int gemm_launch_mixed(hipStream_t stream, const GemmBatch& batch, int tag = TAG_GENERIC, int prec = DVGR_F32);

// Which launch-site tags take the XCD block map (bit per tag); returns the previous mask, mask < 0 = the build's default.
int gemm_xmap_tags(int mask);
// The XCD block map's planner for n problems of (M, N, K) on bm x bn tiles: the grid's x extent, 0 when the batch does not fit the
// scheme, < 0 on bad arguments; seg_out (nullable): 8 x 2 x {problem, m_lo, m_cnt, n_lo, n_cnt}.
int gemm_plan_xcd_blocks(int n, const int* M, const int* N, const int* K, int bm, int bn, int max_per_xcd, int* seg_out);

// Enqueue the batch on `stream`.  la / lb in {KC, IC}.  Returns hipError_t as int.
int gemm_launch(hipStream_t stream, int la, int lb, const GemmBatch& batch, int tag = TAG_GENERIC, int prec = DVGR_F32);

// S6 of the forward as ONE launch (model/fusions/fusions.py:426-440): x0 = ELU(A . W0^T + b0), x1 = ELU(M . W1^T + b1) on the
// same tile coordinates of one workgroup and, in its epilogue, the pooled product z[r][k] = x0[r][2k] x1[r][2k] +
// x0[r][2k+1] x1[r][2k+1] (fp32 + bf16 planes).  bf16 modes, pre-split planes of all four operands, K % 32 == 0.
struct MfbFwdArgs {
  const uint16_t* xa; const uint16_t* xm; int64_t x_lo;     // hi planes of the two layer outputs (rows, K); lo = hi + x_lo
  const uint16_t* w0; const uint16_t* w1; int64_t w_lo;     // hi planes of linear0 / linear1 weights (hidden, K)
  const float* b0; const float* b1;                         // (hidden)
  float* x0; float* x1;                                     // (rows, hidden) fp32, saved for the backward
  float* z; uint16_t* zp; int64_t z_lo;                     // (rows, hidden / 2) fp32 + planes (zp nullable)
  int rows, K, hidden;
};
bool mfb_fwd_supported(int rows, int K, int hidden, int prec);
int mfb_fwd_launch(hipStream_t stream, const MfbFwdArgs& a, int prec);

inline GemmOperand operand(const float* p, int64_t ld, uint32_t drop_key = 0) {
  GemmOperand o; o.p0 = p; o.p1 = p; o.ld = ld; o.drop_key = drop_key; o.h0 = nullptr; o.h1 = nullptr; o.lo_off = 0; return o;
}

}  // namespace dvgr
