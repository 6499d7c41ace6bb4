// dvgr_stages.h -- launchers of the non-GEMM kernels of the unit (row-wise, attention and
// element-wise stages).  Every function enqueues on `st` and returns a hipError_t as int.
#pragma once
#include "dvgr_common.h"

namespace dvgr {

// ---- forward ------------------------------------------------------------------------------------
// S0 (rows): per (b, layer): y -> d = y / max(||y||, 1e-12); r = d.w2 + b2; softmax over all T;
// mask t < len; renormalise with +1e-5; q = sum_t ah_t words_t.   model/utils.py:68-82
struct QaRowsArgs {
  const float* Y;        // [L][B*T][D]  feat_enhance output (workspace)
  float* logit;          // [L][B*T]     scratch: fc logits between the two passes (workspace)
  const float* params;   // flat parameter buffer
  int64_t off_w2, off_b2, layer_stride;
  const float* words;    // (B, T, Wd)
  const int64_t* qlen;   // (B)
  float* saved;          // base of the saved buffer
  int64_t s_layer_stride, s_d, s_nrm, s_a, s_ah, s_S, s_q;
  int B, T, D, Wd, L;
};
int qa_rows_fwd(hipStream_t st, const QaRowsArgs& a);

// S1: s[mod][b][n] = sigmoid(X[mod][b][n][:] . p[mod][b][:])                model/utils.py:101-103
int punish_fwd(hipStream_t st, const float* X0, const float* X1, const float* p0, const float* p1,
               float* s0, float* s1, int B, int N, int D);

// x-site dropout of the two GATs of both modalities, materialised once per layer (training mode):
// out[m][g][i] = X[m][i] * keep(key[m][g], i) / (1 - p).  model/GraphNN.py:175
int drop_x(hipStream_t st, const float* X0, const float* X1, float* out0, float* out1, const uint32_t (&key)[2][2], int64_t n,
           uint16_t* p0 = nullptr, uint16_t* p1 = nullptr, int64_t lo_off = 0);

// hi/lo bf16 planes of up to 4 fp32 tensors (the unit's inputs)
struct SplitJob {
  const float* src; uint16_t* hi; int64_t n;
  // optional (training mode, layer 0's x-site): the planes of the two differently dropped copies of `src` as well,
  // element index = index into src (the convention of k_drop_x / GemmOperand::drop_key); drop_hi[g] == nullptr: none
  uint16_t* drop_hi[2];
  uint32_t drop_key[2];
};
struct SplitBatch { SplitJob j[4]; int n; int64_t lo_off; float* zero3; /* nullable: three floats cleared on the way */ };
int split_planes(hipStream_t st, const SplitBatch& b);

// S3: per (b, head, mod): logits, softmax over j, punish-scaled aggregation, ELU, dropout.
struct GatAttnArgs {
  const float* Wh[2];    // (B*N, 8F) per modality
  const float* s[2];     // (B*N)
  const float* gat_a[2]; // (8, 2F)   [a_l | a_r] per head
  const float* gat_c[2]; // (8)
  float* u[2];           // (B, 8, N) saved
  float* v[2];
  float* out_c[2];       // common-GAT output   (B, N, D)
  float* out_s[2];       // specific-GAT output (B, N, D)
  uint32_t key_alpha[2][8];  // dropout keys (0 = eval)
  uint32_t key_out[2][2];    // [mod][gat]
  uint16_t* zp_c[2];     // bf16 hi plane of out_c (nullable); lo plane at + z_lo_off
  uint16_t* zp_s[2];
  int64_t z_lo_off;
  int B, N, D;
};
int gat_attn_fwd(hipStream_t st, const GatAttnArgs& a);

// S2 + S3 in one launch (dvgr_gatf.hip): the projection Wh = xd . Wcat^T + bcat of one head for a tile of whole samples, with
// the attention of GatAttnArgs as its epilogue.  bf16 modes with pre-split planes of both operands only.
struct GatFusedArgs {
  GatAttnArgs at;              // at.Wh is not read (the projection is produced here); everything else as for gat_attn_fwd
  const uint16_t* xh[2][2];    // [mod][gat] hi plane of the GAT's input (B*N, D): the dropped copy (training) or X itself
  int64_t x_lo_off;            // lo plane = hi plane + x_lo_off elements
  const uint16_t* wh[2];       // [mod] hi plane of Wcat (8F, D)
  int64_t w_lo_off;
  const float* bias[2];        // bcat (8F)
  // optional (both or neither; bf16x3 only): the 8-bit lo plane of Wcat (8F, D) bytes and its per-row scales (8F), dvgr_pack_lo8:
  // the launch then streams it instead of the bf16 lo plane (dvgr_gatf.h, LO8)
  const uint8_t* wq[2];
  const float* wscale[2];
  float* Wh_out[2];            // (B*N, 8F) fp32, saved for the backward
  int kb64;                    // != 0: 64-deep K-tiles (bf16x3, bf16 lo planes; dvgr_gatf.h KB)
  int no_half;                 // != 0: never the half-height (32-row) tiles (DVGR_OPT_NO_HALF_GAT_TILES)
  int spt;                     // samples per tile (set by the launcher)
};
bool gat_fused_supported(int B, int N, int D, int prec);
int gat_fused_fwd(hipStream_t st, const GatFusedArgs& a, int prec);

// S5: w_v = tn_v . p2; beta = softmax over the two views; o = sum_v beta_v z_v; X' = X + o.
struct SfCombineArgs {
  const float* X[2];     // layer input (B*N, D)
  const float* z0[2];    // common output
  const float* z1[2];    // specific output
  const float* tn[2];    // (2, B*N, D)
  const float* p2[2];    // (D)
  const float* wpart[2]; // nullable: (2, D / 32, B*N) partial logits tn_v[row] . p2 per 32-column block, written by the S4
                         // contraction's epilogue (GemmProblem::rowdot_out): tn is then not read here
  float* beta[2];        // (B*N, 2) saved
  float* o[2];           // fused embedding (B*N, D); nullable
  float* Xout[2];        // X' (B*N, D)
  uint16_t* xp[2];       // bf16 planes of X' (nullable)
  int64_t x_lo_off;
  int rows, D;
  // Optional: the next layer's S1 and x-site dropout on the row while it is still in registers (saves two launches and
  // two re-reads of X' per layer).  next_p == nullptr: off.
  const float* next_p[2];   // (B, D) punish vectors of the next layer
  float* next_s[2];         // (B*N)  gates of the next layer
  float* next_xd[2];        // (2, B*N, D) dropped copies for the next layer's two GATs (training), nullable
  uint16_t* next_xd_p[2];   // their bf16 planes (nullable)
  int64_t next_xd_lo_off;
  uint32_t next_key[2][2];  // [mod][gat] x-site dropout keys of the next layer
  int N;                    // clips per sample (row / N = sample), needed with next_p
};
int sf_combine_fwd(hipStream_t st, const SfCombineArgs& a);

// S6 middle: z[r][k] = x0[r][2k] x1[r][2k] + x0[r][2k+1] x1[r][2k+1]   model/fusions/fusions.py:430-440
int mfb_pool_fwd(hipStream_t st, const float* x0, const float* x1, float* z, int rows, uint16_t* zp = nullptr, int64_t lo_off = 0);

// ---- backward -----------------------------------------------------------------------------------
// out[i] = g[i] * ELU'(from output y[i]); g == nullptr -> zeros
int elu_bwd(hipStream_t st, const float* g, const float* y, float* out, int64_t n, uint16_t* op = nullptr, int64_t lo_off = 0);
// dp0 = dz[k/2] * x1 * ELU'(x0), dp1 = dz[k/2] * x0 * ELU'(x1)
int mfb_pool_bwd(hipStream_t st, const float* dz, const float* x0, const float* x1, float* dp0, float* dp1, int rows,
                 uint16_t* p0 = nullptr, uint16_t* p1 = nullptr, int64_t lo_off = 0);

struct SfBwdArgs {
  const float* dX[2];      // grad wrt X' (B*N, D)
  const float* g_embed[2]; // direct upstream grad of the fused embedding (last layer) or nullptr
  const float* z0[2];
  const float* z1[2];
  const float* tn[2];
  const float* beta[2];
  const float* p2[2];
  const float* g_z0[2];    // direct upstream grads of the list outputs, nullable
  const float* g_z1[2];
  float* dw[2];            // (2, B*N)
  float* dpre[2];          // (2, B*N, D)
  float* dz[2];            // (2, B*N, D)  = beta_v * do + g_z_v   (GEMM adds dpre . P1); nullptr: not written (the fused
                           // GAT-view backward forms it itself, GatFusedBwdArgs::dx), g_z0 / g_z1 are then not read
  uint16_t* dpre_p[2];     // bf16 planes of dpre (nullable)
  int64_t lo_off;
  int rows, D;
  // Round 6: the dX half of the layer ABOVE's S1 backward (k_punish_bwd, parts bit 1) folded into this pass, which reads the
  // running gradient anyway: with fin_dX[mod] != nullptr (= dX[mod]) the row's gradient is FIRST completed in place,
  //   dX[row] <- dX[row] + fin_add[row] (nullable) + dtau_row * fin_p[row / fin_N],   dtau = (sum_h fin_ds_part) s (1 - s),
  // (the tensors of the layer above: its attention backward's ds partials (B, 8, N), its gates s (B N), its punish vector p
  // (B, D), the second partial of its dX contraction) and everything below uses the completed row.  One launch fewer on the
  // critical path per layer >= 1, and the row is read once instead of twice.
  float* fin_dX[2];
  const float* fin_add[2];
  const float* fin_ds_part[2];
  const float* fin_s[2];
  const float* fin_p[2];
  int fin_N;
};
int sf_bwd(hipStream_t st, const SfBwdArgs& a);

struct GatAttnBwdArgs {
  const float* dz[2];      // (2, B*N, D): view 0 = grad of common output, view 1 = specific
  const float* out_c[2];   // saved outputs
  const float* out_s[2];
  const float* Wh[2];
  const float* s[2];
  const float* u[2];
  const float* v[2];
  const float* gat_a[2];
  const float* gat_c[2];
  float* dWh[2];           // (B*N, 8F)
  float* ds_part[2];       // (B, 8, N)
  float* da_part[2];       // (B, 8, 2F)
  float* dc_part[2];       // (B, 8)
  uint32_t key_alpha[2][8];
  uint32_t key_out[2][2];
  uint16_t* dWh_p[2];      // bf16 planes of dWh (nullable)
  int64_t lo_off;
  int B, N, D;
};
int gat_attn_bwd(hipStream_t st, const GatAttnBwdArgs& a);

// S4 backward (data) + S3 backward in one launch (dvgr_gatb.hip): dz_g = [what k_sf_bwd left in at.dz] + dpre_g . P1 for a tile
// of whole samples x one head's columns, with the attention backward of GatAttnBwdArgs as its epilogue.  bf16 modes with
// pre-split planes of both operands, graph_layers = 1.  at.dz is READ (the part k_sf_bwd wrote); dz itself is not written back.
struct GatFusedBwdArgs {
  GatAttnBwdArgs at;
  const uint16_t* ah[2];       // [mod] hi plane of dpre (2, B*N, D)
  int64_t a_lo_off;            // lo plane = hi plane + a_lo_off elements
  const uint16_t* bh[2];       // [mod] hi plane of P1^T (D, D): row = input feature i of project[0], reduction index o contiguous
  int64_t b_lo_off;
  const uint8_t* bq[2];        // optional (both or neither; bf16x3 only): the 8-bit lo plane of P1^T (D, D) bytes and its per-row
  const float* bscale[2];      //   scales (D), dvgr_pack_lo8
  // dx[mod] != nullptr: at.dz is NOT read; the part of dz that k_sf_bwd would have written is formed here instead,
  //   dz_g = beta_g (dx + g_embed) + g_z[g]      (model/Attention.py:22-23 backward: o = sum_v beta_v z_v),
  // from the running gradient of the layer output (B*N, D), the direct upstream gradient of the fused embedding (last layer;
  // nullable), the list gradients of the view (nullable) and the saved softmax weights beta (B*N, 2): k_sf_bwd then skips
  // its 2 x (B N D) fp32 store and the re-read of the two list gradients (SfBwdArgs::dz == nullptr).
  const float* dx[2];
  const float* g_embed[2];
  const float* g_z[2][2];      // [mod][view]
  const float* beta[2];
  int kb64;                    // != 0: 64-deep K-tiles (bf16x3, bf16 lo planes, with dx; dvgr_gatf.h KB)
  int no_half;                 // != 0: never the half-height (32-row) tiles (DVGR_OPT_NO_HALF_GAT_TILES)
  int spt;                     // samples per tile (set by the launcher)
};
bool gat_fused_bwd_supported(int B, int N, int D, int prec);
bool gat_fused_bwd_preferred(int B, int N, bool half_tiles_ok = false);     // the fused form is also the faster one at this batch (measured rule); half_tiles_ok: the launch may take 32-row tiles (dvgr_gatf.h)
int gat_fused_bwd(hipStream_t st, const GatFusedBwdArgs& a, int prec);

// bf16 planes of square matrices transposed (hi and lo plane of each job in one workgroup pass): dst[c][r] = src[r][c]
constexpr int kMaxTransposeJobs = 24;
struct TransposeJob { const uint16_t* src; uint16_t* dst; };
struct TransposeBatch {
  TransposeJob j[kMaxTransposeJobs];
  int n;
  int64_t src_lo, dst_lo;      // lo plane = hi plane + offset (elements)
  int D;                       // multiple of 64
};
int transpose_planes(hipStream_t st, const TransposeBatch& b);

// 8-bit block-scaled lo planes (dvgr_gatf.h, LO8): per row of `rows` x D bf16 lo values, s = max |lo| / 127 and
// q = round(lo / s) + 128 (uint8, 128 = zero); lo ~ (q - 128) s.  One workgroup per row.
constexpr int kMaxLo8Jobs = 48;
struct Lo8Job { const uint16_t* lo; uint8_t* q; float* scale; int rows; };
struct Lo8Batch { Lo8Job j[kMaxLo8Jobs]; int n; int D; };
int quantize_lo8(hipStream_t st, const Lo8Batch& b);

// S1 backward: ds = sum_h ds_part; dtau = ds s (1-s); dX += dtau p (+ add); dp = sum_n dtau X.
struct PunishBwdArgs {
  const float* ds_part[2];
  const float* s[2];
  const float* p[2];
  const float* X[2];
  float* dX[2];            // accumulated in place
  const float* add[2];     // nullable: a second partial of dX (B*N, D), folded in by the same read-modify-write
  float* dp_cat;           // (B, 2D): cols [mod*D, (mod+1)*D)
  int B, N, D;
  int parts;               // bit 0: dp (needs only ds_part, s, X), bit 1: the dX update; 0 means both
  float* zero;             // nullable: a (B, zero_cols) buffer the dp part clears on its way (saves a memset launch)
  int zero_cols;
  int copies;              // graph_layers > 1: ds_part holds `copies` consecutive (B, 8, N) blocks (one per graph layer), all
                           // summed; 0 or 1: one block
};
int punish_bwd(hipStream_t st, const PunishBwdArgs& a);

// graph_layers > 1, graph layer j < G - 1 of a unit layer: the gradient tile the attention backward consumes,
//   dz[view 0] = g_c (or 0),   dz[view 1] = g_s (or 0) + in0 + in1      (all (rows, D); in0 / in1: the two partial gradients
// of the next graph layer's input -- its specific GAT output -- through that layer's common / specific GAT; nullable)
struct DzBuildArgs {
  const float* g_c[2];     // per modality, nullable
  const float* g_s[2];
  const float* in0[2];
  const float* in1[2];
  float* dz[2];            // (2, rows, D)
  int64_t n;               // rows * D
};
int dz_build(hipStream_t st, const DzBuildArgs& a);

// S0 backward (rows), all layers, one workgroup per batch element.
struct QaBwdArgs {
  const float* dqbar;      // [L][B][Wd]
  const float* words;
  const int64_t* qlen;
  const float* params; int64_t off_w2, layer_stride;
  const float* saved; int64_t s_layer_stride, s_d, s_nrm, s_a, s_ah, s_S;
  float* gwords;           // (B, T, Wd) fully written
  float* dr;               // [L][B*T]
  float* dy;               // [L][B*T][D]
  float* rowdots;          // [2][L][B*T] scratch between the two passes
  int B, T, D, Wd, L;
  int first;               // 1: this launch initialises gwords, 0: it accumulates
  uint16_t* dy_p;          // bf16 planes of dy (nullable)
  int64_t lo_off;
};
int qa_bwd(hipStream_t st, const QaBwdArgs& a);

// Column sums (bias / vector-parameter gradients): out[c] = sum_r w[r] * M[r][c] over one or two row
// segments.  `chunks` > 1 splits the rows over that many workgroups which ADD into a pre-zeroed `out`
// with float atomics (the gradient buffer is zeroed at the start of the backward).
struct ColsumJob {
  const float* M0; const float* M1;   // row segments (M1 nullable)
  const float* w0; const float* w1;   // per-row weights, nullable = 1
  int rows0, rows1, cols;
  int64_t ld;
  float* out;
};
constexpr int kMaxColsumJobs = 16;
struct ColsumBatch { ColsumJob j[kMaxColsumJobs]; int n; int chunks; };
int colsum(hipStream_t st, const ColsumBatch& b);

// Zero a list of fp32 ranges in ONE launch (the bias / vector gradients that contractions add into, the alignment padding
// of the flat gradient buffer, d(dq)): replaces hipMemsetAsync over whole buffers whose large tensors are overwritten anyway.
constexpr int kMaxZeroSegs = 40;
struct ZeroList { float* p[kMaxZeroSegs]; int64_t n[kMaxZeroSegs]; int count; };
int zero_ranges(hipStream_t st, const ZeroList& z);
// zero_ranges(z) and elu_bwd(g, y, out, n, op, lo_off) as one launch (the head of the backward)
int bwd_head(hipStream_t st, const ZeroList& z, const float* g, const float* y, float* out, int64_t n, uint16_t* op = nullptr,
             int64_t lo_off = 0);

// fp32 -> bf16 shadow copy (round to nearest even)
int to_bf16(hipStream_t st, const float* src, void* dst, int64_t n);

}  // namespace dvgr
