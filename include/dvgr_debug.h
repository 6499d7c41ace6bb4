/*
 * dvgr_debug.h -- test hooks of libdvgr_hip.so: the building-block kernels behind single calls so
 * the parity tests can localise a mismatch (tests/test_hip_blocks.py).  Not part of the drop-in
 * surface; same conventions as dvgr.h (device pointers, caller's stream, 0 = success).
 */
#ifndef DVGR_DEBUG_H
#define DVGR_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One contraction C (+)= act(op(A) . op(B) + bias) on the MFMA engine (csrc/dvgr_gemm.h).
 *   mode 0 (NT): A[M,K] lda, B[N,K] ldb      -- forward Linear  (nn.Linear: y = x W^T + b)
 *   mode 1 (NN): A[M,K] lda, B[K,N] ldb      -- backward data
 *   mode 2 (TN): A[K,M] lda, B[K,N] ldb      -- backward weight; rows >= kseg of A/B come from A2/B2
 * act: 0 none, 1 tanh, 2 ELU.  drop_key_* = 0 disables the dropout masks.
 * prec: DVGR_F32 / DVGR_BF16 / DVGR_BF16X3 (dvgr.h). */
int dvgr_dbg_gemm(int mode, int M, int N, int K, int kseg, const float *A, const float *A2, int64_t lda,
                  const float *B, const float *B2, int64_t ldb, float *C, int64_t ldc, const float *bias,
                  int act, int accumulate, uint32_t drop_key_a, uint32_t drop_key_b, uint32_t drop_key_c,
                  int prec, void *stream);

/* bf16 hi / lo planes of an fp32 array: hi[i] = bf16(src[i]), hi[lo_off + i] = bf16(src[i] - hi[i]); n % 4 == 0. */
int dvgr_dbg_split(const float *src, void *hi, int64_t lo_off, int64_t n, void *stream);

/* dvgr_dbg_gemm with pre-split operands (planes made by dvgr_dbg_split; same shapes / leading dimensions as the
 * fp32 matrices, which are still passed for the fallback path).  act carries the dbg flags in bits 4.. */
int dvgr_dbg_gemm_planes(int mode, int M, int N, int K, const float *A, const void *A_hi, int64_t lda, int64_t a_lo_off,
                         const float *B, const void *B_hi, int64_t ldb, int64_t b_lo_off, float *C, int64_t ldc,
                         int act, int accumulate, int prec, void *stream);

/* Timing aid (tools/gemm_shapes.py): `count` (<= 8) copies of one pre-split problem in ONE launch, outputs count * M rows
 * apart, as the unit's launch sites batch modalities / views; kseg in (0, K): two-segment reduction; tag: launch-site tag
 * (0 generic .. 6, csrc/dvgr_gemm.h).  rowsum (nullable, mode 2 only; count * M floats, pre-zeroed): the bias gradient
 * sum_k A[k][m] is ADDED there (GemmProblem::rowsum; DVGR_E_DIMS when the engine cannot do it for this shape).
 * same_output != 0: every copy adds its tiles atomically into the FIRST output (pre-zeroed), as the batched d(dq). */
int dvgr_dbg_gemm_planes_batch(int mode, int M, int N, int K, int kseg, const float *A, const void *A_hi, int64_t lda,
                               int64_t a_lo_off, const float *B, const void *B_hi, int64_t ldb, int64_t b_lo_off, float *C,
                               int64_t ldc, int act, int accumulate, int atomic_ok, int prec, int count, int tag, float *rowsum,
                               int same_output, void *stream);

/* out[c] = sum_r w[r] * M[r][c]  (w nullable) */
int dvgr_dbg_colsum(const float *M, const float *w, int rows, int cols, int64_t ld, float *out, void *stream);

/* keep[i] = 1.0f / 0.0f for i in [0, n): the dropout keep decision of the device RNG for `key`. */
int dvgr_dbg_keep_mask(uint32_t key, int64_t n, float *keep, void *stream);

/* site key of (seed, layer, mod, gat, which) exactly as the forward derives it. */
uint32_t dvgr_dbg_site_key(uint64_t seed, int layer, int mod, int gat, int which);

/* Kernel launches the library has enqueued in this process so far (every launch site counts itself: DVGR_LAUNCH in
 * csrc/dvgr_common.h).  bench.py reports the difference over its timed steps as launches_per_step. */
long long dvgr_dbg_launch_count(void);

/* Timing events for dvgr_opts.probe_begin / probe_end, created with hipEventDisableSystemFence: the record then costs no
 * cache writeback / invalidate of its own, which is what the flag is documented for (more accurate timings of the commands
 * between two events).  bench.py times the dominant launch with them. */
int dvgr_dbg_event_create(void **event);
int dvgr_dbg_event_elapsed_ms(void *begin, void *end, float *ms);   /* synchronises on `end` first */
int dvgr_dbg_event_destroy(void *event);

/* The XCD block map of the contraction engine's 128 x 128-tile launches (csrc/dvgr_gemm.h, GemmBatch::xmap): bit t of `mask`
 * turns it on for launch-site tag t (0 generic .. 6).  Returns the previous mask; mask < 0 restores the build's default.
 * Any map gives bit-identical results (a workgroup's arithmetic does not depend on which tile list it came from) -- the
 * parity tests hold the engine to that (tests/test_hip_blocks.py). */
int dvgr_dbg_xmap_tags(int mask);

/* The planner of that map, host code only (no GPU needed): n (<= 12) problems of M[i] x N[i] outputs and reduction length K[i]
 * on bm x bn tiles, at most max_per_xcd tiles per XCD.  Returns the launch's grid extent (8 x the tiles of the busiest XCD), 0
 * when the batch does not fit the scheme (more than two reduction lengths, a class whose problem count does not divide 8, a
 * tile grid that cannot be cut into equal rectangles), < 0 on bad arguments.  seg_out (nullable, 80 ints): for XCD x and
 * class k the block {problem, first row tile, row tiles, first column tile, column tiles} at seg_out[(2 x + k) * 5]. */
int dvgr_dbg_plan_xcd_blocks(int n, const int *M, const int *N, const int *K, int bm, int bn, int max_per_xcd, int *seg_out);

#ifdef __cplusplus
}
#endif
#endif /* DVGR_DEBUG_H */
