/* dvgr_inputs.h -- C ABI of the two input-side operators of the reference model that sit between the feature files and the
 * unit and are neither an LSTM (include/dvgr_clip.h) nor the unit itself:
 *
 *   words  = tanh(dropout_0.15(embedding(questions)))        InputUnitLinguisticDynamic.forward,
 *                                                            /root/reference/model/Preprocessing.py:112-113
 *   motion = video_motion_feat . W^T + b                     DualVGR.visual_motion_input_unit = nn.Linear(vision_dim,
 *                                                            module_dim), /root/reference/model/models.py:46, :74
 *
 * Conventions as in include/dvgr.h: every pointer is a device pointer, fp32 unless stated, row-major; functions return 0,
 * a negative DVGR_E_* code or a positive hipError_t; launches go to `stream`, nothing is synchronised.  The dropout site
 * uses the library's counter hash (keep-mask of a seed: dvgr_words_dbg_keep_mask). */
#ifndef DVGR_INPUTS_H
#define DVGR_INPUTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* words[i][:] = tanh(keep_i * table[tokens[i]][:] / 0.85), i < n (= B * T token positions), rows of `wd` floats
 * (wd % 4 == 0).  train == 0: no dropout.  A token outside [0, vocab) reads row 0 and sets *bad_tokens (int32, device,
 * nullable) to 1 -- torch raises for it; the host side checks the flag where it can afford the synchronisation. */
int dvgr_words_forward(const float *table, int64_t vocab, int wd, const int64_t *tokens, int64_t n, int train,
                       uint64_t seed, float *words, int32_t *bad_tokens, void *stream);

/* g_table (vocab x wd) = scatter-add over i of g_words[i] * (1 - words[i]^2) * keep_i / 0.85 into row tokens[i].  The table
 * gradient is OVERWRITTEN (zeroed by the call, then added to with float atomics: the summation order over repeated tokens
 * is not fixed, as in torch's embedding backward). */
int dvgr_words_backward(const float *g_words, const float *words, const int64_t *tokens, int64_t n, int64_t vocab, int wd,
                        int train, uint64_t seed, float *g_table, void *stream);

/* Test aid: the 0/1 keep flags of the dropout site for `seed`, n * wd of them in element order. */
int dvgr_words_dbg_keep_mask(uint64_t seed, int64_t count, float *keep, void *stream);

/* y (M x N) = x (M x K) . w^T (w: N x K) + b (N, nullable).  dtype: DVGR_F32 (exact fp32 MFMA) / DVGR_BF16X3 / DVGR_BF16 as
 * in dvgr_dims.dtype; K % 4 == 0, N % 4 == 0. */
int dvgr_linear_forward(const float *x, const float *w, const float *b, float *y, int M, int N, int K, int dtype,
                        void *stream);

/* gw (N x K) = gy^T . x, gb (N, nullable) = column sums of gy, gx (M x K, nullable: the motion features are inputs and need
 * none) = gy . w.  All three are overwritten. */
int dvgr_linear_backward(const float *x, const float *w, const float *gy, float *gx, float *gw, float *gb, int M, int N,
                         int K, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVGR_INPUTS_H */
