/* dvgr_head.h -- C ABI of the step after the unit: clip read-out and classifier.
 *
 * Replaces, in the reference, `ContextSelfAttn.forward` (model/AnswerDecoder.py:165-182) followed by
 * `SimpleOutputUnitOpenEnded.forward` (model/AnswerDecoder.py:197-202) as `DualVGR.forward` chains them
 * (model/models.py:80-81), and the autograd graph PyTorch builds for them.  Same conventions as dvgr.h: plain device
 * pointers, nothing allocated, nothing synchronised, every launch on `stream`, 0 / DVGR_E_* / hipError_t return codes.
 */
#ifndef DVGR_HEAD_H
#define DVGR_HEAD_H
#include "dvgr.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvgr_head_dims {
  int32_t B;      /* QA pairs                                                        */
  int32_t N;      /* clips per video (the read-out softmax runs over them), <= 64    */
  int32_t D;      /* module_dim, multiple of 64                                      */
  int32_t C;      /* number of answer classes, multiple of 4 (pad the last Linear)   */
  int32_t dtype;  /* DVGR_F32 / DVGR_BF16 / DVGR_BF16X3: operand format of the contractions */
} dvgr_head_dims;

/* Parameters live in one flat fp32 buffer (offsets are multiples of 64 elements); names on the right are the
 * reference's state_dict keys under `feature_aggregation.` / `output_unit.`. */
enum dvgr_head_param {
  DVGR_HP_VPROJ_W = 0, /* (D, D)    feature_aggregation.v_proj.weight   (no bias, AnswerDecoder.py:159) */
  DVGR_HP_ATTN_W,      /* (1, D)    feature_aggregation.attn.weight                                      */
  DVGR_HP_ATTN_B,      /* (1)       feature_aggregation.attn.bias                                        */
  DVGR_HP_QPROJ_W,     /* (D, D)    output_unit.question_proj.weight                                     */
  DVGR_HP_QPROJ_B,     /* (D)       output_unit.question_proj.bias                                       */
  DVGR_HP_FC1_W,       /* (D, 2D)   output_unit.classifier.1.weight                                      */
  DVGR_HP_FC1_B,       /* (D)       output_unit.classifier.1.bias                                        */
  DVGR_HP_BN_W,        /* (D)       output_unit.classifier.3.weight  (BatchNorm1d gamma)                 */
  DVGR_HP_BN_B,        /* (D)       output_unit.classifier.3.bias    (BatchNorm1d beta)                  */
  DVGR_HP_FC2_W,       /* (C, D)    output_unit.classifier.5.weight                                      */
  DVGR_HP_FC2_B,       /* (C)       output_unit.classifier.5.bias                                        */
  DVGR_HP_KINDS
};

int dvgr_head_check_dims(const dvgr_head_dims *d);
int64_t dvgr_head_param_count(const dvgr_head_dims *d);
int dvgr_head_param_offset(const dvgr_head_dims *d, int kind, int64_t *offset, int64_t *rows, int64_t *cols);
size_t dvgr_head_saved_bytes(const dvgr_head_dims *d);
size_t dvgr_head_workspace_bytes(const dvgr_head_dims *d);

typedef struct dvgr_head_io {
  const float *vis;     /* visual_feature       (B, N, D): the unit's first output              */
  const float *qe;      /* question_embedding   (B, D)                                          */
  float *logits;        /* (B, C)                                                               */
  float *run_mean;      /* (D) BatchNorm1d running_mean: read in eval mode, updated in training */
  float *run_var;       /* (D) BatchNorm1d running_var (unbiased), likewise                     */
} dvgr_head_io;

typedef struct dvgr_head_grads {
  const float *g_logits; /* (B, C) upstream gradient                                  */
  float *g_vis;          /* (B, N, D)                                                 */
  float *g_qe;           /* (B, D)                                                    */
  float *gparams;        /* flat, layout of the parameter buffer; every element written */
} dvgr_head_grads;

/* train & 1: the three nn.Dropout(0.15) sites are active (counter-hash RNG of the unit, keyed by `seed`) and
 * BatchNorm1d uses batch statistics and updates the running ones (momentum 0.1, eps 1e-5).
 * train & DVGR_HEAD_TRAIN_NO_DROPOUT: training-mode BatchNorm with the dropout sites switched off (p = 0) -- what a
 * deterministic comparison with the reference's training step needs (tests/golden/make_golden_step.py patches p to 0).
 * The backward enqueues the weight gradients and column sums on the library's internal side stream (see
 * DVGR_OPT_SINGLE_STREAM in dvgr.h); it forks from and joins back into `stream` inside the call. */
#define DVGR_HEAD_TRAIN_NO_DROPOUT 2
int dvgr_head_forward(const dvgr_head_dims *d, const float *params, const dvgr_head_io *io, void *saved,
                      size_t saved_bytes, void *workspace, size_t workspace_bytes, int train, uint64_t seed,
                      void *stream);
int dvgr_head_backward(const dvgr_head_dims *d, const float *params, const dvgr_head_io *io,
                       const dvgr_head_grads *g, const void *saved, size_t saved_bytes, void *workspace,
                       size_t workspace_bytes, int train, uint64_t seed, void *stream);

/* Cross-entropy of the logits against the answer indices (`nn.CrossEntropyLoss()` of train.py:93, :147; mean over the
 * batch), value and gradient in one launch: loss[0] = mean_b (logsumexp(logits_b) - logits_b[target_b]),
 * g_logits = (softmax(logits) - onehot(target)) * scale / B (the padding columns C_valid..C-1 get 0 and are ignored).
 * `target` holds int64 class indices < C_valid; `scale` multiplies the gradient (loss weights, 1 / world size). */
int dvgr_head_cross_entropy(const dvgr_head_dims *d, int C_valid, const float *logits, const int64_t *target, float scale,
                            float *loss, float *g_logits, void *stream);

/* test hook: element keep-mask (1 / 0) of dropout site 0 (read-out input), 1 (concatenation), 2 (after BatchNorm) */
int dvgr_head_dbg_keep_mask(uint64_t seed, int site, int64_t n, float *keep, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVGR_HEAD_H */
