/* dvgr_clip.h -- C ABI of the appearance encoder in front of the unit (second stage).
 *
 * Replaces, in the reference, `VisualAppearanceEncoder.forward` (model/Preprocessing.py:191-234): dropout(0.15) -> tanh,
 * a one-layer bidirectional LSTM over the T frames of every clip (sequence-first, batch = S = B * N clips), the two final
 * hidden states concatenated, dropout(0.18) -- and the autograd graph PyTorch builds for it (cuDNN/MIOpen RNN).  The
 * LSTM follows torch.nn.LSTM's published cell (gate order i, f, g, o; two bias vectors):
 *     i, f, g, o = split(W_ih x_t + b_ih + W_hh h_{t-1} + b_hh)
 *     c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g);   h_t = sigmoid(o) tanh(c_t)
 * Same conventions as dvgr.h: plain device pointers, nothing allocated, nothing synchronised, every launch on `stream`,
 * 0 / DVGR_E_* / hipError_t return codes.  The frame features are data, so no gradient is produced for them.
 */
#ifndef DVGR_CLIP_H
#define DVGR_CLIP_H
#include "dvgr.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvgr_clip_dims {
  int32_t S;      /* clips = B * N (the LSTM's batch)                                 */
  int32_t T;      /* frames per clip (the LSTM's sequence length), 1..64             */
  int32_t V;      /* input feature size (2048 frames / 300 words), multiple of 4     */
  int32_t H;      /* hidden size per direction = module_dim / 2, multiple of 32      */
  int32_t dtype;  /* DVGR_F32 / DVGR_BF16 / DVGR_BF16X3: operand format of the contractions */
} dvgr_clip_dims;

/* Parameters live in one flat fp32 buffer; names on the right are the reference's state_dict keys under
 * `visual_appearance_input_unit.encoder.`.  The two W_ih are adjacent (one (8H, V) matrix for the input contraction). */
enum dvgr_clip_param {
  DVGR_CP_WIH_F = 0,  /* (4H, V)  weight_ih_l0          */
  DVGR_CP_WIH_R,      /* (4H, V)  weight_ih_l0_reverse  */
  DVGR_CP_WHH_F,      /* (4H, H)  weight_hh_l0          */
  DVGR_CP_WHH_R,      /* (4H, H)  weight_hh_l0_reverse  */
  DVGR_CP_BIH_F,      /* (4H)     bias_ih_l0            */
  DVGR_CP_BIH_R,      /* (4H)     bias_ih_l0_reverse    */
  DVGR_CP_BHH_F,      /* (4H)     bias_hh_l0            */
  DVGR_CP_BHH_R,      /* (4H)     bias_hh_l0_reverse    */
  DVGR_CP_KINDS
};

int dvgr_clip_check_dims(const dvgr_clip_dims *d);
int64_t dvgr_clip_param_count(const dvgr_clip_dims *d);
int dvgr_clip_param_offset(const dvgr_clip_dims *d, int kind, int64_t *offset, int64_t *rows, int64_t *cols);
/* Bytes of the buffer the forward fills for the backward (activations of every step), and of the scratch workspace
 * either call needs (max of the two).  Both grow with S * T. */
size_t dvgr_clip_saved_bytes(const dvgr_clip_dims *d);
size_t dvgr_clip_workspace_bytes(const dvgr_clip_dims *d);

/* x: (S, T, V) fp32 frame features (= the reference's (B, N, frames, feat) tensor, contiguous);
 * out: (S, 2H) fp32 = [h_forward(T-1) | h_reverse(0)] after the output dropout.
 * train != 0: the two dropout sites draw their keep-masks from `seed` (counter hash, dvgr_clip_dbg_keep_mask
 * reproduces them); train == 0: no dropout, `seed` ignored. */
int dvgr_clip_forward(const dvgr_clip_dims *d, const float *params, const float *x, float *out, void *saved,
                      size_t saved_bytes, void *workspace, size_t workspace_bytes, int train, uint64_t seed, void *stream);

/* g_out: (S, 2H) gradient of `out`; gparams: flat buffer like `params`, fully written.  `saved` is the buffer the
 * matching forward filled (it is consumed: the step gradients overwrite the gate activations); `train`/`seed` as there. */
int dvgr_clip_backward(const dvgr_clip_dims *d, const float *params, const float *g_out, float *gparams, void *saved,
                       size_t saved_bytes, void *workspace, size_t workspace_bytes, int train, uint64_t seed, void *stream);

/* ---- the same engine as a general one-layer bidirectional LSTM over padded, variable-length sequences ----------------
 * Replaces `DynamicRNN.forward` (model/Preprocessing.py:17-45: sort, pack_padded_sequence, nn.LSTM, pad_packed_sequence,
 * un-sort) and the packed `self.encoder` call of `InputUnitLinguisticDynamic.forward` (model/Preprocessing.py:116-121).
 * Packing semantics without packing: sequence s is active at step t iff t < lens[s]; an inactive step leaves (h, c) as
 * they were and emits zeros, so the forward direction's final state is the one at lens[s]-1 and the reverse direction
 * starts at lens[s]-1 from zeros -- exactly what pack_padded_sequence gives.  `d->V` only has to be a multiple of 4 here
 * (word vectors are 300 wide). */
#define DVGR_LSTM_EMBED_INPUT 1 /* x is raw: apply dropout(0.15) then tanh first (the appearance encoder's prologue) */
#define DVGR_LSTM_DROP_FINAL  2 /* dropout(0.18) on out_final                                                      */

typedef struct dvgr_lstm_io {
  const float *x;        /* (S, T, V) batch-first input                                          */
  const int64_t *lens;   /* (S) lengths in 1..T, or NULL: every sequence has T steps             */
  float *out_final;      /* (S, 2H) [h_forward(last) | h_reverse(first)], or NULL               */
  float *out_seq;        /* (S, T, 2H) hidden states of every step, zeros past lens[s], or NULL */
} dvgr_lstm_io;

typedef struct dvgr_lstm_grads {
  const float *g_final;  /* (S, 2H) or NULL (= zeros)                                  */
  const float *g_seq;    /* (S, T, 2H) or NULL (= zeros)                               */
  float *g_x;            /* (S, T, V) fully written, or NULL: not wanted               */
  float *gparams;        /* flat buffer like `params`, fully written                   */
} dvgr_lstm_grads;

int dvgr_lstm_forward(const dvgr_clip_dims *d, const float *params, const dvgr_lstm_io *io, int flags, void *saved,
                      size_t saved_bytes, void *workspace, size_t workspace_bytes, int train, uint64_t seed, void *stream);
int dvgr_lstm_backward(const dvgr_clip_dims *d, const float *params, const dvgr_lstm_io *io, const dvgr_lstm_grads *grads,
                       int flags, void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes, int train,
                       uint64_t seed, void *stream);

/* Test aid: 0/1 keep flags of dropout site `site` (0: input, p = 0.15, n = S*T*V; 1: output, p = 0.18, n = S*2H). */
int dvgr_clip_dbg_keep_mask(uint64_t seed, int site, int64_t n, float *keep, void *stream);

#ifdef __cplusplus
}
#endif
#endif
