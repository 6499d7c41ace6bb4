/* dvgr_optim.h -- C ABI of the optimiser step that follows the path's backward.
 *
 * Replaces, in the reference's training loop, `nn.utils.clip_grad_norm_(model.parameters(), max_norm=12)` (train.py:158)
 * and `optimizer.step()` of `optim.Adam(model.parameters(), cfg.train.lr)` (train.py:85, :159): two launches over a list of
 * flat segments instead of ~1000 small element-wise kernels (PyTorch's per-tensor loop over ~250 parameters).
 * Same conventions as dvgr.h: plain device pointers, nothing allocated, nothing synchronised, every launch on `stream`,
 * 0 / DVGR_E_* / hipError_t return codes.
 */
#ifndef DVGR_OPTIM_H
#define DVGR_OPTIM_H
#include "dvgr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One contiguous run of parameters with its gradient and Adam moments (all fp32, n elements, 4-byte aligned; 16-byte
 * aligned segments with n % 4 == 0 take the vector path).  The unit's flat parameter / gradient buffers are ONE segment. */
typedef struct dvgr_optim_seg {
  float *param;
  float *grad;      /* read by both calls; dvgr_optim_adam_step writes the clipped gradient back, as clip_grad_norm_ does */
  float *exp_avg;   /* Adam first moment  (state['exp_avg'])    */
  float *exp_avg_sq;/* Adam second moment (state['exp_avg_sq']) */
  int64_t n;
  void *planes;     /* nullable: bf16 hi plane of these parameters (dvgr_pack_weights layout: lo plane at planes + lo_off
                       elements); dvgr_optim_adam_step rewrites both planes from the updated values in the same pass */
  int64_t lo_off;
} dvgr_optim_seg;

#define DVGR_OPTIM_MAX_SEGS 4096

/* sqnorm[0] = sum over all segments of sum(grad^2): the square of clip_grad_norm_'s total_norm (norm_type 2: the 2-norm of
 * the per-tensor 2-norms = the global 2-norm).  `segs` is a HOST array; sqnorm is a device float, zeroed by the call. */
int dvgr_optim_grad_sqnorm(const dvgr_optim_seg *segs, int nseg, float *sqnorm, void *stream);

/* For every element: g *= min(1, max_norm / (sqrt(sqnorm[0]) + 1e-6)) (torch/nn/utils/clip_grad.py; skipped when
 * max_norm <= 0 or sqnorm == NULL), written back to grad; then torch.optim.Adam's update (amsgrad False, weight_decay 0,
 * maximize False):  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
 *                   p -= (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps),   step >= 1. */
int dvgr_optim_adam_step(const dvgr_optim_seg *segs, int nseg, const float *sqnorm, float max_norm, float lr, float beta1,
                         float beta2, float eps, int64_t step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVGR_OPTIM_H */
