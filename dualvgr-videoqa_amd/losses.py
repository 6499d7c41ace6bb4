"""Auxiliary losses that consume the unit's list outputs -- the caller-side counterpart of
train.py:146-154 + utils.py:10-31 of the reference.

Two implementations of the same arithmetic:
  * common_loss / loss_dependence / unit_aux_losses / scope_u_loss: plain PyTorch on whatever device the unit's
    outputs live on (the reference hard-codes `.cuda()`, utils.py:22) -- the readable statement, used by tests;
  * fused_scope_u_loss / ScopeULoss: the HIP kernels of csrc/dvgr_loss.hip through `dvgr_scope_u_loss`
    (SURVEY.md section 8f row f1): value and gradient in three launches, no N x N autograd graph."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib as L_


def common_loss(emb1: torch.Tensor, emb2: torch.Tensor) -> torch.Tensor:
    """utils.py:10-18: centre over the node dim, L2-normalise rows, Gram matrices, MSE between them."""
    e1 = F.normalize(emb1 - emb1.mean(dim=1, keepdim=True), p=2, dim=2)
    e2 = F.normalize(emb2 - emb2.mean(dim=1, keepdim=True), p=2, dim=2)
    return ((e1 @ e1.transpose(1, 2) - e2 @ e2.transpose(1, 2)) ** 2).mean()


def loss_dependence(emb1: torch.Tensor, emb2: torch.Tensor, dim: int) -> torch.Tensor:
    """utils.py:20-31: HSIC = sum_b tr(R K1 R K2), R = I - 1/dim (a SUM over the batch)."""
    R = torch.eye(dim, dtype=emb1.dtype, device=emb1.device) - 1.0 / dim
    K1 = emb1 @ emb1.transpose(1, 2)
    K2 = emb2 @ emb2.transpose(1, 2)
    return torch.diagonal((R @ K1) @ (R @ K2), dim1=1, dim2=2).sum()


def unit_aux_losses(com_app, com_motion, aq_fusion, mq_fusion, num_nodes: int):
    """train.py:148-154: (sum_l common_l) / L' and (sum_l dep_l) / L' over the four lists."""
    n = len(aq_fusion)
    com = sum(common_loss(com_app[l], com_motion[l]) for l in range(n)) / n
    dep = sum(loss_dependence(aq_fusion[l], com_app[l], num_nodes) + loss_dependence(mq_fusion[l], com_motion[l], num_nodes)
              for l in range(n)) / n
    return com, dep


def scope_u_loss(outs, num_nodes: int, alpha: float = 1.0, beta: float = 1e-8, world_size: int = 1) -> torch.Tensor:
    """Scalar of the unit-only benchmark scope: sum(visual_feature) +
    alpha * common + beta * HSIC.  Under data parallelism with gradient AVERAGING the batch-SUM
    HSIC term is scaled by world_size so the single-process objective is kept (section 8e)."""
    vis, _, _, ca, cm, aq, mq = outs
    com, dep = unit_aux_losses(ca, cm, aq, mq, num_nodes)
    return vis.sum() + alpha * com + beta * world_size * dep


def fused_scope_u_loss(raw, alpha: float = 1.0, beta: float = 1e-8, world_size: int = 1, g_vis=None):
    """HIP path of scope_u_loss on the tensors `DualVGRUnit_multiple.forward_raw` returns
    (vis, aq_embed, mq_embed, com_app, com_mot, aq_fusion, mq_fusion; the lists as (L,B,N,D) tensors).

    Returns (terms, ups): `terms` = device tensor [sum(vis), mean_l common, mean_l dependence] and `ups` = the seven
    upstream gradients of loss = terms[0] + alpha*terms[1] + beta*world_size*terms[2], in forward_raw's order (None
    for aq_embed / mq_embed), ready for `torch.autograd.backward(raw, ups)`.  `g_vis`: optional preallocated
    (B,N,D) tensor that receives d loss / d vis (= 1)."""
    lib = L_.load()
    vis, _, _, ca, cm, aq, mq = raw
    if not ca.is_cuda:
        raise RuntimeError("fused_scope_u_loss runs on a ROCm device only (no CPU fallback)")
    ca, cm, aq, mq = (t.detach().contiguous() for t in (ca, cm, aq, mq))
    vis = None if vis is None else vis.detach().contiguous()     # None: the auxiliary terms alone
    Lr, B, N, D = ca.shape
    dims = L_.make_dims(B, N, 1, D, 4, Lr, 1, L_.DVGR_F32)
    L_.check(lib.dvgr_check_dims(C.byref(dims)), "dvgr_check_dims")
    dev = ca.device
    terms = torch.empty(3, device=dev, dtype=torch.float32)
    if g_vis is None and vis is not None:
        g_vis = torch.empty_like(vis)
    grads = [torch.empty_like(t) for t in (ca, cm, aq, mq)]
    ws = torch.empty(4 * (Lr * B * (4 * N * N + 4 * N * 64) + 128), device=dev, dtype=torch.uint8)
    io = L_.IO(None, None, None, None, None, None if vis is None else vis.data_ptr(), None, None, ca.data_ptr(),
               cm.data_ptr(), aq.data_ptr(), mq.data_ptr())
    stream = torch.cuda.current_stream(dev).cuda_stream
    L_.check(lib.dvgr_scope_u_loss(C.byref(dims), C.byref(io), float(alpha), float(beta), float(world_size),
                                   terms.data_ptr(), None if vis is None else g_vis.data_ptr(),
                                   *[g.data_ptr() for g in grads], ws.data_ptr(),
                                   ws.numel(), stream), "dvgr_scope_u_loss")
    return terms, [g_vis, None, None, *grads]


class ScopeULoss(torch.autograd.Function):
    """scope_u_loss as one autograd node: forward runs the fused kernels and keeps the gradients they produce,
    backward scales them by the incoming scalar.  `vis=None` gives alpha*common + beta*dependence alone (what
    train.py:148-154 adds to the cross-entropy)."""

    @staticmethod
    def forward(ctx, alpha, beta, world_size, vis, ca, cm, aq, mq):
        terms, ups = fused_scope_u_loss((vis, None, None, ca, cm, aq, mq), alpha, beta, world_size)
        ctx.has_vis = vis is not None
        ctx.save_for_backward(*([ups[0]] if ctx.has_vis else []), *ups[3:])
        ctx.mark_non_differentiable(terms)
        return terms[0] + alpha * terms[1] + beta * world_size * terms[2], terms

    @staticmethod
    def backward(ctx, g, _):
        grads = [g * t for t in ctx.saved_tensors]
        return (None, None, None, *([] if ctx.has_vis else [None]), *grads)
