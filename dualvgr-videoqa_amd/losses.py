"""Auxiliary losses that consume the unit's list outputs -- the caller-side counterpart of
train.py:146-154 + utils.py:10-31 of the reference.

Plain PyTorch on whatever device the unit's outputs live on (the reference hard-codes `.cuda()`,
utils.py:22).  These are harness code around the hot path, not part of the HIP kernels yet
(SURVEY.md section 8f ranks fusing them with the unit's tail as the next row)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


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
