#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference implementation.

Runs only in the build container (needs /root/reference; the GPU box does not have it).  The
reference sources are imported in place and never copied: this script only records inputs,
weights (or the generator key that reproduces them), outputs, losses and gradients as .npz data.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Shims applied at import time (reference files untouched):
  * `Tensor.to('cuda:1')` -> CPU  (model/models.py:118-119, model/utils.py:72 hard-code the device)
  * `Tensor.cuda()` -> identity   (utils.py:22)
  * `F.dropout` replaced by a recording version in train-mode cases, so the Bernoulli keep-masks
    the reference drew can be injected into the oracle / replayed by the HIP path.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"

# ---- shims ---------------------------------------------------------------------------------
_orig_to = torch.Tensor.to


def _to_cpu(self, *args, **kwargs):
    args = tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else a for a in args)
    if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
        kwargs["device"] = "cpu"
    return _orig_to(self, *args, **kwargs)


torch.Tensor.to = _to_cpu
torch.Tensor.cuda = lambda self, *a, **k: self

sys.path.insert(0, REF)
import model.models as ref_models  # noqa: E402  (the reference, imported in place)
import importlib.util as _ilu  # noqa: E402

_spec = _ilu.spec_from_file_location("ref_top_utils", os.path.join(REF, "utils.py"))
ref_utils = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(ref_utils)

from dualvgr_videoqa_amd import synth  # noqa: E402
from oracle import unit_torch as O  # noqa: E402

torch.set_num_threads(4)


class DropoutRecorder:
    """Stand-in for F.dropout inside the reference: draws the mask with torch's RNG like the
    original would, records it, applies inverted scaling."""

    def __init__(self):
        self.masks = []

    def __call__(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = torch.bernoulli(torch.full_like(x, 1.0 - p))
        self.masks.append(keep)
        return x * keep / (1.0 - p)


def site_order(unit_layers, graph_layers):
    """Order in which the reference calls F.dropout (model/models.py:150-160 then
    model/GraphNN.py:175-177 and :107), mapped to the oracle's site names."""
    names = []
    for i in range(unit_layers):
        for fams in (("acGCN", "appearance_GCN"), ("mcGCN", "motion_GCN")):
            for j in range(graph_layers):
                for fam in fams:
                    base = f"L{i}.G{j}.{fam}"
                    names.append(base + ".x")
                    names += [f"{base}.h{h}.alpha" for h in range(4)]
                    names.append(base + ".out")
    return names


def build_ref_unit(sd_np, word_dim, module_dim, N, L, G=1):
    unit = ref_models.DualVGRUnit_multiple(word_dim=word_dim, module_dim=module_dim, num_of_nodes=N,
                                           appearance_graph_layers=G, motion_graph_layers=G,
                                           graph_module="GAT", unit_layers=L)
    unit.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    return unit


def upstream_weights(B, N, D, L, G, seed):
    """Fixed pseudo-random upstream-gradient weights so that every output gets a distinct,
    non-uniform gradient (a plain sum would hide transposition mistakes)."""
    r = {"R_vis": synth.normal("R_vis", (B, N, D), 1.0, seed),
         "R_aq": synth.normal("R_aq", (B, N, D), 0.3, seed),
         "R_mq": synth.normal("R_mq", (B, N, D), 0.3, seed)}
    for l in range(L * G):
        for nm in ("ca", "cm", "aq", "mq"):
            r[f"R_{nm}{l}"] = synth.normal(f"R_{nm}{l}", (B, N, D), 0.2, seed)
    return r


def scalar_loss(outs, R, N, alpha, beta, common_loss, loss_dependence):
    vis, aqe, mqe, ca, cm, aq, mq = outs
    n = len(aq)
    loss = (vis * R["R_vis"]).sum() + (aqe * R["R_aq"]).sum() + (mqe * R["R_mq"]).sum()
    com = 0
    dep = 0
    for l in range(n):
        com = com + common_loss(ca[l], cm[l])
        dep = dep + loss_dependence(aq[l], ca[l], N) + loss_dependence(mq[l], cm[l], N)
        loss = loss + (ca[l] * R[f"R_ca{l}"]).sum() + (cm[l] * R[f"R_cm{l}"]).sum()
        loss = loss + (aq[l] * R[f"R_aq{l}"]).sum() + (mq[l] * R[f"R_mq{l}"]).sum()
    loss = loss + alpha * com / n + beta * dep / n
    return loss, com / n, dep / n


def run_case(name, *, Wd, D, N, L, B, T, qlen=None, G=1, train=False, seed=666, full=True,
             alpha=1.0, beta=1e-3, bias_range=0.1, big_limit=1 << 30):
    shapes = O.param_shapes(Wd, D, L, G)
    sd_np = synth.xavier_state_dict(shapes, seed=seed, bias_range=bias_range)
    inp = synth.unit_inputs(B, N, T, D, Wd, seed=seed)
    if qlen is not None:
        inp["qlen"] = np.asarray(qlen, dtype=np.int64)
        valid = (np.arange(T)[None, :] < inp["qlen"][:, None]).astype(np.float32)
        inp["dq"] = synth.normal("dq", (B, T, D), 0.5, seed) * valid[:, :, None]
    unit = build_ref_unit(sd_np, Wd, D, N, L, G)
    assert set(unit.state_dict().keys()) == set(shapes.keys()), "state_dict key mismatch"
    unit.train(train)
    rec = DropoutRecorder()
    if train:
        torch.manual_seed(seed)
        ref_models.F.dropout = rec          # `from .X import *` chains share torch.nn.functional
        assert F.dropout is rec

    t = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    R = {k: torch.from_numpy(v) for k, v in upstream_weights(B, N, D, L, G, seed).items()}
    outs = unit(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    loss, com, dep = scalar_loss(outs, R, N, alpha, beta, ref_utils.common_loss, ref_utils.loss_dependence)
    loss.backward()
    if train:
        F.dropout = _REAL_DROPOUT
        ref_models.F.dropout = _REAL_DROPOUT

    vis, aqe, mqe, ca, cm, aq, mq = outs
    out = {"meta": np.array([Wd, D, N, L, B, T, G, int(train), seed], dtype=np.int64),
           "meta_f": np.array([alpha, beta, bias_range], dtype=np.float64),
           "qlen": inp["qlen"], "loss": loss.detach().numpy(), "common": com.detach().numpy(),
           "dep": dep.detach().numpy()}
    # modules never reached by the [i+j] indexing at graph_layers > 1 get no gradient
    grads = {n_: (p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
             for n_, p in unit.named_parameters()}
    # weights are regenerable from (seed, name); commit a checksum to detect generator drift
    out["w_checksum"] = np.array([float(np.abs(sd_np[k].astype(np.float64)).sum()) for k in sorted(sd_np)])
    if full:
        # small case: commit inputs, outputs and gradients verbatim
        for k in ("A", "M", "dq", "words"):
            out["in/" + k] = inp[k]
            out["gin/" + k] = t[k].grad.numpy()
        out["out/vis"] = vis.detach().numpy()
        out["out/aq_embed"] = aqe.detach().numpy()
        out["out/mq_embed"] = mqe.detach().numpy()
        for l in range(len(aq)):
            out[f"out/com_app{l}"] = ca[l].detach().numpy()
            out[f"out/com_motion{l}"] = cm[l].detach().numpy()
            out[f"out/aq_fusion{l}"] = aq[l].detach().numpy()
            out[f"out/mq_fusion{l}"] = mq[l].detach().numpy()
        big = [k for k, v in grads.items() if v.size > big_limit]
        for k, v in grads.items():
            if k not in big:
                out["gw/" + k] = v
        if big:
            out["gwbig_names"] = np.array(sorted(big))
            out["gwbig_sum"] = np.array([grads[k].astype(np.float64).sum() for k in sorted(big)])
            out["gwbig_abs"] = np.array([np.abs(grads[k].astype(np.float64)).sum() for k in sorted(big)])
            out["gwbig_head"] = np.stack([grads[k].ravel()[:64] for k in sorted(big)])
    else:
        # real-dim case: weights/inputs are regenerable from (seed, name); commit outputs and
        # compact gradient evidence only
        last = len(aq) - 1
        out["out/vis"] = vis.detach().numpy()
        out[f"out/com_app{last}"] = ca[last].detach().numpy()
        out[f"out/aq_fusion{last}"] = aq[last].detach().numpy()
        out["out/mq_embed"] = mqe.detach().numpy()
        out["gin/A"] = t["A"].grad.numpy()
        out["gin/M"] = t["M"].grad.numpy()
        out["gin/dq"] = t["dq"].grad.numpy()
        out["gin/words"] = t["words"].grad.numpy()
        names = sorted(grads.keys())
        out["gw_names"] = np.array(names)
        out["gw_sum"] = np.array([grads[k].astype(np.float64).sum() for k in names])
        out["gw_abs"] = np.array([np.abs(grads[k].astype(np.float64)).sum() for k in names])
        out["gw_head"] = np.stack([np.resize(grads[k].ravel()[:8], 8) for k in names])
    if train:
        order = site_order(L, G)
        assert len(order) == len(rec.masks), (len(order), len(rec.masks))
        for nm, m in zip(order, rec.masks):
            out["mask/" + nm] = np.packbits(m.numpy().astype(np.uint8).ravel())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={float(loss):.6f} common={float(com):.6e} dep={float(dep):.6e} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


_REAL_DROPOUT = F.dropout

if __name__ == "__main__":
    # G-small (+ quirk lengths 1 and T), eval and train (recorded masks)
    run_case("g_small_eval", Wd=32, D=64, N=5, L=2, B=3, T=6, qlen=[6, 4, 1])
    run_case("g_small_train", Wd=32, D=64, N=5, L=2, B=3, T=6, qlen=[6, 4, 1], train=True, big_limit=8192)
    # G-quirk: N=1 (single-node graph), graph_layers=2 (module index overlap, model/models.py:151)
    run_case("g_quirk_n1", Wd=32, D=64, N=1, L=1, B=2, T=4, qlen=[4, 2], big_limit=8192)
    run_case("g_quirk_g2", Wd=32, D=64, N=4, L=2, B=2, T=5, qlen=[5, 3], G=2, big_limit=8192, alpha=0.0)  # common_loss is ill-conditioned here (near-identical rows)
    # G-real: D=768, Wd=300 at the clip counts of the three datasets
    run_case("g_real_n20_l1", Wd=300, D=768, N=20, L=1, B=2, T=9, full=False, beta=1e-6)
    run_case("g_real_n20_l3", Wd=300, D=768, N=20, L=3, B=2, T=9, full=False, beta=1e-6)
    run_case("g_real_n8_l3", Wd=300, D=768, N=8, L=3, B=2, T=9, full=False, beta=1e-6)
    run_case("g_real_n16_l1", Wd=300, D=768, N=16, L=1, B=2, T=9, full=False, beta=1e-6)
