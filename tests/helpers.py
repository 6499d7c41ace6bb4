"""Shared helpers for the parity tests: golden-case loading and error metrics."""
import os

import numpy as np
import torch

from dualvgr_videoqa_amd import synth
from oracle import unit_torch as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_CASES = ["g_small_eval", "g_small_train", "g_quirk_n1", "g_quirk_g2"]
REAL_CASES = ["g_real_n20_l1", "g_real_n20_l3", "g_real_n8_l3", "g_real_n16_l1"]


class GoldenCase:
    """One .npz written by tests/golden/make_golden.py, with weights/inputs regenerated from
    the (seed, name) generator exactly as that script built them."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        (self.Wd, self.D, self.N, self.L, self.B, self.T, self.G, train, self.seed) = [int(v) for v in self.z["meta"]]
        self.train = bool(train)
        self.alpha, self.beta, self.bias_range = [float(v) for v in self.z["meta_f"]]
        self.full = "in/A" in self.z.files
        shapes = O.param_shapes(self.Wd, self.D, self.L, self.G)
        self.sd_np = synth.xavier_state_dict(shapes, seed=self.seed, bias_range=self.bias_range)
        chk = np.array([float(np.abs(self.sd_np[k].astype(np.float64)).sum()) for k in sorted(self.sd_np)])
        assert np.allclose(chk, self.z["w_checksum"], rtol=1e-12), "weight generator drifted"
        inp = synth.unit_inputs(self.B, self.N, self.T, self.D, self.Wd, seed=self.seed)
        qlen = self.z["qlen"].astype(np.int64)
        if not np.array_equal(qlen, inp["qlen"]):
            valid = (np.arange(self.T)[None, :] < qlen[:, None]).astype(np.float32)
            inp["dq"] = synth.normal("dq", (self.B, self.T, self.D), 0.5, self.seed) * valid[:, :, None]
            inp["qlen"] = qlen
        self.inp = inp
        if self.full:
            for k in ("A", "M", "dq", "words"):
                assert np.array_equal(self.inp[k], self.z["in/" + k]), "input generator drifted: " + k

    # upstream-gradient weights, as make_golden.upstream_weights
    def upstream(self):
        r = {"R_vis": synth.normal("R_vis", (self.B, self.N, self.D), 1.0, self.seed),
             "R_aq": synth.normal("R_aq", (self.B, self.N, self.D), 0.3, self.seed),
             "R_mq": synth.normal("R_mq", (self.B, self.N, self.D), 0.3, self.seed)}
        for l in range(self.L * self.G):
            for nm in ("ca", "cm", "aq", "mq"):
                r[f"R_{nm}{l}"] = synth.normal(f"R_{nm}{l}", (self.B, self.N, self.D), 0.2, self.seed)
        return r

    def masks(self):
        if not self.train:
            return None
        out = {}
        for site, shape in O.dropout_site_shapes(self.B, self.N, self.D, self.L, self.G).items():
            bits = np.unpackbits(self.z["mask/" + site])[: int(np.prod(shape))]
            out[site] = torch.from_numpy(bits.reshape(shape).astype(np.float32))
        return out

    def torch_state(self, device="cpu", dtype=torch.float32):
        return {k: torch.from_numpy(v.copy()).to(device=device, dtype=dtype) for k, v in self.sd_np.items()}

    def torch_inputs(self, device="cpu", dtype=torch.float32, requires_grad=False):
        t = {}
        for k in ("A", "M", "dq", "words"):
            t[k] = torch.from_numpy(self.inp[k].copy()).to(device=device, dtype=dtype).requires_grad_(requires_grad)
        t["qlen"] = torch.from_numpy(self.inp["qlen"].copy()).to(device)
        return t


def scalar_loss(outs, R, N, alpha, beta, common_loss=O.common_loss, loss_dependence=O.loss_dependence):
    """The scalar the golden gradients were taken of (make_golden.scalar_loss)."""
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
    return loss + alpha * com / n + beta * dep / n, com / n, dep / n


def rel_err(got, want):
    """max |got - want| / max |want| (the 'relative fp32' metric of BASELINE.json's north_star)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    denom = max(float(np.abs(want).max()), 1e-30)
    return float(np.abs(got - want).max()) / denom
