"""Golden-case loader of the answer-head tests (tests/golden/g_head_*.npz, written by make_golden_head.py)."""
import os

import numpy as np
import torch

from dualvgr_videoqa_amd import synth
from oracle import head_torch as H

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEAD_CASES = ["g_head_small_eval", "g_head_small_train", "g_head_real_eval", "g_head_real_train", "g_head_n1_b3_train"]


class HeadCase:
    """Weights and inputs are regenerated from the (seed, name) generator exactly as make_golden_head.py built them."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.D, self.C, self.N, self.B, train, self.seed = [int(v) for v in self.z["meta"]]
        self.train = bool(train)
        D, C, N, B, seed = self.D, self.C, self.N, self.B, self.seed
        self.sd_np = synth.xavier_state_dict(H.param_shapes(D, C), seed=seed, bias_range=0.1)
        self.sd_np[H.PREFIX_O + "classifier.3.weight"] = (1.0 + 0.2 * synth.normal("bn_w", (D,), 1.0, seed)).astype(np.float32)
        self.sd_np[H.PREFIX_O + "classifier.3.bias"] = (0.1 * synth.normal("bn_b", (D,), 1.0, seed)).astype(np.float32)
        self.run_mean, self.run_var = self.z["run_mean"], self.z["run_var"]
        self.vis = synth.normal("head_vis", (B, N, D), 0.7, seed)
        self.qe = synth.normal("head_qe", (B, D), 0.5, seed)
        self.R = synth.normal("head_R", (B, C), 1.0, seed)

    def masks(self):
        if not self.train:
            return None
        shapes = {"agg": (self.B, self.N, self.D), "cat": (self.B, 2 * self.D), "bn": (self.B, self.D)}
        out = {}
        for nm, shape in shapes.items():
            bits = np.unpackbits(self.z["mask/" + nm])[: int(np.prod(shape))]
            out[nm] = torch.from_numpy(bits.reshape(shape).astype(np.float32))
        return out

    def check_param_grads(self, grads, tol):
        """Compact gradient evidence of the golden file: sum, sum of magnitudes and the first 16 entries per tensor."""
        names = [str(n) for n in self.z["gw_names"]]
        for i, k in enumerate(names):
            g = grads[k].astype(np.float64)
            if k.endswith("attn.bias"):
                # softmax over the clips is shift invariant: this gradient is analytically zero, both sides hold rounding
                ref_scale = float(self.z["gw_abs"][names.index(k[:-4] + "weight")])
                assert abs(g.sum()) <= 1e-4 * ref_scale and abs(float(self.z["gw_sum"][i])) <= 1e-4 * ref_scale, k
                continue
            scale = max(float(self.z["gw_abs"][i]), 1e-12)
            assert abs(np.abs(g).sum() - self.z["gw_abs"][i]) <= tol * scale, k
            assert abs(g.sum() - self.z["gw_sum"][i]) <= tol * scale, k
            head = np.resize(grads[k].ravel()[:16], 16)
            hs = max(float(np.abs(self.z["gw_head"][i]).max()), float(np.abs(g).max()) * 1e-3, 1e-12)
            assert np.abs(head - self.z["gw_head"][i]).max() <= tol * hs * 5, k
