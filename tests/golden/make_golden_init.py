#!/usr/bin/env python3
"""Initialisation statistics of a freshly constructed reference `DualVGR` (model/models.py:36-53: every nn.Linear / nn.LSTM
weight goes through `init_modules(..., 'xavier_uniform')`, model/utils.py:8-33, biases are zeroed, the embedding is
U(-1, 1)), recorded by IMPORTING the reference (build container only; nothing is copied).  Per state_dict key: shape,
max |w|, standard deviation, mean -- what tests/test_model_init.py holds a freshly constructed product model to.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_init.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

REF = "/root/reference"
_orig_to = torch.Tensor.to


def _to_cpu(self, *args, **kwargs):
    args = tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else a for a in args)
    if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
        kwargs["device"] = "cpu"
    return _orig_to(self, *args, **kwargs)


torch.Tensor.to = _to_cpu
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
import model.models as ref_models  # noqa: E402

DIMS = dict(V=256, D=256, Wd=64, N=5, L=2, vocab_q=200, vocab_a=40)

if __name__ == "__main__":
    d = DIMS
    vocab = {"question_token_to_idx": {str(i): i for i in range(d["vocab_q"])},
             "answer_token_to_idx": {str(i): i for i in range(d["vocab_a"])}}
    torch.manual_seed(1234)
    m = ref_models.DualVGR(vision_dim=d["V"], module_dim=d["D"], word_dim=d["Wd"], vocab=vocab, num_of_nodes=d["N"],
                           graph_module="GAT", graph_layers=1, unit_layers=d["L"])
    sd = {k: v for k, v in m.state_dict().items() if v.dtype == torch.float32}
    keys = sorted(sd)
    out = {"meta": np.array([d["V"], d["D"], d["Wd"], d["N"], d["L"], d["vocab_q"], d["vocab_a"]], dtype=np.int64),
           "keys": np.array(keys),
           "shapes": np.array([str(tuple(sd[k].shape)) for k in keys]),
           "absmax": np.array([float(sd[k].abs().max()) for k in keys]),
           "std": np.array([float(sd[k].double().std(unbiased=False)) for k in keys]),
           "mean": np.array([float(sd[k].double().mean()) for k in keys])}
    np.savez_compressed(os.path.join(HERE, "g_init_stats.npz"), **out)
    nz = [k for k in keys if k.endswith("bias") and float(sd[k].abs().max()) != 0]
    print(len(keys), "float tensors; non-zero biases:", nz)
