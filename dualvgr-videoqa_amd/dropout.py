"""numpy replica (bit for bit) of the device dropout RNG in csrc/dvgr_common.h, so tests can feed
the exact keep-masks a training-mode HIP forward used into the oracle (mask-injected parity)."""
from __future__ import annotations

import numpy as np

from .synth import mix32

DROP_P = 0.15
THRESH16 = 9830          # floor(0.15 * 65536)
GAT_FAMILIES = (("acGCN", "appearance_GCN"), ("mcGCN", "motion_GCN"))   # [mod][gat]


def _mix32_scalar(x: int) -> int:
    return int(mix32(np.array([x & 0xFFFFFFFF], dtype=np.uint32))[0])


def site_key(seed: int, site: int) -> int:
    k = _mix32_scalar(site + 0x9E3779B1)
    k = _mix32_scalar(k ^ ((seed >> 32) & 0xFFFFFFFF))
    k = _mix32_scalar(k ^ (seed & 0xFFFFFFFF))
    return k | 1


def drop_site(layer: int, mod: int, gat: int, which: int) -> int:
    return ((layer * 2 + mod) * 2 + gat) * 6 + which


def keep_mask(key: int, n: int) -> np.ndarray:
    """0/1 keep flags of elements 0..n-1 for one site key."""
    idx = np.arange(n, dtype=np.uint64)
    r = mix32((((idx >> np.uint64(1)) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) ^ np.uint64(key))
    r = r.astype(np.uint64)
    f = np.where(idx & np.uint64(1), r >> np.uint64(16), r & np.uint64(0xFFFF))
    return (f >= THRESH16).astype(np.float32)


def site_masks(seed: int, B: int, N: int, D: int, L: int, G: int = 1) -> dict:
    """Every keep-mask of one training forward, keyed by the oracle's site names
    (oracle/unit_torch.py: 'L{l}.G{j}.{family}.x', '.h{h}.alpha', '.out').  The device numbers the sites of unit layer l,
    graph layer j as "layer" l * G + j (csrc/dvgr_api.hip)."""
    out = {}
    for l in range(L):
        for j in range(G):
            e = l * G + j
            for mod in range(2):
                for gat in range(2):
                    base = f"L{l}.G{j}.{GAT_FAMILIES[mod][gat]}"
                    out[base + ".x"] = keep_mask(site_key(seed, drop_site(e, mod, gat, 0)), B * N * D).reshape(B, N, D)
                    out[base + ".out"] = keep_mask(site_key(seed, drop_site(e, mod, gat, 5)), B * N * D).reshape(B, N, D)
                    for h in range(4):
                        out[f"{base}.h{h}.alpha"] = keep_mask(site_key(seed, drop_site(e, mod, gat, 1 + h)), B * N * N).reshape(B, N, N)
    return out
