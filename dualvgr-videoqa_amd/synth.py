"""Framework-independent deterministic generator for synthetic weights and inputs.

Values depend only on (seed, tensor name, element index) through a 32-bit counter hash, so the
same tensors can be regenerated on any machine (the GPU box has no access to the reference or to
files outside this repository).  Used by bench.py (synthetic data of SURVEY.md section 8(d)), by
the golden-vector script and by the parity tests.
"""
from __future__ import annotations

import math
import zlib

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def mix32(x: np.ndarray) -> np.ndarray:
    """'lowbias32' integer hash on uint32 arrays (computed in uint64 to dodge overflow warnings).
    The device dropout RNG in csrc/dvgr_common.h uses the same function bit for bit."""
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def name_key(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


def uniform01(name: str, n: int, seed: int = 666) -> np.ndarray:
    """n float64 values in [0, 1) keyed by (seed, name, index)."""
    idx = np.arange(n, dtype=np.uint64)
    key = np.uint64(name_key(name, seed))
    h = mix32(((idx * np.uint64(0x9E3779B1)) + key) & _M32)
    h = mix32(h.astype(np.uint64) ^ (idx >> np.uint64(32)) ^ key)
    return h.astype(np.float64) / 4294967296.0


def uniform(name: str, shape, lo: float, hi: float, seed: int = 666) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * uniform01(name, n, seed)).reshape(shape).astype(np.float32)


def normal(name: str, shape, std: float = 1.0, seed: int = 666) -> np.ndarray:
    """Box-Muller on two hashed uniforms."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(name + "#u1", n, seed)
    u2 = uniform01(name + "#u2", n, seed)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
    return (std * z).reshape(shape).astype(np.float32)


def xavier_state_dict(shapes: dict, seed: int = 666, bias_range: float = 0.0) -> dict:
    """Xavier-uniform weights (gain 1) as model/utils.py:8-33 `init_modules` produces; biases are
    zero there -- `bias_range` > 0 draws them from U(-r, r) instead so tests exercise them."""
    sd = {}
    for name, shape in shapes.items():
        if len(shape) == 2:
            fan_out, fan_in = shape
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            sd[name] = uniform(name, shape, -bound, bound, seed)
        elif bias_range > 0.0:
            sd[name] = uniform(name, shape, -bias_range, bias_range, seed)
        else:
            sd[name] = np.zeros(shape, dtype=np.float32)
    return sd


def unit_inputs(B: int, N: int, T: int, D: int, Wd: int, seed: int = 666) -> dict:
    """Scope-U synthetic inputs of SURVEY.md section 8(d): A, M ~ 0.5*N(0,1); dq ~ 0.5*N(0,1)
    zeroed past question_len; words = tanh(N(0,1)); question_len ~ U{3..T} with sample 0 forced
    to T and (when B > 1) sample 1 forced to min(3, T)."""
    qlen = (3 + np.floor(uniform01("qlen", B, seed) * max(T - 2, 1))).astype(np.int64)
    qlen = np.clip(qlen, 1, T)
    qlen[0] = T
    if B > 1:
        qlen[1] = min(3, T)
    A = normal("A", (B, N, D), 0.5, seed)
    M = normal("M", (B, N, D), 0.5, seed)
    dq = normal("dq", (B, T, D), 0.5, seed)
    words = np.tanh(normal("words", (B, T, Wd), 1.0, seed))
    valid = (np.arange(T)[None, :] < qlen[:, None])
    dq = dq * valid[:, :, None].astype(np.float32)
    return {"A": A, "M": M, "dq": dq.astype(np.float32), "words": words.astype(np.float32), "qlen": qlen}
