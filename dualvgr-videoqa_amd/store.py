"""Feature store + batch loader on libdvgr_hip.so (include/dvgr_store.h).

`VideoQAFlatLoader` yields the batches of the reference's `VideoQADataLoader` (DataLoader.py:90-168) -- same keyword
arguments, same tuple order, same `vocab` / `glove_matrix` / `batch_size` attributes, `len()` = number of batches -- from
the same questions pickle and vocab json, but reads the features from flat files (`write_flat`, tools/h5_to_flat.py) mapped
once instead of opening two `.h5` files per sample, and hands over DEVICE tensors: a native worker thread gathers the rows of
upcoming batches into pinned memory (`prefetch` batches ahead, no interpreter lock involved) and each batch costs one
host-to-device copy per feature kind on a copy stream, overlapped with the consumer's step.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import pickle
import struct

import numpy as np
import torch

from . import _lib as L_

MAGIC = b"DVGRFEAT"
QUESTION_CATEGORY = {'count': 0, 'exist': 1, 'query_color': 2, 'query_size': 3, 'query_actiontype': 4, 'query_actiondir': 5,
                     'query_shape': 6, 'greater_than': 7, 'equal_to': 8, 'less_than': 9, 'equal_color': 10, 'equal_size': 11,
                     'equal_actiontype': 12, 'equal_actiondir': 13, 'equal_shape': 14}      # DataLoader.py:29-30


class StoreInfo(C.Structure):
    _fields_ = [("items", C.c_int64), ("row_elems", C.c_int64), ("ndim", C.c_int32), ("elem_type", C.c_int32),
                ("shape", C.c_int64 * 4)]


STORE_SYMBOLS = [
    ("dvgr_store_open", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("dvgr_store_close", None, [C.c_void_p]),
    ("dvgr_store_get_info", C.c_int, [C.c_void_p, C.POINTER(StoreInfo)]),
    ("dvgr_store_ids", C.POINTER(C.c_int64), [C.c_void_p]),
    ("dvgr_store_lookup", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("dvgr_store_gather_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    ("dvgr_store_gather_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    ("dvgr_prefetch_start", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    ("dvgr_prefetch_next", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]),
    ("dvgr_prefetch_stop", None, [C.c_void_p]),
]
_bound = False


def load():
    global _bound
    lib = L_.load()
    if not _bound:
        for name, restype, argtypes in STORE_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _bound = True
    return lib


def write_flat(path, ids, array) -> None:
    """One flat feature file: `ids` (V,) ints, `array` (V, ...) float32 with 1..4 item dims (include/dvgr_store.h)."""
    array = np.ascontiguousarray(array, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    item = array.shape[1:]
    if not (1 <= len(item) <= 4) or ids.shape != (array.shape[0],):
        raise ValueError("array must be (V, d1[, d2[, d3[, d4]]]) and ids (V,)")
    shape = list(item) + [1] * (4 - len(item))
    ids_off = 80
    data_off = (ids_off + 8 * len(ids) + 4095) // 4096 * 4096
    with open(path, "wb") as f:
        f.write(struct.pack("<8sIIII4QQQQ", MAGIC, 1, 0, len(item), 0, *shape, len(ids), ids_off, data_off))
        f.write(ids.tobytes())
        f.write(b"\0" * (data_off - ids_off - 8 * len(ids)))
        f.write(array.tobytes())


class FeatureStore:
    """A mapped flat feature file: id -> row lookup and row gathers into host or device memory."""

    def __init__(self, path: str):
        self._lib = load()
        h = C.c_void_p()
        L_.check(self._lib.dvgr_store_open(str(path).encode(), C.byref(h)), "dvgr_store_open")
        self._h = h
        info = StoreInfo()
        L_.check(self._lib.dvgr_store_get_info(self._h, C.byref(info)), "dvgr_store_get_info")
        self.items = int(info.items)
        self.item_shape = tuple(int(info.shape[i]) for i in range(info.ndim))
        self.row_elems = int(info.row_elems)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dvgr_store_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ids(self) -> np.ndarray:
        p = self._lib.dvgr_store_ids(self._h)
        return np.ctypeslib.as_array(p, shape=(self.items,)).copy()

    def lookup(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        rows = np.empty_like(ids)
        rc = self._lib.dvgr_store_lookup(self._h, ids.ctypes.data, ids.size, rows.ctypes.data)
        if rc == -22:
            raise KeyError(f"video ids not in the store: {ids[rows < 0][:8].tolist()}")
        L_.check(rc, "dvgr_store_lookup")
        return rows

    def gather_host(self, rows, threads: int = 0) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((rows.size,) + self.item_shape, dtype=np.float32)
        L_.check(self._lib.dvgr_store_gather_host(self._h, rows.ctypes.data, rows.size, out.ctypes.data, threads),
                 "dvgr_store_gather_host")
        return out

    def gather_device(self, rows, out: torch.Tensor, stream=None, threads: int = 0) -> torch.Tensor:
        """rows -> `out` ((n,) + item_shape, float32, contiguous, on the GPU), enqueued on `stream` (default: current)."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        if not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != rows.size * self.row_elems:
            raise RuntimeError("gather_device needs a contiguous float32 device tensor of n * row_elems elements")
        st = stream if stream is not None else torch.cuda.current_stream(out.device)
        L_.check(self._lib.dvgr_store_gather_device(self._h, rows.ctypes.data, rows.size, out.data_ptr(), threads, st.cuda_stream),
                 "dvgr_store_gather_device")
        return out


def load_vocab(path):
    """DataLoader.py:36-42."""
    with open(path, "r") as f:
        vocab = json.load(f)
    for k in ("question", "answer", "question_answer"):
        if k + "_token_to_idx" in vocab:
            vocab[k + "_idx_to_token"] = {v: t for t, v in vocab[k + "_token_to_idx"].items()}
    return vocab


class VideoQAFlatLoader:
    """Batches of the reference's VideoQADataLoader from flat feature files, on the device, prefetched.

    Keyword arguments as there: vocab_json, question_pt, appearance_feat, motion_feat, batch_size, shuffle,
    train_num / val_num / test_num (subset); plus device, prefetch, drop_last, seed.  num_workers / pin_memory are
    accepted and ignored.  Each batch is the tuple of DataLoader.py:77-84 after default collation:
    (video_idx, question_idx, [question_category,] answer, appearance_feat, motion_feat, question, question_len)."""

    def __init__(self, **kwargs):
        self.vocab = load_vocab(str(kwargs.pop("vocab_json")))
        with open(str(kwargs.pop("question_pt")), "rb") as f:
            obj = pickle.load(f)
        cat = obj.get("question_category")
        fields = {k: obj[k] for k in ("questions", "questions_len", "video_ids", "question_id", "answers")}
        self.glove_matrix = obj.get("glove")
        for key in ("train_num", "val_num", "test_num"):
            n = kwargs.pop(key, 0)
            if n and n > 0:
                fields = {k: v[:n] for k, v in fields.items()}
                cat = cat[:n] if cat is not None else None
        self.questions = torch.from_numpy(np.asarray(fields["questions"]).astype(np.int64))
        self.questions_len = torch.from_numpy(np.asarray(fields["questions_len"]).astype(np.int64))
        self.video_ids = np.asarray(fields["video_ids"]).astype(np.int64)
        self.q_ids = np.asarray(fields["question_id"])
        self.answers = None if fields["answers"] is None else np.asarray(fields["answers"]).astype(np.int64)
        if cat is None:
            self.category = None
        else:      # the reader maps names (DataLoader.py:65); the SVQA writer already stores ints (svqa.py:95): accept both
            self.category = np.asarray([QUESTION_CATEGORY[c] if isinstance(c, str) else int(c) for c in cat], dtype=np.int64)
        self.app = FeatureStore(kwargs.pop("appearance_feat"))
        self.mot = FeatureStore(kwargs.pop("motion_feat"))
        self.app_rows = self.app.lookup(self.video_ids)          # DataLoader.py:70-71, once instead of per item
        self.mot_rows = self.mot.lookup(self.video_ids)
        self.batch_size = int(kwargs.pop("batch_size"))
        self.shuffle = bool(kwargs.pop("shuffle", False))
        self.drop_last = bool(kwargs.pop("drop_last", False))
        self.device = torch.device(kwargs.pop("device", "cuda"))
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.prefetch = max(1, int(kwargs.pop("prefetch", 2)))
        self.seed = kwargs.pop("seed", None)
        self.threads = int(kwargs.pop("threads", 0))
        kwargs.pop("num_workers", None)
        kwargs.pop("pin_memory", None)
        if kwargs:
            raise TypeError(f"unexpected arguments: {sorted(kwargs)}")
        self._epoch = 0

    def __len__(self):
        n = len(self.questions)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def batch_indices(self):
        n = len(self.questions)
        if self.shuffle:
            g = torch.Generator()
            if self.seed is not None:
                g.manual_seed(int(self.seed) + self._epoch)
            else:
                g.manual_seed(int(torch.randint(0, 2 ** 31, (1,)).item()))
            order = torch.randperm(n, generator=g).numpy()
        else:
            order = np.arange(n)
        self._epoch += 1
        return [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)][:len(self)]

    def _to_device_tables(self):
        """The small per-question tables live on the device; a batch of them is an index_select."""
        if getattr(self, "_tables", None) is None:
            dev = self.device
            self._tables = {"q": self.questions.to(dev), "ql": self.questions_len.to(dev),
                            "vid": torch.from_numpy(self.video_ids).to(dev),
                            "ans": None if self.answers is None else torch.from_numpy(self.answers).to(dev),
                            "cat": None if self.category is None else torch.from_numpy(self.category).to(dev)}
        return self._tables

    def __iter__(self):
        """One epoch.  A native worker thread (dvgr_prefetch_*) stages the feature rows of upcoming batches in pinned memory;
        this generator keeps one batch's host-to-device copies in flight on a copy stream while the consumer computes on
        the previous one."""
        lib = load()
        dev = self.device
        torch.cuda.set_device(dev)
        batches = self.batch_indices()
        if not batches:
            return
        order = np.ascontiguousarray(np.concatenate(batches), dtype=np.int64)
        rows = [np.ascontiguousarray(self.app_rows[order]), np.ascontiguousarray(self.mot_rows[order])]
        stores = (C.c_void_p * 2)(self.app._h, self.mot._h)
        rowp = (C.c_void_p * 2)(rows[0].ctypes.data, rows[1].ctypes.data)
        h = C.c_void_p()
        L_.check(lib.dvgr_prefetch_start(stores, 2, rowp, order.size, self.batch_size, self.prefetch + 1, self.threads, C.byref(h)),
                 "dvgr_prefetch_start")
        tabs = self._to_device_tables()
        copy_stream = L_.streams(dev)["copy"]
        order_dev = torch.from_numpy(order).to(dev)

        def issue(k):
            idx = batches[k]
            B = len(idx)
            with torch.cuda.stream(copy_stream):
                app = torch.empty((B,) + self.app.item_shape, device=dev, dtype=torch.float32)
                mot = torch.empty((B,) + self.mot.item_shape, device=dev, dtype=torch.float32)
                dst = (C.c_void_p * 2)(app.data_ptr(), mot.data_ptr())
                n = C.c_int64()
                rc = lib.dvgr_prefetch_next(h, dst, C.byref(n), copy_stream.cuda_stream)
                L_.check(rc, "dvgr_prefetch_next")
                if n.value != B:
                    raise RuntimeError("prefetcher and loader disagree on the batch size")
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return idx, app, mot, ev

        try:
            pending = issue(0)
            pos = 0
            for k in range(len(batches)):
                idx, app, mot, ev = pending
                pending = issue(k + 1) if k + 1 < len(batches) else None
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                app.record_stream(cur)
                mot.record_stream(cur)
                sel = order_dev[pos:pos + len(idx)]
                pos += len(idx)
                qid = self.q_ids[idx]
                qid = torch.from_numpy(qid.astype(np.int64)) if np.issubdtype(qid.dtype, np.integer) else list(qid)
                ans = None if tabs["ans"] is None else tabs["ans"][sel]
                head = (tabs["vid"][sel], qid) if tabs["cat"] is None else (tabs["vid"][sel], qid, tabs["cat"][sel])
                yield head + (ans, app, mot, tabs["q"][sel], tabs["ql"][sel])
        finally:
            try:                               # copies still in flight read the pinned slots stop() frees
                copy_stream.synchronize()
            except Exception:                  # interpreter shutdown: torch may already be torn down
                pass
            lib.dvgr_prefetch_stop(h)
