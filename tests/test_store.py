"""CPU: the feature store (row f4, include/dvgr_store.h) -- flat file format, id index, threaded row gather -- against the
numpy restatement of the reference's per-item reads (oracle/store_numpy.py).  Byte-exact: this is a copy path."""
import ctypes as C
import os

import numpy as np
import pytest

from dualvgr_videoqa_amd import store as ST
from oracle import store_numpy as SN
from store_helpers import make_dataset


def test_flat_file_roundtrip_lookup_and_gather(tmp_path):
    obj, (app, ids), (mot, mot_ids), paths = make_dataset(str(tmp_path))
    s = ST.FeatureStore(paths["app.dvgrf"])
    assert s.items == len(ids) and s.item_shape == app.shape[1:]
    assert np.array_equal(s.ids, ids)
    table = SN.id_to_index(ids)
    want_rows = np.array([table[str(int(v))] for v in obj["video_ids"]])
    assert np.array_equal(s.lookup(obj["video_ids"]), want_rows)
    with pytest.raises(KeyError):
        s.lookup([int(ids[0]), 5])
    rows = np.array([3, 3, 0, len(ids) - 1, 1, 0, 7], dtype=np.int64)       # duplicates, unordered
    for threads in (0, 1, 3, 8):                                             # 3 threads: rows split between threads
        assert np.array_equal(s.gather_host(rows, threads), app[rows])
    assert s.gather_host(np.zeros(0, dtype=np.int64)).shape == (0,) + app.shape[1:]
    lib = ST.load()
    bad = np.array([len(ids)], dtype=np.int64)
    buf = np.empty(app.shape[1:], dtype=np.float32)
    assert lib.dvgr_store_gather_host(s._h, bad.ctypes.data, 1, buf.ctypes.data, 1) == -1
    s.close()
    m = ST.FeatureStore(paths["mot.dvgrf"])
    assert np.array_equal(m.gather_host(m.lookup(obj["video_ids"])), mot[[SN.id_to_index(mot_ids)[str(int(v))] for v in obj["video_ids"]]])


def test_duplicate_ids_resolve_like_the_dict_comprehension(tmp_path):
    ids = np.array([5, 9, 5, 2], dtype=np.int64)
    arr = np.arange(4 * 6, dtype=np.float32).reshape(4, 6)
    p = str(tmp_path / "d.dvgrf")
    ST.write_flat(p, ids, arr)
    s = ST.FeatureStore(p)
    assert s.lookup([5, 2, 9]).tolist() == [SN.id_to_index(ids)[k] for k in ("5", "2", "9")] == [2, 3, 1]


def test_large_rows_are_split_across_threads_bytewise(tmp_path):
    rng = np.random.default_rng(0)
    arr = rng.standard_normal((5, 3, 1000, 130)).astype(np.float32)          # 1.56 MB rows: the 8-thread path
    p = str(tmp_path / "big.dvgrf")
    ST.write_flat(p, np.arange(5), arr)
    s = ST.FeatureStore(p)
    rows = np.array([4, 0, 2, 2, 1, 3, 0], dtype=np.int64)
    got = s.gather_host(rows)
    assert got.tobytes() == arr[rows].tobytes()


def test_malformed_files_are_refused(tmp_path):
    lib = ST.load()
    h = C.c_void_p()
    assert lib.dvgr_store_open(str(tmp_path / "missing").encode(), C.byref(h)) == -20
    p = tmp_path / "short"
    p.write_bytes(b"DVGRFEAT")
    assert lib.dvgr_store_open(str(p).encode(), C.byref(h)) == -21
    good = str(tmp_path / "g.dvgrf")
    ST.write_flat(good, np.arange(3), np.zeros((3, 4), dtype=np.float32))
    raw = bytearray(open(good, "rb").read())
    for mutate in (lambda b: b.__setitem__(slice(0, 8), b"NOTMAGIC"), lambda b: b.__setitem__(8, 2),
                   lambda b: b.__setitem__(56, 200)):                        # magic, version, item count beyond the file
        b = bytearray(raw)
        mutate(b)
        q = tmp_path / "bad"
        q.write_bytes(bytes(b))
        assert lib.dvgr_store_open(str(q).encode(), C.byref(h)) == -21
    q = tmp_path / "trunc"
    q.write_bytes(bytes(raw[:-8]))
    assert lib.dvgr_store_open(str(q).encode(), C.byref(h)) == -21
    assert lib.dvgr_store_open(None, C.byref(h)) == -2


def test_oracle_batches_follow_the_reference_tuple_order(tmp_path):
    obj, (app, ids), (mot, mot_ids), _ = make_dataset(str(tmp_path))
    b = next(SN.batches(obj, app, ids, mot, mot_ids, batch_size=4))
    assert len(b) == 8 and b[4].shape == (4,) + app.shape[1:] and b[5].shape == (4,) + mot.shape[1:]
    assert b[2].tolist() == [0, 1, 2, 3] and b[6].dtype == np.int64
    obj.pop("question_category")
    b = next(SN.batches(obj, app, ids, mot, mot_ids, batch_size=4))
    assert len(b) == 7 and b[3].shape == (4,) + app.shape[1:]
