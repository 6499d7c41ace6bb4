"""ctypes binding of libdvgr_hip.so (C ABI declared in include/dvgr.h).

There is NO fallback: if the HIP library is missing this module raises, and so does every
product entry point that needs it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVGR_LIB") or os.path.join(_HERE, "libdvgr_hip.so")   # DVGR_LIB: debug builds only

DVGR_F32, DVGR_BF16, DVGR_BF16X3 = 0, 1, 2

# enum dvgr_param_kind
(P_QA_W1, P_QA_B1, P_QA_W2, P_QA_B2, P_WQ, P_BQ, P_GAT_W, P_GAT_B, P_GAT_A, P_GAT_C,
 P_SF_P1, P_SF_BP, P_SF_P2, P_MFB_W0, P_MFB_B0, P_MFB_W1, P_MFB_B1, P_MFB_WO, P_MFB_BO, P_KINDS) = range(20)
# enum dvgr_saved_kind
(S_QA_D, S_QA_NRM, S_QA_A, S_QA_AH, S_QA_S, S_QA_Q, S_P, S_S, S_X, S_WH, S_U, S_V, S_TN, S_BETA,
 S_MFB_X0, S_MFB_X1, S_MFB_Z, S_XD, S_KINDS) = range(19)

ERRORS = {-1: "DVGR_E_DIMS (unsupported dimensions)", -2: "DVGR_E_NULL (null pointer)",
          -3: "DVGR_E_WORKSPACE (saved/workspace buffer too small)", -4: "DVGR_E_ALIGN (pointer not 16-byte aligned)",
          -5: "DVGR_E_KIND (unknown tensor kind)",
          -6: "DVGR_E_ORDER (rounds 2-3; no longer returned: dvgr_comm_place puts the communication stream on a free pipe)"}


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "T", "D", "Wd", "L", "G", "dtype")]


class IO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("A", "M", "dq", "words", "qlen", "vis", "aq_embed", "mq_embed",
                                          "com_app", "com_mot", "aq_fusion", "mq_fusion")]


class Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g_vis", "g_aq_embed", "g_mq_embed", "g_com_app", "g_com_mot",
                                          "g_aq_fusion", "g_mq_fusion", "gA", "gM", "gdq", "gwords", "gparams")]


class LossSpec(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("dep_scale", C.c_float), ("loss3", C.c_void_p),
                ("g_vis", C.c_void_p), ("g_com_app", C.c_void_p), ("g_com_mot", C.c_void_p), ("g_aq_fusion", C.c_void_p),
                ("g_mq_fusion", C.c_void_p)]


class Opts(C.Structure):
    _fields_ = [("train", C.c_int32), ("probe_site", C.c_int32), ("probe_layer", C.c_int32), ("flags", C.c_int32),
                ("seed", C.c_uint64), ("probe_begin", C.c_void_p), ("probe_end", C.c_void_p),
                ("layer_done_events", C.POINTER(C.c_void_p)), ("loss", C.POINTER(LossSpec)),
                ("layer_done_cb", C.c_void_p), ("layer_done_user", C.c_void_p), ("comm", C.c_void_p), ("p1t", C.c_void_p), ("lo8", C.c_void_p)]


LAYER_DONE_CB = C.CFUNCTYPE(None, C.c_int32, C.c_void_p)


OPT_SINGLE_STREAM = 1
OPT_NO_QUERY_STREAM = 2
OPT_NO_FUSED_GAT = 4
OPT_LEAVES_ON_QUERY = 8
OPT_NO_FUSED_GAT_BWD = 16
OPT_FORCE_FUSED_GAT_BWD = 32
OPT_FUSED_BWD_READS_DZ = 64
OPT_S5_READS_TN = 128
OPT_NO_LO8 = 256
OPT_KB32 = 512
OPT_NO_HALF_GAT_TILES = 1024
OPT_NO_FOLDED_PUNISH = 2048
OPT_S0_ON_MAIN = 4096
OPT_S0_ON_SIDE = 8192
OPT_DP_ON_MAIN = 16384
OPT_FORKS_BY_RECORD = 32768

# enum dvgr_probe_site
(PROBE_NONE, PROBE_FWD_QA_GEMM, PROBE_FWD_GAT_GEMM, PROBE_FWD_GAT_ATTN, PROBE_FWD_SF_GEMM, PROBE_FWD_MFB_GEMM,
 PROBE_BWD_SF_DW, PROBE_BWD_SF_DX, PROBE_BWD_GAT_ATTN, PROBE_BWD_GAT_DW, PROBE_BWD_GAT_DX, PROBE_SITES) = range(12)

# every symbol include/dvgr.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("dvgr_version", C.c_char_p, []),
    ("dvgr_streams_init", C.c_int, []),
    ("dvgr_streams_place", C.c_int, [C.c_void_p]),
    ("dvgr_stream_create_placed", C.c_int, [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int]),
    ("dvgr_check_dims", C.c_int, [C.POINTER(Dims)]),
    ("dvgr_param_count", C.c_int64, [C.POINTER(Dims)]),
    ("dvgr_param_offset", C.c_int, [C.POINTER(Dims), C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("dvgr_param_layer_range", C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("dvgr_saved_bytes", C.c_size_t, [C.POINTER(Dims)]),
    ("dvgr_saved_offset", C.c_int, [C.POINTER(Dims), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    ("dvgr_workspace_bytes", C.c_size_t, [C.POINTER(Dims)]),
    ("dvgr_pack_weights", C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("dvgr_p1t_elems", C.c_size_t, [C.POINTER(Dims)]),
    ("dvgr_backward_uses_p1t", C.c_int, [C.POINTER(Dims), C.c_int32]),
    ("dvgr_lo8_bytes", C.c_size_t, [C.POINTER(Dims)]),
    ("dvgr_pack_lo8", C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("dvgr_pack_p1t", C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("dvgr_unit_forward", C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(IO), C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.POINTER(Opts), C.c_void_p]),
    ("dvgr_unit_backward", C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.POINTER(IO), C.POINTER(Grads),
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Opts), C.c_void_p]),
    ("dvgr_scope_u_loss", C.c_int, [C.POINTER(Dims), C.POINTER(IO), C.c_float, C.c_float, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
]

# test hooks declared in include/dvgr_debug.h
SYMBOLS += [
    ("dvgr_dbg_gemm", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    ("dvgr_dbg_split", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    ("dvgr_dbg_gemm_planes", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p]),
    ("dvgr_dbg_colsum", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    ("dvgr_dbg_keep_mask", C.c_int, [C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p]),
    ("dvgr_dbg_site_key", C.c_uint32, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("dvgr_dbg_event_create", C.c_int, [C.POINTER(C.c_void_p)]),
    ("dvgr_dbg_event_elapsed_ms", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    ("dvgr_dbg_gemm_planes_batch", C.c_int, [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                                 C.c_int64, C.c_int64, C.c_void_p, C.c_int64] + [C.c_int] * 6 + [C.c_void_p, C.c_int, C.c_void_p]),
    ("dvgr_dbg_event_destroy", C.c_int, [C.c_void_p]),
    ("dvgr_dbg_launch_count", C.c_longlong, []),
    ("dvgr_dbg_xmap_tags", C.c_int, [C.c_int]),
    ("dvgr_dbg_plan_xcd_blocks", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
]

_lib = None


def load():
    """Load libdvgr_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C dualvgr-videoqa_amd/csrc).  There is no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class ProbeEvent:
    """A timing hipEvent without the system-scope fence at its record (dvgr_dbg_event_create); quacks like the
    torch.cuda.Event the probe hooks take (`cuda_event`, `elapsed_time`)."""

    def __init__(self):
        h = C.c_void_p()
        check(load().dvgr_dbg_event_create(C.byref(h)), "dvgr_dbg_event_create")
        self.cuda_event = h.value

    def record(self):                 # recorded by the library around the probed launch
        return None

    def elapsed_time(self, end: "ProbeEvent") -> float:
        ms = C.c_float()
        check(load().dvgr_dbg_event_elapsed_ms(self.cuda_event, end.cuda_event, C.byref(ms)), "dvgr_dbg_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.cuda_event:
                load().dvgr_dbg_event_destroy(self.cuda_event)
        except Exception:
            pass


_streams = {}
STREAM_ORDER = os.environ.get("DVGR_STREAM_ORDER", "side,encoder,copy").split(",")
STREAM_RANK = {"encoder": 2, "copy": 9}      # importance when a later stream has to share a pipe (csrc/dvgr_side.h: kRank*)
PLACEMENT = {}                                # device index -> {stream name: placement report} (bench.py prints it)


def _report(rep):
    nc, nb, best, mask = int(rep[0]), int(rep[1]), int(rep[2]), int(rep[3])
    return {"candidates": nc, "measured_against": nb, "chosen": best, "collisions_of_chosen": mask,
            "undisturbed_chain_us": [round(float(rep[4 + i]), 1) for i in range(nb)],
            "chain_us_per_candidate": [[round(float(rep[4 + nb + k * nb + i]), 1) for i in range(nb)] for k in range(nc)]}


def streams(dev):
    """Every stream this package uses on `dev` besides the caller's, created ONCE, whichever component asks first:
    {'encoder': stream of DualVGR's question encoder, 'copy': stream of the loader's host-to-device copies} as
    torch.cuda.ExternalStream; the library's internal side / query streams are created in the same pass.

    Each is put on the hardware queue that disturbs the current stream and the streams placed before it least
    (dvgr_streams_place / dvgr_stream_create_placed; csrc/dvgr_place.hip): two busy hardware queues on one of the command
    processor's four pipes cost every dispatch of both ~25 us, and which queues share a pipe follows creation order across
    the whole process (torch's pools, the process group, RCCL) -- with the loader's copy stream created before the model's
    streams the whole-model step cost 10.7 ms instead of 5.3 ms, with a torch.distributed group initialised first 8.7 ms
    (profiles/HISTORY.md section 5)."""
    import torch
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = dev.index
    if key not in _streams:
        made, reports = {}, {}
        lib = load()
        with torch.cuda.device(dev):
            caller = torch.cuda.current_stream(dev).cuda_stream
            for name in STREAM_ORDER:
                if name == "side":
                    check(lib.dvgr_streams_place(caller), "dvgr_streams_place")
                elif name == "comm":     # (rounds 2-3 probe: the communication stream at a chosen place in the creation order)
                    h = C.c_void_p()
                    check(lib.dvgr_comm_device_stream(C.byref(h)), "dvgr_comm_device_stream")
                else:
                    h, rep = C.c_void_p(), (C.c_float * 64)()
                    check(lib.dvgr_stream_create_placed(0, STREAM_RANK.get(name, 8), caller, C.byref(h), rep, 64),
                          "dvgr_stream_create_placed")
                    made[name] = torch.cuda.ExternalStream(h.value, device=dev)
                    reports[name] = _report(rep)
        _streams[key] = made
        PLACEMENT[key] = reports
    return _streams[key]


def check(code: int, what: str):
    if code == 0:
        return
    if code < 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(code, code)}")
    raise RuntimeError(f"{what} failed: hipError_t {code}")


def make_dims(B, N, T, D, Wd, L, G=1, dtype=DVGR_F32) -> Dims:
    return Dims(B, N, T, D, Wd, L, G, dtype)


def param_offset(dims: Dims, kind: int, layer: int = 0, mod: int = 0):
    off, rows, cols = C.c_int64(), C.c_int64(), C.c_int64()
    check(load().dvgr_param_offset(C.byref(dims), kind, layer, mod, C.byref(off), C.byref(rows), C.byref(cols)),
          "dvgr_param_offset")
    return off.value, rows.value, cols.value


def saved_offset(dims: Dims, kind: int, layer: int = 0, mod: int = 0) -> int:
    off = C.c_size_t()
    check(load().dvgr_saved_offset(C.byref(dims), kind, layer, mod, C.byref(off)), "dvgr_saved_offset")
    return off.value
