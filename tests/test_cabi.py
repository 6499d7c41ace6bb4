"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, and its host-side
logic (layouts, argument checks, the drop-in module's parameter surface) is consistent.
No compute call is made here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L
from dualvgr_videoqa_amd import dropout
from oracle import unit_torch as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):          # every header under include/
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(dvgr_[a-z_0-9]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    lib = L.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    from dualvgr_videoqa_amd import head as HD
    from dualvgr_videoqa_amd import clip as CL
    from dualvgr_videoqa_amd import store as ST
    from dualvgr_videoqa_amd import trainer as TR
    from dualvgr_videoqa_amd import inputs as IN
    from dualvgr_videoqa_amd import parallel as PA
    IN.load()
    HD.load()
    CL.load()
    ST.load()
    TR.load()
    bound = ({n for n, _, _ in L.SYMBOLS} | {n for n, _, _ in HD.HEAD_SYMBOLS} | {n for n, _, _ in CL.CLIP_SYMBOLS}
             | {n for n, _, _ in ST.STORE_SYMBOLS} | {n for n, _, _ in TR.OPTIM_SYMBOLS} | {n for n, _, _ in IN.INPUT_SYMBOLS})
    PA._comm_lib()                                # resolves RCCL by dlopen (no GPU call); binds the dvgr_comm_* entry points
    bound |= {n for n, _, _ in PA.COMM_SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.dvgr_version().startswith(b"dvgr-hip")


def test_optimizer_refuses_bad_segments_without_touching_the_gpu():
    from dualvgr_videoqa_amd import trainer as TR
    lib = TR.load()
    assert C.sizeof(TR.OptimSeg) == 7 * 8
    segs = (TR.OptimSeg * 1)()
    buf = (C.c_float * 8)()
    assert lib.dvgr_optim_grad_sqnorm(segs, 1, None, None) == -2                 # no output
    segs[0].n = 4
    assert lib.dvgr_optim_adam_step(segs, 1, None, 12.0, 1e-4, 0.9, 0.999, 1e-8, 1, None) == -2     # NULL gradient
    segs[0].grad = C.addressof(buf)
    assert lib.dvgr_optim_adam_step(segs, 1, None, 12.0, 1e-4, 0.9, 0.999, 1e-8, 1, None) == -2     # NULL param / moments
    segs[0].param = segs[0].exp_avg = segs[0].exp_avg_sq = C.addressof(buf)
    assert lib.dvgr_optim_adam_step(segs, 1, None, 12.0, 1e-4, 0.9, 0.999, 1e-8, 0, None) == -1     # step must be >= 1
    assert lib.dvgr_optim_adam_step(segs, 1, None, 12.0, 1e-4, 1.0, 0.999, 1e-8, 1, None) == -1     # beta1 < 1
    assert lib.dvgr_optim_adam_step(segs, -1, None, 12.0, 1e-4, 0.9, 0.999, 1e-8, 1, None) == -1


def test_input_operators_refuse_bad_arguments_without_touching_the_gpu():
    """include/dvgr_inputs.h: argument checks come before any launch."""
    from dualvgr_videoqa_amd import inputs as IN
    lib = IN.load()
    buf = (C.c_float * 64)()
    a = C.addressof(buf)
    a16 = (a + 15) // 16 * 16
    assert lib.dvgr_words_forward(None, 10, 8, a16, 2, 0, 0, a16, None, None) == -2
    assert lib.dvgr_words_forward(a16, 10, 6, a16, 2, 0, 0, a16, None, None) == -1          # wd % 4
    assert lib.dvgr_words_forward(a16, 0, 8, a16, 2, 0, 0, a16, None, None) == -1           # empty table
    assert lib.dvgr_words_forward(a16, 10, 8, a16, 2, 0, 0, a16 + 4, None, None) == -4      # misaligned output
    assert lib.dvgr_words_forward(a16, 10, 8, a16, 0, 0, 0, a16, None, None) == 0           # nothing to do
    assert lib.dvgr_words_backward(a16, a16, a16, 2, 10, 8, 0, 0, None, None) == -2
    assert lib.dvgr_linear_forward(a16, a16, None, None, 4, 8, 8, 0, None) == -2
    assert lib.dvgr_linear_forward(a16, a16, None, a16, 4, 6, 8, 0, None) == -1             # N % 4
    assert lib.dvgr_linear_forward(a16, a16, None, a16, 4, 8, 8, 7, None) == -1             # unknown dtype
    assert lib.dvgr_linear_backward(a16, a16, a16, None, None, None, 4, 8, 8, 0, None) == -2


def test_struct_sizes_match_header():
    assert C.sizeof(L.Dims) == 8 * 4
    assert C.sizeof(L.IO) == 12 * 8
    assert C.sizeof(L.Grads) == 12 * 8


@pytest.mark.parametrize("bad", [dict(G=5), dict(G=0), dict(D=100), dict(Wd=301), dict(N=65), dict(T=257), dict(B=0), dict(dtype=7)])
def test_check_dims_rejects(bad):
    lib = L.load()
    kw = dict(B=4, N=20, T=15, D=768, Wd=300, L=3, G=1, dtype=0)
    assert lib.dvgr_check_dims(C.byref(L.make_dims(**kw))) == 0
    kw.update(bad)
    assert lib.dvgr_check_dims(C.byref(L.make_dims(**kw))) == -1


def test_null_and_small_buffers_are_refused_without_touching_the_gpu():
    lib = L.load()
    d = L.make_dims(2, 5, 6, 64, 32, 2)
    io = L.IO()
    assert lib.dvgr_unit_forward(C.byref(d), None, None, C.byref(io), None, 0, None, 0, None, None) == -2
    buf = (C.c_char * 64)()
    io = L.IO(*([C.addressof(buf) // 16 * 16 + 16] * 12))
    p = C.addressof(buf) // 16 * 16 + 16
    assert lib.dvgr_unit_forward(C.byref(d), p, None, C.byref(io), p, 16, p, 16, None, None) == -3


@pytest.mark.parametrize("D,Wd,Lr,G", [(64, 32, 2, 1), (768, 300, 1, 1), (768, 300, 3, 1), (64, 32, 2, 2), (64, 32, 3, 3)])
def test_param_layout_is_disjoint_and_aligned(D, Wd, Lr, G):
    d = L.make_dims(1, 8, 1, D, Wd, Lr, G)
    total = L.load().dvgr_param_count(C.byref(d))
    tab = dv.param_table(d)
    shapes = O.param_shapes(Wd, D, Lr, G)
    assert {n for n, _, _ in tab} == set(shapes)
    used = np.zeros(total, dtype=np.int8)
    for name, shape, off in tab:
        assert tuple(shape) == tuple(shapes[name]), name
        n = int(np.prod(shape))
        assert off >= 0 and off + n <= total
        used[off:off + n] += 1
    assert used.max() == 1, "overlapping parameters"
    for kind in range(L.P_KINDS):
        off, rows, cols = L.param_offset(d, kind, 0, 0)
        assert off % 64 == 0
    # the blocks tile the buffer: layer blocks 0 .. L-1, the MFB block (L), the query block (L + 1)
    lib = L.load()
    prev = 0
    for l in range(Lr + 2):
        a, b = C.c_int64(), C.c_int64()
        assert lib.dvgr_param_layer_range(C.byref(d), l, C.byref(a), C.byref(b)) == 0
        assert a.value == prev and b.value > a.value
        prev = b.value
    # block L + 2: the GAT modules L .. L * G - 1 of graph_layers > 1 (empty at G = 1)
    a, b = C.c_int64(), C.c_int64()
    assert lib.dvgr_param_layer_range(C.byref(d), Lr + 2, C.byref(a), C.byref(b)) == 0
    assert a.value == prev and b.value == total and (b.value > a.value) == (G > 1)
    assert lib.dvgr_param_layer_range(C.byref(d), Lr + 3, C.byref(a), C.byref(b)) == -5
    # which parameter lives in which block
    rng = {}
    for l in range(Lr + 3):
        assert lib.dvgr_param_layer_range(C.byref(d), l, C.byref(a), C.byref(b)) == 0
        rng[l] = (a.value, b.value)
    for name, shape, off in tab:
        blk = next(l for l, (x, y) in rng.items() if x <= off < y)
        if name.startswith(("queryAttn.", "queryPunish_")):
            assert blk == Lr + 1, name
        elif name.startswith("visualfusion."):
            assert blk == Lr, name
        else:
            idx = int(name.split(".")[1])                  # GAT module index (or SF-attention layer): modules >= L: extra block
            assert blk == (idx if idx < Lr else Lr + 2), name
    n_real = sum(int(np.prod(s)) for s in shapes.values())
    assert int(used.sum()) == n_real


def test_saved_and_workspace_sizes():
    lib = L.load()
    d = L.make_dims(64, 20, 15, 768, 300, 3)
    sb, wb = lib.dvgr_saved_bytes(C.byref(d)), lib.dvgr_workspace_bytes(C.byref(d))
    assert 50e6 < sb < 800e6 and 10e6 < wb < 800e6
    offs = [L.saved_offset(d, k, 2, 1) for k in range(L.S_KINDS)]
    assert all(o % 256 == 0 and o < sb // 2 for o in offs)
    assert len(set(offs)) == len(offs)


def test_module_surface_matches_reference_state_dict():
    m = dv.DualVGRUnit_multiple(word_dim=32, module_dim=64, num_of_nodes=5, unit_layers=2)
    sd = m.state_dict()
    shapes = O.param_shapes(32, 64, 2)
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    # every parameter aliases the flat kernel-side buffer, also after dtype/device round trips
    new = {k: torch.randn(v.shape) for k, v in sd.items()}
    m.load_state_dict(new, strict=True)
    for name, shape, off in m._table:
        n = int(np.prod(shape))
        assert torch.equal(m._flat[off:off + n].view(shape), new[name]), name
    m = m.double().float()
    for name, shape, off in m._table:
        n = int(np.prod(shape))
        assert torch.equal(m._flat[off:off + n].view(shape), new[name]), name
    p = dict(m.named_parameters())["mcGCN.1.attention_3.a.bias"]
    with torch.no_grad():
        p.add_(1.0)                                  # an optimizer-style in-place update
    off = [o for n_, _, o in m._table if n_ == "mcGCN.1.attention_3.a.bias"][0]
    assert m._flat[off].item() == pytest.approx(new["mcGCN.1.attention_3.a.bias"].item() + 1.0)
    assert tuple(m.appearance_adj.shape) == (5, 5) and float(m.appearance_adj.min()) > 0


def test_module_refuses_cpu_tensors():
    m = dv.DualVGRUnit_multiple(word_dim=32, module_dim=64, num_of_nodes=5, unit_layers=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 5, 64), torch.zeros(2, 5, 64), torch.zeros(2, 4, 64), torch.zeros(2, 4, 32), torch.tensor([4, 2]))


def test_constructor_mirrors_reference_errors():
    with pytest.raises(ValueError):
        dv.DualVGRUnit_multiple(graph_module="GCN")        # reference builds no GNN lists then fails in forward
    with pytest.raises(NotImplementedError):               # the two modalities are batched: one graph_layers for both
        dv.DualVGRUnit_multiple(module_dim=64, word_dim=32, appearance_graph_layers=2, motion_graph_layers=1)
    m = dv.DualVGRUnit_multiple(module_dim=64, word_dim=32, appearance_graph_layers=2, motion_graph_layers=2, unit_layers=2)
    assert len(m.acGCN) == 4 and len(m.motion_GCN) == 4   # L * G modules per family, as the reference builds them


def test_dropout_replica_statistics_and_site_keys():
    lib = L.load()
    seed = 0x1234_5678_9ABC_DEF0
    keys = set()
    for l in range(3):
        for mod in range(2):
            for gat in range(2):
                for which in range(6):
                    k = dropout.site_key(seed, dropout.drop_site(l, mod, gat, which))
                    assert k == lib.dvgr_dbg_site_key(seed, l, mod, gat, which)
                    keys.add(k)
    assert len(keys) == 72
    m = dropout.keep_mask(dropout.site_key(seed, 3), 2_000_000)
    assert abs(m.mean() - 0.85) < 2e-3
    # lag-1 independence of neighbouring elements (they share one 32-bit hash)
    assert abs(np.mean(m[0::2] * m[1::2]) - 0.85 ** 2) < 2e-3
    masks = dropout.site_masks(seed, 2, 5, 64, 2)
    assert set(masks) == set(O.dropout_site_shapes(2, 5, 64, 2))


def test_head_layout_and_argument_checks():
    """Answer head (include/dvgr_head.h): dims checks, parameter table = the reference's state_dict surface, refusals."""
    from dualvgr_videoqa_amd import head as HD
    from oracle import head_torch as H
    lib = HD.load()
    ok = HD.HeadDims(4, 20, 768, 1000, 0)
    assert lib.dvgr_head_check_dims(C.byref(ok)) == 0
    for bad in (HD.HeadDims(0, 20, 768, 1000, 0), HD.HeadDims(4, 65, 768, 1000, 0), HD.HeadDims(4, 20, 100, 1000, 0),
                HD.HeadDims(4, 20, 768, 1001, 0), HD.HeadDims(4, 20, 768, 1000, 9)):
        assert lib.dvgr_head_check_dims(C.byref(bad)) == -1
    m = dv.AnswerHead(module_dim=64, num_answers=10)
    want = dict(H.param_shapes(64, 10))
    got = {k: tuple(v.shape) for k, v in m.named_parameters()}
    assert got == want
    offs = sorted(m._table)
    assert all(o % 64 == 0 for o in offs) and len(set(offs)) == len(offs)
    io = HD.HeadIO()
    assert lib.dvgr_head_forward(C.byref(ok), None, C.byref(io), None, 0, None, 0, 0, 0, None) == -2
    assert lib.dvgr_head_saved_bytes(C.byref(ok)) > 0 and lib.dvgr_head_workspace_bytes(C.byref(ok)) > 0


def test_clip_layout_and_argument_checks():
    """Appearance encoder (include/dvgr_clip.h): dims checks, parameter table = torch.nn.LSTM's state_dict surface, refusals."""
    from dualvgr_videoqa_amd import clip as CL
    from oracle import clip_torch as CT
    lib = CL.load()
    ok = CL.ClipDims(64 * 20, 16, 2048, 384, 0)
    assert lib.dvgr_clip_check_dims(C.byref(ok)) == 0
    for bad in (CL.ClipDims(0, 16, 2048, 384, 0), CL.ClipDims(8, 0, 2048, 384, 0), CL.ClipDims(8, 65, 2048, 384, 0),
                CL.ClipDims(8, 16, 2002, 384, 0), CL.ClipDims(8, 16, 2048, 100, 0), CL.ClipDims(8, 16, 2048, 384, 5),
                CL.ClipDims(1 << 20, 64, 2048, 384, 0)):
        assert lib.dvgr_clip_check_dims(C.byref(bad)) == -1
    m = dv.ClipEncoder(appearance_dim=64, module_dim=64)
    want = dict(CT.param_shapes(64, 64))
    got = {k: tuple(v.shape) for k, v in m.named_parameters()}
    assert got == want
    ref = torch.nn.LSTM(64, 32, bidirectional=True)
    assert list(m.encoder.state_dict().keys()) == list(ref.state_dict().keys())
    assert len(set(m.encoder._table)) == 8 and m.encoder._total == sum(p.numel() for p in ref.parameters())
    q = CL.QuestionEncoder(50, 300, 64)
    assert {k: tuple(v.shape) for k, v in q.state_dict().items()} == CT.question_param_shapes(50, 300, 64)
    io = CL.LstmIO()
    assert lib.dvgr_lstm_forward(C.byref(ok), None, C.byref(io), 0, None, 0, None, 0, 0, 0, None) == -2
    assert lib.dvgr_clip_forward(C.byref(ok), None, None, None, None, 0, None, 0, 0, 0, None) == -2
    assert lib.dvgr_clip_saved_bytes(C.byref(ok)) > 0 and lib.dvgr_clip_workspace_bytes(C.byref(ok)) > 0


def _plan(problems, bm=128, bn=128, max_per_xcd=32):
    lib = L.load()
    n = len(problems)
    M = (C.c_int * n)(*[p[0] for p in problems])
    N = (C.c_int * n)(*[p[1] for p in problems])
    K = (C.c_int * n)(*[p[2] for p in problems])
    seg = (C.c_int * 80)()
    gx = lib.dvgr_dbg_plan_xcd_blocks(n, C.addressof(M), C.addressof(N), C.addressof(K), bm, bn, max_per_xcd, C.addressof(seg))
    return gx, [[tuple(seg[(2 * x + k) * 5:(2 * x + k) * 5 + 5]) for k in range(2)] for x in range(8)]


@pytest.mark.parametrize("name,problems,bm,bn,cap", [
    ("C2 weight gradients: dP1 x 2 (K = 2 B N) + dWcat x 4 (K = B N)", [(768, 768, 2560)] * 2 + [(768, 768, 1280)] * 4, 128, 128, 32),
    ("C2 S4: 2 views x 2 modalities", [(1280, 768, 768)] * 4, 128, 128, 32),
    ("C2 layer-0 projection", [(1280, 1536, 768)] * 2, 128, 128, 32),
    ("C3 S4", [(1024, 768, 768)] * 4, 128, 128, 32),
    ("C4 S4 on half-height tiles", [(512, 768, 768)] * 4, 64, 128, 32),
    ("C5 S4 on 128 x 256 tiles", [(4096, 768, 768)] * 4, 128, 256, 96),
    ("eval-mode weight gradients", [(768, 768, 2560)] * 2 + [(1536, 768, 1280)] * 2, 128, 128, 32),
    ("a partial last row tile", [(1200, 768, 768)] * 4, 128, 128, 32),
])
def test_xcd_block_map_planner_covers_every_tile_once_and_balances_the_xcds(name, problems, bm, bn, cap):
    """csrc/dvgr_gemm.hip plan_xcd_blocks (host code; round 6): every output tile of every problem belongs to exactly one XCD's
    block, the blocks of a class are equal, every XCD gets one block per class (equal work), the grid covers the busiest XCD."""
    gx, seg = _plan(problems, bm, bn, cap)
    assert gx > 0, name
    seen = {}
    per_xcd = []
    classes = []                                         # reduction lengths in order of first appearance = the planner's class order
    for p in problems:
        if p[2] not in classes:
            classes.append(p[2])
    for x in range(8):
        cnt = 0
        for k in range(2):
            pi, m_lo, m_cnt, n_lo, n_cnt = seg[x][k]
            if m_cnt * n_cnt == 0:
                continue
            assert classes.index(problems[pi][2]) == k or len(classes) == 1     # block k of an XCD belongs to class k
            for mi in range(m_lo, m_lo + m_cnt):
                for ni in range(n_lo, n_lo + n_cnt):
                    assert (pi, mi, ni) not in seen, (name, pi, mi, ni)
                    seen[(pi, mi, ni)] = x
            cnt += m_cnt * n_cnt
        per_xcd.append(cnt)
    want = {(i, mi, ni) for i, (M, N, _) in enumerate(problems) for mi in range(-(-M // bm)) for ni in range(-(-N // bn))}
    assert set(seen) == want, name
    assert len(set(per_xcd)) == 1 and gx == 8 * per_xcd[0] and per_xcd[0] <= cap, (name, per_xcd, gx)
    # equal K-steps per XCD: one block of each class everywhere
    work = [sum(seg[x][k][2] * seg[x][k][4] * problems[seg[x][k][0]][2] for k in range(2) if seg[x][k][2]) for x in range(8)]
    assert len(set(work)) == 1, (name, work)


@pytest.mark.parametrize("name,problems,bm,bn,cap", [
    ("three problems: 8 % 3 != 0", [(1280, 768, 768)] * 3, 128, 128, 32),
    ("three reduction lengths", [(768, 768, 256), (768, 768, 512), (768, 768, 1024)], 128, 128, 32),
    ("more tiles per XCD than one round", [(4096, 768, 768)] * 4, 128, 128, 32),
    ("a tile grid that cannot be cut in two equal rectangles", [(640, 384, 768)] * 4, 128, 128, 32),
    ("unequal problems in one class", [(1280, 768, 768), (640, 768, 768)] * 2, 128, 128, 32),
])
def test_xcd_block_map_planner_declines_what_it_cannot_balance(name, problems, bm, bn, cap):
    gx, _ = _plan(problems, bm, bn, cap)
    assert gx == 0, (name, gx)


def test_xcd_block_map_planner_argument_checks():
    lib = L.load()
    assert lib.dvgr_dbg_plan_xcd_blocks(0, None, None, None, 128, 128, 32, None) < 0
    assert lib.dvgr_dbg_plan_xcd_blocks(13, None, None, None, 128, 128, 32, None) < 0
    one = (C.c_int * 1)(768)
    assert lib.dvgr_dbg_plan_xcd_blocks(1, C.addressof(one), C.addressof(one), C.addressof(one), 0, 128, 32, None) < 0
    # one problem on all eight XCDs: 6 x 6 tiles cannot be cut into 8 equal rectangles, 8 x 6 can (4 x 2 blocks of 2 x 3)
    assert _plan([(768, 768, 768)])[0] == 0
    gx, seg = _plan([(1024, 768, 768)])
    assert gx == 8 * 6 and all(seg[x][0][2] * seg[x][0][4] == 6 for x in range(8))
