"""GPU: the building-block kernels behind the debug hooks of the C ABI (include/dvgr_debug.h):
the MFMA contraction engine in its three operand layouts, column sums and the dropout RNG.
Integer-valued operands make the fp32 MFMA results exact, so those cases compare bit for bit
(an asymmetric B catches a transposed C write)."""
import ctypes as C

import numpy as np
import pytest
import torch

from dualvgr_videoqa_amd import _lib as L
from dualvgr_videoqa_amd import dropout

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _gemm(mode, A, B, M, N, K, bias=None, act=0, accumulate=False, C0=None, A2=None, B2=None, kseg=0,
          ka=0, kb=0, kc=0, prec=0):
    lib = L.load()
    Cm = C0.clone() if C0 is not None else torch.full((M, N), float("nan"), device="cuda")
    rc = lib.dvgr_dbg_gemm(mode, M, N, K, kseg, A.data_ptr(), None if A2 is None else A2.data_ptr(), A.stride(0),
                           B.data_ptr(), None if B2 is None else B2.data_ptr(), B.stride(0), Cm.data_ptr(), Cm.stride(0),
                           None if bias is None else bias.data_ptr(), act, int(accumulate), ka, kb, kc, prec, _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return Cm


def _ints(shape, lo=-4, hi=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float().cuda()


SHAPES = [(15, 128, 64), (16, 16, 4), (80, 192, 768), (100, 300, 128), (64, 768, 300), (3, 64, 32),
          (1280, 1536, 768), (257, 132, 36), (130, 260, 2560)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nt_exact_integers(M, N, K):
    A, B = _ints((M, K), seed=1), _ints((N, K), seed=2)
    got = _gemm(0, A, B, M, N, K)
    want = (A.double() @ B.double().T).float()
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nn_exact_integers(M, N, K):
    A, B = _ints((M, K), seed=3), _ints((K, N), seed=4)
    got = _gemm(1, A, B, M, N, K)
    want = (A.double() @ B.double()).float()
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("M,N,K", [(128, 64, 15), (16, 16, 4), (192, 768, 80), (300, 100, 128), (768, 300, 64),
                                   (64, 32, 3), (1536, 768, 1280), (132, 256, 37), (512, 768, 2561)])
def test_gemm_tn_exact_integers(M, N, K):
    M4, N4 = M // 4 * 4, N // 4 * 4
    A, B = _ints((K, M4), seed=5), _ints((K, N4), seed=6)
    got = _gemm(2, A, B, M4, N4, K)
    want = (A.double().T @ B.double()).float()
    assert torch.equal(got, want), float((got - want).abs().max())


def test_gemm_tn_two_row_segments():
    K0, K1, M, N = 37, 51, 64, 96
    A0, A1, B0, B1 = _ints((K0, M), seed=7), _ints((K1, M), seed=8), _ints((K0, N), seed=9), _ints((K1, N), seed=10)
    got = _gemm(2, A0, B0, M, N, K0 + K1, A2=A1, B2=B1, kseg=K0)
    want = (A0.double().T @ B0.double() + A1.double().T @ B1.double()).float()
    assert torch.equal(got, want)
    # NN with the B operand (k rows) in two segments
    A = _ints((20, K0 + K1), seed=11)
    got = _gemm(1, A, B0, 20, N, K0 + K1, B2=B1, kseg=K0)
    want = (A.double() @ torch.cat([B0, B1]).double()).float()
    assert torch.equal(got, want)


def test_gemm_strided_views_bias_activation_accumulate():
    torch.manual_seed(0)
    M, N, K = 45, 192, 128
    big = torch.randn(M, 4 * K, device="cuda")
    A = big[:, K:2 * K]                               # lda = 4K, a column slice like dWh[:, g*4F:]
    W = torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda")
    ref = A.double() @ W.double().T + bias.double()
    got = _gemm(0, A, W, M, N, K, bias=bias)
    assert float((got - ref.float()).abs().max()) < 2e-4 * float(ref.abs().max())
    got = _gemm(0, A, W, M, N, K, bias=bias, act=1)
    assert float((got - torch.tanh(ref).float()).abs().max()) < 1e-5
    got = _gemm(0, A, W, M, N, K, bias=bias, act=2)
    assert float((got - torch.nn.functional.elu(ref).float()).abs().max()) < 2e-4 * float(ref.abs().max())
    C0 = torch.randn(M, N, device="cuda")
    got = _gemm(0, A, W, M, N, K, accumulate=True, C0=C0)
    assert float((got - (C0.double() + A.double() @ W.double().T).float()).abs().max()) < 2e-4 * float(ref.abs().max())
    # output written into a wider buffer (ldc > N) leaves the other columns untouched
    wide = torch.full((M, 2 * N), 7.0, device="cuda")
    lib = L.load()
    assert lib.dvgr_dbg_gemm(0, M, N, K, 0, A.data_ptr(), None, A.stride(0), W.data_ptr(), None, K, wide[:, N:].data_ptr(),
                             2 * N, None, 0, 0, 0, 0, 0, 0, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.all(wide[:, :N] == 7.0)
    assert float((wide[:, N:] - (A.double() @ W.double().T).float()).abs().max()) < 2e-4 * float(ref.abs().max())


def test_tanh_epilogue_keeps_a_nan_a_nan():
    """the hardware-exp tanh of the S4 epilogue clamps |x| with fminf, which would turn a NaN pre-activation
    into +-1 and hide a diverged run; the reference's tanh propagates it (model/Attention.py:16)."""
    torch.manual_seed(1)
    M, N, K = 128, 128, 64
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda")
    A[5, 7] = float("nan")
    A[9, 0] = float("inf")
    W[:, 0] = W[:, 0].abs() + 0.1          # +inf * positive = +inf -> tanh = 1 (the correct limit)
    for prec in (0, 2):                    # DVGR_F32, DVGR_BF16X3
        got = _gemm(0, A, W, M, N, K, act=1, prec=prec)
        assert torch.isnan(got[5]).all(), prec
        assert torch.all(got[9] == 1.0), prec
        ok = torch.ones(M, dtype=torch.bool, device="cuda")
        ok[5] = ok[9] = False
        assert torch.isfinite(got[ok]).all()


def test_keep_mask_device_equals_numpy_replica():
    lib = L.load()
    n = 300_001
    for key in (1, 0x9E3779B1, dropout.site_key(42, 17)):
        keep = torch.empty(n, device="cuda")
        assert lib.dvgr_dbg_keep_mask(key, n, keep.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        assert np.array_equal(keep.cpu().numpy(), dropout.keep_mask(key, n))


def test_gemm_dropout_on_operands_and_result():
    M, N, K = 50, 64, 96
    key = dropout.site_key(7, 3)
    A, W = _ints((M, K), seed=12), _ints((N, K), seed=13)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(0.15))
    mA = torch.from_numpy(dropout.keep_mask(key, M * K).reshape(M, K)).cuda()
    got = _gemm(0, A, W, M, N, K, ka=key)
    want = ((A * mA * float(scale)).double() @ W.double().T)
    assert float((got.double() - want).abs().max()) < 1e-4 * float(want.abs().max())
    X = _ints((K, N), seed=14)                       # TN: B operand masked, index = k * ldb + n
    G = _ints((K, 48), seed=15)                       # IC operands need a multiple of 4 columns
    Mg = G.shape[1]
    mX = torch.from_numpy(dropout.keep_mask(key, K * N).reshape(K, N)).cuda()
    got = _gemm(2, G, X, Mg, N, K, kb=key)
    want = G.double().T @ (X * mX * float(scale)).double()
    assert float((got.double() - want).abs().max()) < 1e-4 * float(want.abs().max())
    B2 = _ints((K, N), seed=16)                      # NN: result masked, index = m * ldc + n
    mC = torch.from_numpy(dropout.keep_mask(key, M * N).reshape(M, N)).cuda()
    got = _gemm(1, A, B2, M, N, K, kc=key)
    want = (A.double() @ B2.double()) * mC.double() * float(scale)
    assert float((got.double() - want).abs().max()) < 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("rows,cols", [(1, 1), (960, 1), (64, 8), (1280, 1536), (2560, 768), (37, 70)])
def test_colsum(rows, cols):
    lib = L.load()
    torch.manual_seed(1)
    Mx = torch.randn(rows, cols, device="cuda")
    w = torch.randn(rows, device="cuda")
    out = torch.empty(cols, device="cuda")
    assert lib.dvgr_dbg_colsum(Mx.data_ptr(), None, rows, cols, cols, out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    ref = Mx.double().sum(0)
    assert float((out.double() - ref).abs().max()) < 1e-5 * max(1.0, float(Mx.abs().double().sum(0).max()))
    assert lib.dvgr_dbg_colsum(Mx.data_ptr(), w.data_ptr(), rows, cols, cols, out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    ref = (Mx.double() * w.double()[:, None]).sum(0)
    assert float((out.double() - ref).abs().max()) < 1e-5 * max(1.0, float((Mx.abs() * w.abs()[:, None]).double().sum(0).max()))


# ---- low-precision MFMA modes --------------------------------------------------------------------
LP_SHAPES = [(80, 192, 768), (100, 300, 128), (64, 768, 300), (1280, 1536, 768), (257, 132, 36), (15, 128, 64)]


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", LP_SHAPES)
def test_gemm_bf16_exact_on_small_integers(mode, M, N, K):
    """Integers in [-4, 4] are exact in bf16, so plain-bf16 MFMA must reproduce the fp32 result bit for
    bit in all three operand layouts (this is the A/B lane-map check of the bf16 kernel)."""
    M, N = M // 4 * 4, N // 4 * 4
    if mode == 0:
        A, B = _ints((M, K), seed=21), _ints((N, K), seed=22)
        want = A.double() @ B.double().T
    elif mode == 1:
        A, B = _ints((M, K), seed=23), _ints((K, N), seed=24)
        want = A.double() @ B.double()
    else:
        A, B = _ints((K, M), seed=25), _ints((K, N), seed=26)
        want = A.double().T @ B.double()
    for prec in (1, 2):
        got = _gemm(mode, A, B, M, N, K, prec=prec)
        assert torch.equal(got, want.float()), (prec, float((got - want.float()).abs().max()))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gemm_bf16x3_is_fp32_grade(mode):
    """Random fp32 operands: bf16x3 within 2e-5 of the exact product (relative to max |C|), plain bf16
    within 2e-2 -- the tolerances the unit-level tests of those modes build on."""
    torch.manual_seed(3)
    M, N, K = 320, 384, 768
    if mode == 0:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05
        want = A.double() @ B.double().T
    elif mode == 1:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda") * 0.05
        want = A.double() @ B.double()
    else:
        A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda") * 0.05
        want = A.double().T @ B.double()
    scale = float(want.abs().max())
    e32 = float((_gemm(mode, A, B, M, N, K, prec=0).double() - want).abs().max()) / scale
    ex3 = float((_gemm(mode, A, B, M, N, K, prec=2).double() - want).abs().max()) / scale
    e16 = float((_gemm(mode, A, B, M, N, K, prec=1).double() - want).abs().max()) / scale
    print(f"  mode {mode}: rel err fp32 {e32:.2e}  bf16x3 {ex3:.2e}  bf16 {e16:.2e}")
    assert e32 < 2e-6 and ex3 < 2e-5 and e16 < 2e-2


def test_gemm_bf16x3_dropout_and_segments():
    M, N, K = 50, 64, 96
    key = dropout.site_key(7, 3)
    A, W = _ints((M, K), seed=12), _ints((N, K), seed=13)
    mA = torch.from_numpy(dropout.keep_mask(key, M * K).reshape(M, K)).cuda()
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(0.15)))
    got = _gemm(0, A, W, M, N, K, ka=key, prec=2)
    want = ((A * mA * scale).double() @ W.double().T)
    assert float((got.double() - want).abs().max()) < 1e-4 * float(want.abs().max())
    K0, K1 = 37, 51
    A0, A1, B0, B1 = _ints((K0, 64), seed=7), _ints((K1, 64), seed=8), _ints((K0, 96), seed=9), _ints((K1, 96), seed=10)
    got = _gemm(2, A0, B0, 64, 96, K0 + K1, A2=A1, B2=B1, kseg=K0, prec=2)
    assert torch.equal(got, (A0.double().T @ B0.double() + A1.double().T @ B1.double()).float())


@pytest.mark.parametrize("mode,M,N,K", [(2, 768, 768, 2560), (2, 128, 128, 2560), (1, 64, 300, 1536), (0, 256, 128, 640)])
@pytest.mark.parametrize("prec", [0, 2])
def test_gemm_splitk_atomic_partials(mode, M, N, K, prec):
    """Few output tiles + long K: the launcher splits K over several workgroups that add their partial tiles
    into the (accumulating) output with float atomics.  Integer operands keep every partial exact, so the
    result is order independent and must match bit for bit."""
    if mode == 0:
        A, B = _ints((M, K), seed=31), _ints((N, K), seed=32)
        want = A.double() @ B.double().T
    elif mode == 1:
        A, B = _ints((M, K), seed=33), _ints((K, N), seed=34)
        want = A.double() @ B.double()
    else:
        A, B = _ints((K, M), seed=35), _ints((K, N), seed=36)
        want = A.double().T @ B.double()
    C0 = _ints((M, N), seed=37)
    got = _gemm(mode, A, B, M, N, K, accumulate=True, C0=C0, prec=prec)
    assert torch.equal(got, (C0.double() + want).float()), float((got.double() - C0.double() - want).abs().max())


# ---- pre-split bf16 planes -------------------------------------------------------------------------
def _planes(x):
    lib = L.load()
    n = x.numel()
    pl = torch.empty(2 * n, dtype=torch.bfloat16, device="cuda")
    assert lib.dvgr_dbg_split(x.data_ptr(), pl.data_ptr(), n, n, _stream()) == 0
    return pl, n


def _gemm_planes(mode, A, B, M, N, K, prec, accumulate=False, C0=None):
    lib = L.load()
    pa, na = _planes(A)
    pb, nb = _planes(B)
    Cm = C0.clone() if C0 is not None else torch.full((M, N), float("nan"), device="cuda")
    rc = lib.dvgr_dbg_gemm_planes(mode, M, N, K, A.data_ptr(), pa.data_ptr(), A.stride(0), na, B.data_ptr(), pb.data_ptr(),
                                  B.stride(0), nb, Cm.data_ptr(), Cm.stride(0), 0, int(accumulate), prec, _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return Cm


def test_split_planes_reconstruct():
    torch.manual_seed(5)
    x = torch.randn(4096, device="cuda") * 3
    pl, n = _planes(x)
    torch.cuda.synchronize()
    hi, lo = pl[:n].float(), pl[n:].float()
    assert torch.equal(hi, x.to(torch.bfloat16).float())                 # hi is the round-to-nearest bf16
    assert float((hi + lo - x).abs().max()) < 2 ** -15 * float(x.abs().max())


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(80, 192, 768), (1280, 1536, 768), (136, 264, 2560), (128, 128, 32)])
def test_gemm_planes_kernel(mode, M, N, K):
    """Operands handed over as hi / lo planes: exact on small integers in both bf16 modes, and bf16x3 keeps
    fp32-grade accuracy on random operands (same bounds as the converting kernels)."""
    shapes = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[mode]
    A, B = _ints(shapes[0], seed=41), _ints(shapes[1], seed=42)
    ref = lambda a, b: (a.double() @ b.double().T) if mode == 0 else (a.double() @ b.double()) if mode == 1 else (a.double().T @ b.double())
    want = ref(A, B)
    for prec in (1, 2):
        got = _gemm_planes(mode, A, B, M, N, K, prec)
        assert torch.equal(got, want.float()), (prec, float((got - want.float()).abs().max()))
    torch.manual_seed(6)
    A, B = torch.randn(shapes[0], device="cuda"), torch.randn(shapes[1], device="cuda") * 0.05
    want = ref(A, B)
    scale = float(want.abs().max())
    assert float((_gemm_planes(mode, A, B, M, N, K, 2).double() - want).abs().max()) / scale < 2e-5
    assert float((_gemm_planes(mode, A, B, M, N, K, 1).double() - want).abs().max()) / scale < 2e-2


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(2560, 6144, 64), (2568, 6024, 96)])
def test_gemm_planes_wide_kernel(mode, M, N, K):
    """>= 480 tiles of 128 x 256: the launcher takes gemm_planes_wide_kernel (64 x 64 per wave).  Exact on small integers
    in both bf16 modes, ragged edges included (the last tile's second B half holds 8 valid columns), fp32-grade on random
    operands in bf16x3."""
    shapes = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[mode]
    A, B = _ints(shapes[0], seed=51), _ints(shapes[1], seed=52)
    ref = lambda a, b: (a.double() @ b.double().T) if mode == 0 else (a.double() @ b.double()) if mode == 1 else (a.double().T @ b.double())
    want = ref(A, B)
    for prec in (1, 2):
        got = _gemm_planes(mode, A, B, M, N, K, prec)
        assert torch.equal(got, want.float()), (prec, float((got - want.float()).abs().max()))
    torch.manual_seed(7)
    A, B = torch.randn(shapes[0], device="cuda"), torch.randn(shapes[1], device="cuda") * 0.05
    want = ref(A, B)
    assert float((_gemm_planes(mode, A, B, M, N, K, 2).double() - want).abs().max()) / float(want.abs().max()) < 2e-5


@pytest.mark.parametrize("mode,M,N,K", [(2, 1024, 4096, 8192), (2, 3072, 2048, 8192), (0, 1024, 4096, 8192)])
def test_gemm_planes_wide_splitk(mode, M, N, K):
    """Long reductions into an accumulating output (the LSTM's weight gradients): the wide kernel with K split over
    workgroups that add their partial tiles with float atomics -- exact on small integers, on top of a non-zero C."""
    shapes = {0: ((M, K), (N, K)), 2: ((K, M), (K, N))}[mode]
    A, B = _ints(shapes[0], seed=61), _ints(shapes[1], seed=62)
    want = (A.double() @ B.double().T) if mode == 0 else (A.double().T @ B.double())
    C0 = _ints((M, N), seed=63)
    got = _gemm_planes(mode, A, B, M, N, K, 2, accumulate=True, C0=C0)
    assert torch.equal(got, (C0.double() + want).float()), float((got.double() - C0.double() - want).abs().max())


# ---- bias gradients inside the weight-gradient contraction, several problems into one output ------------------------------
def _planes(x):
    lib, st = L.load(), _stream
    n = x.numel()
    pl = torch.empty(2 * n, device="cuda", dtype=torch.bfloat16)
    assert lib.dvgr_dbg_split(x.data_ptr(), pl.data_ptr(), n, n, st()) == 0
    return pl, n


@pytest.mark.parametrize("prec", [L.DVGR_F32, L.DVGR_BF16X3])
@pytest.mark.parametrize("M,N,K,count", [(768, 768, 1280, 4), (256, 128, 2560, 2), (200, 72, 96, 3), (1536, 768, 1280, 2)])
def test_gemm_tn_carries_the_bias_gradient(M, N, K, count, prec):
    """dW = dY^T . X with d b = dY^T . 1 taken from the A fragments (GemmProblem::rowsum): `count` problems in one launch
    (split-K included: K = 2560 with few tiles), odd extents, both operand modes; against float64."""
    lib, st = L.load(), _stream
    g = torch.Generator().manual_seed(5)
    A = torch.randn(K, M, generator=g).cuda()            # dY [k][m]
    Bm = torch.randn(K, N, generator=g).cuda()           # X  [k][n]
    pa, na = _planes(A)
    pb, nb = _planes(Bm)
    Cm = torch.zeros(count * M, N, device="cuda")
    rs = torch.zeros(count * M, device="cuda")
    rc = lib.dvgr_dbg_gemm_planes_batch(2, M, N, K, 0, A.data_ptr(), pa.data_ptr(), A.stride(0), na, Bm.data_ptr(), pb.data_ptr(),
                                        Bm.stride(0), nb, Cm.data_ptr(), N, 0, 0, 1, prec, count, 3, rs.data_ptr(), 0, st())
    if K <= 256:                       # batch-sized reductions run on the small kernels, which carry no row sums: the
        assert rc == -1                # engine says so (gemm_rowsum_supported) and the caller uses a column-sum job
        return
    assert rc == 0, rc
    torch.cuda.synchronize()
    want = (A.double().T @ Bm.double()).cpu()
    want_rs = A.double().sum(0).cpu()
    tol = 2e-6 if prec == L.DVGR_F32 else 2e-5
    for i in range(count):
        assert float((Cm[i * M:(i + 1) * M].double().cpu() - want).abs().max()) < tol * float(want.abs().max())
        assert float((rs[i * M:(i + 1) * M].double().cpu() - want_rs).abs().max()) < tol * float(A.abs().sum(0).max())


@pytest.mark.parametrize("prec", [L.DVGR_F32, L.DVGR_BF16X3])
@pytest.mark.parametrize("M,N,K,count", [(960, 768, 768, 3), (90, 64, 64, 2), (36, 64, 128, 5)])
def test_gemm_nn_several_problems_add_into_one_output(M, N, K, count, prec):
    """The batched d(dq): `count` problems of one launch ADD their tiles atomically into one pre-zeroed output
    (GemmProblem::atomic_out), on the ring kernels and on the batch-sized kernels (M <= 128)."""
    lib, st = L.load(), _stream
    g = torch.Generator().manual_seed(6)
    A = torch.randn(M, K, generator=g).cuda()
    Bm = torch.randn(K, N, generator=g).cuda()
    pa, na = _planes(A)
    pb, nb = _planes(Bm)
    Cm = torch.zeros(M, N, device="cuda")
    rc = lib.dvgr_dbg_gemm_planes_batch(1, M, N, K, 0, A.data_ptr(), pa.data_ptr(), A.stride(0), na, Bm.data_ptr(), pb.data_ptr(),
                                        Bm.stride(0), nb, Cm.data_ptr(), N, 0, 1, 1, prec, count, 0, None, 1, st())
    assert rc == 0, rc
    torch.cuda.synchronize()
    want = count * (A.double() @ Bm.double()).cpu()
    tol = 2e-6 if prec == L.DVGR_F32 else 2e-5
    assert float((Cm.double().cpu() - want).abs().max()) < tol * float(want.abs().max())


# ---- the XCD block map (round 6): a different workgroup -> tile map, the same arithmetic -------------------------------------
@pytest.mark.parametrize("mode,M,N,K,count,tag", [(0, 1280, 768, 768, 4, 2),      # S4: 4 problems of 10 x 6 tiles -> 2 XCDs each
                                                  (1, 1280, 768, 768, 4, 4),      # dX
                                                  (2, 768, 768, 1280, 4, 3),      # dWcat alone: one class of 4
                                                  (0, 1280, 1536, 768, 2, 1),     # layer 0's projection: 2 problems, 2 x 2 blocks each
                                                  (0, 1024, 768, 768, 4, 2),      # C3's row count
                                                  (0, 1200, 768, 768, 4, 2),      # a partial last row tile
                                                  (0, 512, 768, 768, 4, 2),       # C4's S4: 96 tiles -> the half-height (64-row) launch, 24 tiles per XCD
                                                  (0, 4096, 768, 768, 4, 2),      # C5's S4: 384 tiles of 128 x 256, 48 per XCD (1.5 rounds)
                                                  (1, 4096, 768, 768, 4, 4),      # C5's dX on the wide tiles
                                                  (0, 1280, 768, 768, 3, 2),      # 8 % 3 != 0: the default map runs
                                                  (2, 1536, 768, 1280, 2, 3)])
def test_xcd_block_map_is_bit_identical(mode, M, N, K, count, tag):
    """GemmBatch::xmap (include/dvgr_debug.h: dvgr_dbg_xmap_tags) changes which workgroup computes which tile, nothing else:
    the outputs (and the TN launch's row sums up to their atomic adds' order: compared to rounding) equal the default map's."""
    lib, st = L.load(), _stream
    g = torch.Generator().manual_seed(11)
    if mode == 2:
        A, Bm = torch.randn(K, M, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
    elif mode == 1:
        A, Bm = torch.randn(M, K, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
    else:
        A, Bm = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda()
    pa, na = _planes(A)
    pb, nb = _planes(Bm)
    outs = {}
    prev = lib.dvgr_dbg_xmap_tags(0)
    try:
        for mask in (0, 127):
            lib.dvgr_dbg_xmap_tags(mask)
            Cm = torch.full((count * M, N), float("nan"), device="cuda")
            rs = torch.zeros(count * M, device="cuda") if mode == 2 else None
            rc = lib.dvgr_dbg_gemm_planes_batch(mode, M, N, K, 0, A.data_ptr(), pa.data_ptr(), A.stride(0), na, Bm.data_ptr(),
                                                pb.data_ptr(), Bm.stride(0), nb, Cm.data_ptr(), N, 0, 0, 0, L.DVGR_BF16X3, count, tag,
                                                None if rs is None else rs.data_ptr(), 0, st())
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs[mask] = (Cm.cpu(), None if rs is None else rs.cpu())
    finally:
        lib.dvgr_dbg_xmap_tags(prev if prev >= 0 else -1)
    assert torch.equal(outs[0][0], outs[127][0])
    assert not torch.isnan(outs[127][0]).any()
    want = (A.double().T @ Bm.double() if mode == 2 else A.double() @ (Bm.double() if mode == 1 else Bm.double().T)).cpu()
    assert float((outs[127][0][:M].double() - want).abs().max()) < 2e-5 * float(want.abs().max())
    if mode == 2:
        assert float((outs[0][1] - outs[127][1]).abs().max()) <= 1e-5 * float(outs[0][1].abs().max())
