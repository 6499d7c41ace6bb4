import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L
lib = L.load()
st = lambda: torch.cuda.current_stream().cuda_stream
def planes(x):
    n = x.numel(); pl = torch.empty(2 * n, dtype=torch.bfloat16, device="cuda")
    assert lib.dvgr_dbg_split(x.data_ptr(), pl.data_ptr(), n, n, st()) == 0
    return pl, n
def run(M, N, K, prec, dbg, reps=20):
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    pa, na = planes(A); pb, nb = planes(B)
    C = torch.empty(M, N, device="cuda")
    def call():
        assert lib.dvgr_dbg_gemm_planes(0, M, N, K, A.data_ptr(), pa.data_ptr(), K, na, B.data_ptr(), pb.data_ptr(), K, nb, C.data_ptr(), N, dbg << 4, 0, prec, st()) == 0
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M, N = 1280, 1536
for K in (768, 3072):
    for prec in (2, 1):
        print(f"planes K{K} prec{prec}: full {run(M,N,K,prec,0):.1f}  no-epilogue {run(M,N,K,prec,1):.1f}  no-loop {run(M,N,K,prec,3):.1f} us")
