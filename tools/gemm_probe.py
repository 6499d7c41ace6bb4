import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L
lib = L.load()
def run(M, N, K, prec, dbg, reps=20):
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    def call():
        assert lib.dvgr_dbg_gemm(0, M, N, K, 0, A.data_ptr(), None, K, B.data_ptr(), None, K, C.data_ptr(), N, None, dbg << 4, 0, 0, 0, 0, prec, torch.cuda.current_stream().cuda_stream) == 0
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N) in [(128, 128), (1280, 1536)]:
    for K in (32, 768):
        for dbg in (0, 1, 2, 3):
            print(f"M{M} N{N} K{K} prec1 dbg{dbg}: {run(M, N, K, 1, dbg):.1f} us")
