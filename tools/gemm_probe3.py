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
M, N = 1280, 1536
for K in (768, 3072):
    for prec in (2, 1):
        print(f"K{K} prec{prec}: full {run(M,N,K,prec,0):.1f}  no-epilogue {run(M,N,K,prec,1):.1f}  no-compute {run(M,N,K,prec,1|4):.1f}  no-dma {run(M,N,K,prec,1|8):.1f}  neither {run(M,N,K,prec,1|4|8):.1f} us")
