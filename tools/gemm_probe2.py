import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L
lib = L.load()
def run(mode, M, N, K, prec, acc, reps=20):
    if mode == 0: A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    elif mode == 1: A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    else: A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    C = torch.zeros(M, N, device="cuda")
    def call():
        assert lib.dvgr_dbg_gemm(mode, M, N, K, 0, A.data_ptr(), None, A.stride(0), B.data_ptr(), None, B.stride(0), C.data_ptr(), N, None, 0, acc, 0, 0, 0, prec, torch.cuda.current_stream().cuda_stream) == 0
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mode, M, N, K in [(2, 768, 768, 2560), (2, 1536, 768, 1280), (1, 1280, 768, 1536), (0, 1280, 1536, 768), (2, 768, 768, 320), (2,768,768,640)]:
    for prec in (2, 1):
        print(f"mode{mode} M{M} N{N} K{K} prec{prec}: store {run(mode, M, N, K, prec, 0):.1f} us   accumulate/split {run(mode, M, N, K, prec, 1):.1f} us")
