import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L
lib = L.load()
M, N, K, prec = 1280, 1536, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 1
A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
C = torch.empty(M, N, device="cuda")
for _ in range(10):
    assert lib.dvgr_dbg_gemm(0, M, N, K, 0, A.data_ptr(), None, K, B.data_ptr(), None, K, C.data_ptr(), N, None, 0, 0, 0, 0, 0, prec, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
