"""Times dvgr_scope_u_loss kernels for one shape (debug aid).  usage: loss_probe.py L B N D"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import losses
L, B, N, D = [int(x) for x in sys.argv[1:5]]
lists = [torch.randn(L, B, N, D, device="cuda") for _ in range(4)]
vis = torch.randn(B, N, D, device="cuda")
raw = (vis, None, None, *lists)
for _ in range(5):
    losses.fused_scope_u_loss(raw)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    losses.fused_scope_u_loss(raw)
b.record()
torch.cuda.synchronize()
print(f"L={L} B={B} N={N} D={D} dbg={os.environ.get('DVGR_LOSS_DBG')}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call")
