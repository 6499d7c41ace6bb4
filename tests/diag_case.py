import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, numpy as np
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import synth
from oracle import unit_torch as O
from helpers import rel_err
N, D, Lr, B, T = [int(x) for x in sys.argv[1:6]]
dtype = sys.argv[6]
Wd = 32
sd_np = synth.xavier_state_dict(O.param_shapes(Wd, D, Lr), seed=11, bias_range=0.1)
inp = synth.unit_inputs(B, N, T, D, Wd, seed=11)
m = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype=dtype)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
m = m.cuda().eval()
m.no_planes = len(sys.argv) > 7 and sys.argv[7] == 'noplanes'
t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
for k in ("A", "M", "dq", "words"): t[k].requires_grad_(True)
outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
Rv = torch.from_numpy(synth.normal("Rv", (B, N, D), 1.0, 11)).cuda()
(outs[0] * Rv).sum().add(sum((x * Rv).sum() for x in outs[3] + outs[6])).backward()
torch.cuda.synchronize()
sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd_np.items()}
ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
for k in ("A", "M", "dq", "words"): ti[k].requires_grad_(True)
oo = O.unit_forward(sd, ti["A"], ti["M"], ti["dq"], ti["words"], ti["qlen"], Lr, 1, naive=False)
(oo[0] * Rv.cpu()).sum().add(sum((x * Rv.cpu()).sum() for x in oo[3] + oo[6])).backward()
print(sys.argv[1:], "gin", {k: f"{rel_err(t[k].grad.cpu(), ti[k].grad):.1e}" for k in ("A", "M", "dq", "words")})
errs = {k: float((p.grad.cpu() - sd[k].grad).abs().max()) / max(float(sd[k].grad.abs().max()), 1e-3) for k, p in m.named_parameters()}
for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:4]: print(f"  {v:.2e} {k}")
