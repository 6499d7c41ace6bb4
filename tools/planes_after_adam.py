import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import synth
B, N, Lr, T, D, WD = 8, 8, 2, 10, 768, 300
V, FR, VQ, VA = 256, 4, 200, 12
dev = torch.device("cuda", 0)
vocab = {"question_token_to_idx": {str(i): i for i in range(VQ)}, "answer_token_to_idx": {str(i): i for i in range(VA)}}
torch.manual_seed(1)
m = dv.DualVGR(vision_dim=V, module_dim=D, word_dim=WD, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1, unit_layers=Lr, dtype="bf16x3").to(dev).train()
app = torch.rand(B, N, FR, V, device=dev); mot = torch.rand(B, N, V, device=dev)
qlen = torch.randint(3, T + 1, (B,), device=dev); qlen[0] = T
q = torch.randint(2, VQ, (B, T), device=dev) * (torch.arange(T, device=dev)[None] < qlen[:, None])
ans = torch.randint(0, VA, (B,), device=dev)
ts = dv.TrainStep(m, lr=1e-3, alpha=1.0, beta=1e-8, max_norm=12.0)
unit = m.visual_input_unit
for i in range(4):
    out = ts(app, mot, q, qlen, ans)
    torch.cuda.synchronize()
    planes = unit._lowp.clone()
    flat = unit.flat_params().clone()
    unit.params_changed(); unit.lowp_ptr(); torch.cuda.synchronize()
    fresh = unit._lowp
    n = flat.numel()
    hi = planes[:n].float(); lo = planes[n:].float()
    rec = hi + lo
    print(i, "planes equal fresh:", bool(torch.equal(planes, fresh)), "max |hi+lo - p| / max|p|:", float((rec - flat).abs().max() / flat.abs().max()),
          "ce", float(out["ce"]), "gnorm", float(out["grad_sqnorm"]) ** 0.5)
