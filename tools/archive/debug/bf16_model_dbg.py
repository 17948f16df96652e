import os, sys
os.environ["KAGNN_ACT"] = "bf16"
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
from helpers import bf16_gather_oracle, oracle_node_model_fwd_bwd
n, e = 30000, 210000
ei = orc.powerlaw_graph(n, e, seed=21)
x = torch.randn(n, 128, generator=torch.Generator().manual_seed(22)) * 0.5
gout = torch.randn(n, 40, generator=torch.Generator().manual_seed(23)) / n
torch.manual_seed(1)
L = int(os.environ.get("LAYERS", "3"))
model = kagnn_amd.GKAN_Nodes("gin", L, 128, 64, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2)
state = {k: v.detach().clone() for k, v in model.state_dict().items()}
with bf16_gather_oracle():
    want, gxw, gw = oracle_node_model_fwd_bwd(x, ei, state, gout, "kan", "gin", L, 3, 8192, torch.float64)
model = model.cuda().train()
xd = x.cuda().requires_grad_(True)
out = model(xd, ei.cuda()); out.backward(gout.cuda())
l2 = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
print("logits", l2(out.detach(), want), "gx", l2(xd.grad, gxw))
for name, p in model.named_parameters():
    if p.grad is not None and name in gw and float(gw[name].norm()) > 0:
        print(f"{name:45s} {l2(p.grad, gw[name]):.2e}")
