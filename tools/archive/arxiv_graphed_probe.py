import sys, os, torch
sys.path.insert(0, os.getcwd())
import kagnn_amd
from kagnn_amd.harness import time_model
from oracle import kan_oracle as orc
dev='cuda'; n, e = 169343, 1166243
ei = orc.powerlaw_graph(n, e, seed=1).to(dev)
x = (torch.randn(n, 128) * 0.5).to(dev); y = torch.randint(0, 40, (n,)).to(dev); mask = (torch.rand(n) < 0.5).to(dev)
for graphed in (False, True, False, True):
    torch.manual_seed(0)
    m = kagnn_amd.GKAN_Nodes('gin', 3, 128, 64, 40, grid_size=5, spline_order=3, hidden_layers=2).to(dev)
    t, losses = time_model(m, x, ei, y, mask, nb_epochs=10, warmup=3, graphed=graphed)
    print("graphed" if graphed else "eager  ", t*1e3, "ms/epoch", losses[-1], flush=True)
