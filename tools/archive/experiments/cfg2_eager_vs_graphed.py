"""BASELINE config 2 (arxiv shape, 3 x KAN-GIN 64): epoch time eager vs captured in a HIP graph, and the host issue time of the
eager epoch (is the epoch device- or host-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kagnn_amd
from kagnn_amd import ops
from kagnn_amd.harness import time_model
from oracle import kan_oracle as orc
dev = "cuda"
n, e = 169343, 1166243
ei = orc.powerlaw_graph(n, e, seed=1).to(dev)
x = (torch.randn(n, 128) * 0.5).to(dev); y = torch.randint(0, 40, (n,)).to(dev); mask = (torch.rand(n) < 0.5).to(dev)
for kind in ("gin", "gcn"):
    torch.manual_seed(0)
    m = kagnn_amd.GKAN_Nodes(kind, 3, 128, 64, 40, grid_size=5, spline_order=3, hidden_layers=2).to(dev)
    for graphed in (False, True, False, True):
        t, losses = time_model(m, x, ei, y, mask, nb_epochs=20, warmup=3, graphed=graphed)
        print(f"cfg2 {kind} {'HIP-graphed' if graphed else 'eager'}: {t * 1e3:.3f} ms/epoch, loss {losses[-1]:.4f}", flush=True)
    # host issue time of the eager epoch
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    def epoch():
        opt.zero_grad(); loss = ops.softmax_cross_entropy(m(x, ei), y, mask, pre_softmax=True); loss.backward(); opt.step()
    for _ in range(3): epoch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): epoch()
    host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
    print(f"cfg2 {kind} eager: host issue {host * 1e3:.3f} ms/epoch, wall {wall * 1e3:.3f} ms/epoch", flush=True)
