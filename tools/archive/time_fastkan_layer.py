"""GPU box: FastKAN-GIN conv layer (aggregate + FastKAN([F,F,F], num_grids)) forward+backward on the headline graph, ms per step and
per entry point -- the RBF-basis twin of bench.py's KAN-GIN figure (BASELINE.md section 2 has the reference CPU time of this layer)."""
import sys
import torch
import kagnn_amd
from kagnn_amd import ops
from bench import powerlaw_graph

n, e, f, grids = 1_000_000, 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 8
dev = "cuda:0"
ei = powerlaw_graph(n, e, 0).to(dev)
g = ops.GraphIndex(ei, n)
torch.manual_seed(0)
conv = kagnn_amd.GIFASTKANLayer(f, f, grid_size=grids, hidden_dim=f, nb_layers=2).to(dev)
x = (torch.randn(n, f) * 0.25).to(dev).requires_grad_(True)
gy = torch.randn(n, f).to(dev)


def step():
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    conv(x, g).backward(gy)


for _ in range(5):
    step()
timer = ops.EntryPointTimer()
ops.set_timer(timer)
for _ in range(3):
    step()
torch.cuda.synchronize()
ops.set_timer(None)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(20):
    step()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f"FastKAN-GIN layer F={f} grids={grids}: {ms:.3f} ms per step = {e / ms * 1e3:.3e} edges/s")
print({k: round(v["total_ms"] / 3, 3) for k, v in timer.summary().items()})
