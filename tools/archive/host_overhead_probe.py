#!/usr/bin/env python3
"""Host time to ENQUEUE one headline layer step (no sync inside) vs its device time, at full and P=8 row counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, kagnn_amd
from kagnn_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
ROWS = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1_000_000, 125_000]
for rows in ROWS:
    chain = kagnn_amd.KAN([64, 64, 64], grid_size=5, spline_order=3).to(dev)
    h = (torch.randn(rows, 64, device=dev) * 0.3).requires_grad_(True)
    gy = torch.randn(rows, 64, device=dev)
    def step():
        h.grad = None
        for p in chain.parameters(): p.grad = None
        chain(h).backward(gy)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"KAN chain fwd+bwd rows={rows}: host enqueue {1e3*(t1-t0)/50:.3f} ms/step, total {1e3*(t2-t0)/50:.3f} ms/step")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
