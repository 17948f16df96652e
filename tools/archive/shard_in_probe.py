"""GPU box: KANLinear kernels at the input widths the feature-sharded layer hands each rank (in = F/P), N = 1M rows:
headline shape (out 64, grid 5) and config 3's (out 128, grid 8)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import kagnn_amd
from kagnn_amd import ops
dev = 'cuda'; n = 1_000_000
for fout, grid, widths in ((64, 5, (8, 16, 32, 64)), (128, 8, (16, 32, 64, 128))):
    for fin in widths:
        lay = kagnn_amd.KANLinear(fin, fout, grid_size=grid, spline_order=3).to(dev)
        h = (torch.randn(n, fin, device=dev) * 0.3).requires_grad_(True); gy = torch.randn(n, fout, device=dev)
        def fb():
            lay.zero_grad(); h.grad = None; lay(h).backward(gy)
        for _ in range(3): fb()
        tm = ops.EntryPointTimer(); ops.set_timer(tm)
        for _ in range(5): fb()
        torch.cuda.synchronize(); ops.set_timer(None)
        print(f"out {fout} grid {grid} in {fin}", {k[6:]: round(v["total_ms"] / 5, 3) for k, v in tm.summary().items()}, flush=True)
