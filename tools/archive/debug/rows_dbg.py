import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import kagnn_amd
from kagnn_amd import ops
dev='cuda'
def wall(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
for rows in (125000, 250000, 500000):
    chain = kagnn_amd.KAN([64, 64, 64], grid_size=5, spline_order=3).to(dev)
    h = (torch.randn(rows, 64, device=dev) * 0.3).requires_grad_(True); gy = torch.randn(rows, 64, device=dev)
    def fb():
        chain.zero_grad(); h.grad = None; chain(h).backward(gy)
    def f():
        with torch.no_grad(): chain(h)
    print("rows", rows, "fwd+bwd wall", round(wall(fb),3), "fwd wall", round(wall(f),3), flush=True)
    import cProfile, pstats
    if rows == 125000:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(20): fb()
        torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
