"""Host time of kagnn_amd.harness.Adam.step() vs torch's fused Adam on 40 small tensors (diagnostic)."""
import os, sys, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kagnn_amd.harness import Adam
dev = "cuda:0"
ps = [torch.nn.Parameter(torch.randn(64, 64, 8 if k % 3 == 0 else 1, device=dev)) for k in range(40)]
gsets = [[torch.randn_like(p) for p in ps] for _ in range(4)]
for name, opt in (("harness.Adam", Adam(ps, lr=1e-3)), ("torch fused", torch.optim.Adam(ps, lr=1e-3, fused=True))):
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(200):
            for p, g in zip(ps, gsets[s % 4]):
                p.grad = g
            opt.step()
        host = (time.perf_counter() - t0) / 200
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 200
        print(f"{name}: host {host * 1e6:.1f} us/step, wall {wall * 1e6:.1f} us/step", flush=True)
opt = Adam(ps, lr=1e-3)
pr = cProfile.Profile(); pr.enable()
for s in range(200):
    for p, g in zip(ps, gsets[s % 4]):
        p.grad = g
    opt.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
