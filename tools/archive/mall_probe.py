#!/usr/bin/env python3
"""Does the gather speed up when the gathered matrix fits the 256 MB Infinity Cache?  E fixed, N varied."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kagnn_amd import ops
dev = 'cuda'; e = 10_000_000
for n in (62_500, 125_000, 250_000, 500_000, 1_000_000, 2_000_000):
    g = torch.Generator().manual_seed(0)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)]).to(dev)
    gi = ops.GraphIndex(ei, n)
    x = torch.randn(n, 64, device=dev)
    fn = lambda: ops._aggregate_raw(x, gi, False, 1.0, None, None, None, None, False)
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"N={n:8d} x={n*256/1e6:6.0f} MB  agg {dt*1e3:.3f} ms  gather {e*256/dt/1e12:.2f} TB/s", flush=True)
