#!/usr/bin/env python3
"""KANLinear fwd+bwd at N=1M, 64->64 for grids beyond 16 coefficients: split mode (coefficient groups) vs exact fp32."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kagnn_amd
from kagnn_amd import ops
dev = 'cuda'; n = 1_000_000
x = (torch.randn(n, 64, device=dev) * 0.5).requires_grad_(True); gy = torch.randn(n, 64, device=dev)
GRIDS = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(5, 3), (8, 3), (13, 3), (14, 3), (20, 3), (29, 3), (32, 4)]
for G, k in GRIDS:
    lay = kagnn_amd.KANLinear(64, 64, grid_size=G, spline_order=k).to(dev)
    row = []
    for mode in (ops.PREC_SPLIT, ops.PREC_FP32):
        lay.precision = mode
        def fb():
            lay.zero_grad(); x.grad = None; lay(x).backward(gy)
        fb(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fb()
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"G={G:2d} k={k} C={G+k:2d}: split {row[0]:7.3f} ms   fp32 {row[1]:7.3f} ms", flush=True)
