#!/usr/bin/env python3
"""Randomised shapes of KANLinear / KAN chains / FastKANLayer, both precision modes, against the fp64 oracle.
usage: python tools/fuzz_kanlinear.py [cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
from helpers import assert_close, oracle_kan_linear_fwd_bwd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
bad = 0
for it in range(cases):
    n = rng.choice([1, 2, 31, 32, 33, 63, 200, 257, 1000, 4097])
    fi = rng.choice([1, 2, 3, 7, 8, 15, 16, 17, 31, 33, 64, 65, 100, 128, 130, 300])
    fo = rng.choice([1, 2, 5, 16, 31, 32, 33, 64, 65, 96, 128, 129, 200])
    G = rng.choice([1, 2, 3, 4, 5, 6, 8, 11, 13, 14, 17, 24, 32])
    k = rng.choice([1, 2, 3, 3, 3, 4])
    mode = rng.choice([ops.PREC_SPLIT, ops.PREC_SPLIT, ops.PREC_FP32])
    tag = f"case {it}: n={n} in={fi} out={fo} G={G} k={k} mode={mode}"
    try:
        gen = torch.Generator().manual_seed(it)
        p = orc.init_kan_linear(fi, fo, G, k, gen)
        x = torch.randn(n, fi, generator=gen) * rng.choice([0.3, 0.8, 2.0])
        gy = torch.randn(n, fo, generator=gen)
        y64, gx64, g64 = oracle_kan_linear_fwd_bwd(x, gy, p, k)
        layer = kagnn_amd.KANLinear(fi, fo, grid_size=G, spline_order=k)
        layer.load_state_dict(p); layer = layer.to(DEV); layer.precision = mode
        xd = x.to(DEV).requires_grad_(True)
        y = layer(xd); y.backward(gy.to(DEV))
        assert_close(y, y64, what="y"); assert_close(xd.grad, gx64, what="gx")
        for nme in ("base_weight", "spline_weight", "spline_scaler"):
            assert_close(getattr(layer, nme).grad, g64[nme], what="g_" + nme)
        print("ok  ", tag, flush=True)
    except Exception as ex:                                   # keep going, report all
        bad += 1
        print("FAIL", tag, "->", str(ex)[:200], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
