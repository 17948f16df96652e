#!/usr/bin/env python3
"""Randomised shapes of FastKANLayer, both precision modes, against the fp64 oracle.
usage: python tools/fuzz_fastkan.py [cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
from helpers import assert_close

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
bad = 0
for it in range(cases):
    n = rng.choice([1, 2, 31, 33, 64, 200, 257, 1000, 3000])
    fi = rng.choice([1, 2, 3, 7, 8, 16, 17, 33, 64, 65, 100, 128, 300, 1100])
    fo = rng.choice([1, 2, 5, 16, 32, 33, 64, 65, 128, 129, 200])
    ng = rng.choice([2, 3, 4, 5, 8, 9, 12, 16, 17, 24, 32])     # 1 divides by zero in the reference constructor too
    use_ln, use_base = rng.random() < 0.7, rng.random() < 0.7
    # LayerNorm over <= 3 features is degenerate (2 features: the output is +-1 whatever the input, the exact input gradient is
    # O(eps) and fp32 -- here and in the reference -- returns rounding noise amplified by rstd ~ 1/sqrt(var + eps))
    use_ln = use_ln and fi >= 4
    mode = rng.choice([ops.PREC_SPLIT, ops.PREC_SPLIT, ops.PREC_FP32])
    tag = f"case {it}: n={n} in={fi} out={fo} ng={ng} ln={use_ln} base={use_base} mode={mode}"
    try:
        torch.manual_seed(it)
        layer = kagnn_amd.FastKANLayer(fi, fo, num_grids=ng, use_base_update=use_base, use_layernorm=use_ln and fi > 1)
        if use_ln and fi > 1:
            layer.layernorm.weight.data.uniform_(0.5, 1.5); layer.layernorm.bias.data.uniform_(-0.3, 0.3)
        x = torch.randn(n, fi) * rng.choice([0.5, 1.3]) + 0.2
        gy = torch.randn(n, fo)
        p64 = {k: v.detach().double() for k, v in layer.state_dict().items()}
        w64 = {k: v.clone().requires_grad_(True) for k, v in p64.items() if k != "rbf.grid"}
        x64 = x.double().requires_grad_(True)
        y64 = orc.fastkan_layer_forward(x64, w64.get("layernorm.weight"), w64.get("layernorm.bias"), p64["rbf.grid"],
                                        layer.rbf.denominator, w64["spline_linear.weight"],
                                        w64.get("base_linear.weight"), w64.get("base_linear.bias"))
        y64.backward(gy.double())
        layer = layer.to(DEV); layer.precision = mode
        xd = x.to(DEV).requires_grad_(True)
        y = layer(xd); y.backward(gy.to(DEV))
        assert_close(y, y64, what="y"); assert_close(xd.grad, x64.grad, what="gx")
        for name, prm in layer.named_parameters():
            if prm.requires_grad:
                assert_close(prm.grad, w64[name].grad, what="g_" + name)
        print("ok  ", tag, flush=True)
    except Exception as ex:
        bad += 1
        print("FAIL", tag, "->", str(ex)[:200], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
