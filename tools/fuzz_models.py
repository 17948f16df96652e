#!/usr/bin/env python3
"""Random GKAN_Nodes configurations (widths, depths, row counts, class counts, chain lengths, dropout, skip on / off, power-law graphs with
hubs in both directions): the default path -- norms folded into their consumers, their backward statistics travelling with the
gradients -- against the same model with every fold switched off (KAGNN_LAZY_NORM = 0, KAGNN_FOLD_NORM_STATS = 0: normalising and
statistics passes).  Loss, input gradient, every parameter gradient and every buffer; 1e-4 of each tensor's largest element with a
floor at 1e-6 of the model's largest gradient (cancelling sums).  usage: python tools/fuzz_models.py [cases] [seed]"""
import copy, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import kagnn_amd
from kagnn_amd import models as M, ops
from oracle import kan_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
M._SPLIT_READOUT_MIN_ROWS = 0
M._SPLIT_READOUT_MIN_ROWS_ONE_LAUNCH = 0
bad = folded = summed = 0
for case in range(cases):
    f_in = rng.choice([64, 128, 64])
    hidden = rng.choice([64, 64, 128, 40, 32])
    mp, hl = rng.choice([1, 2, 3, 4]), rng.choice([1, 2])
    n = rng.choice([700, 5003, 40000, 70001])
    p_drop = rng.choice([0.0, 0.0, 0.0, 0.3])
    classes = rng.choice([7, 40, 64, 70])
    skip = rng.random() < 0.85
    ei = orc.powerlaw_graph(n, 6 * n, seed=case)
    if rng.random() < 0.5:
        ei = torch.cat([ei, ei.flip(0)], dim=1)
    g = ops.GraphIndex(ei.to(DEV), n)
    x = (torch.randn(n, f_in, generator=torch.Generator().manual_seed(case)) * 0.4).to(DEV)
    y = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(case + 1)).to(DEV)
    torch.manual_seed(case)
    model0 = kagnn_amd.GKAN_Nodes("gin", mp, f_in, hidden, classes, skip=skip, grid_size=5, spline_order=3, hidden_layers=hl,
                                  dropout=p_drop).to(DEV)
    label = f"case {case}: f_in {f_in} hidden {hidden} mp {mp} chain {hl} n {n} dropout {p_drop} classes {classes} skip {skip}"
    res = []
    try:
        for on in (True, False):
            M._LAZY_NORM = on
            ops._FOLD_NORM_STATS = on
            model = copy.deepcopy(model0)
            xr = x.clone().requires_grad_(case % 2 == 0)
            torch.manual_seed(1000 + case)
            timer = ops.EntryPointTimer()
            ops.set_timer(timer)
            try:
                loss = ops.softmax_cross_entropy(model(xr, g), y)
                loss.backward()
            finally:
                ops.set_timer(None)
            if on:
                names = [r[0] for r in timer.records]
                folded += names.count("kagnn_batchnorm_stats_affine")
                summed += names.count("kagnn_gin_kan_layer_bwd_bn_sums") + names.count("kagnn_kan_linear_bwd_input_affine_sums")
            res.append([loss.detach().clone()] + ([xr.grad.clone()] if xr.requires_grad else [])
                       + [p.grad.clone() for p in model.parameters()] + [b.clone() for b in model.buffers() if b.dtype.is_floating_point])
        floor = 1e-6 * max(float(t.abs().max()) for t in res[1][1:])
        worst = 0.0
        for k, (a, b) in enumerate(zip(*res)):
            scale = float(b.abs().max())
            err = float((a - b).abs().max())
            worst = max(worst, err / max(1e-4 * scale, floor))
        ok = worst <= 1.0
        print(("ok  " if ok else "FAIL"), label, f"worst {worst:.3f} of tolerance", flush=True)
        bad += not ok
    except Exception as ex:                               # noqa: BLE001
        print("ERR ", label, type(ex).__name__, str(ex)[:200], flush=True)
        bad += 1
print(f"failures: {bad}   (folded norms: {folded}, statistics made by producers: {summed})")
sys.exit(1 if bad else 0)
