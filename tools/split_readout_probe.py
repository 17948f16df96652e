#!/usr/bin/env python3
"""GPU box: GKAN_Nodes('gin') training step with the skip read-out over column blocks (one forward launch, skip gradients handed
to the convolutions) vs over the concatenation, by graph size.  usage: python tools/split_readout_probe.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch, kagnn_amd
    from kagnn_amd.harness import time_model
    from oracle import kan_oracle as orc
    n = int(sys.argv[1]); e = 7 * n
    ei = orc.powerlaw_graph(n, e, seed=1).to("cuda")
    x = (torch.randn(n, 128) * 0.5).cuda(); y = torch.randint(0, 40, (n,)).cuda(); mask = (torch.rand(n) < 0.5).cuda()
    torch.manual_seed(0)
    m = kagnn_amd.GKAN_Nodes("gin", 3, 128, 64, 40, grid_size=5, spline_order=3, hidden_layers=2).cuda()
    t, _ = time_model(m, x, ei, y, mask, nb_epochs=20, warmup=3)
    print(f"{t * 1e3:.3f}")
    sys.exit(0)
for n in (2000, 10000, 30000, 100000):
    row = []
    for thr in ("0", "1000000000"):
        env = dict(os.environ, KAGNN_SPLIT_READOUT_MIN_ROWS=thr)
        row.append(subprocess.run([sys.executable, __file__, str(n)], env=env, capture_output=True, text=True).stdout.strip())
    print(f"N={n}: blocks {row[0]} ms   concat {row[1]} ms", flush=True)
