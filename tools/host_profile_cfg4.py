#!/usr/bin/env python3
"""cProfile of the host side of the config-4 (ZINC-like mini-batch) training step."""
import os, sys, cProfile, pstats, runpy
sys.argv = [sys.argv[0], "4"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs_sweep.py"))
import torch
zstep = ns["zstep"]
for _ in range(5): zstep()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): zstep()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
