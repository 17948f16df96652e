#!/usr/bin/env python3
"""The secondary figure of bench.py alone: full GKAN_Nodes training step on the headline graph (for rocprofv3).
usage: python tools/model_step.py [steps] [conv_type]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, kagnn_amd
from kagnn_amd import harness, ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
conv = sys.argv[2] if len(sys.argv) > 2 else "gin"
dev = torch.device("cuda", 0)
n, e, f = 1_000_000, 10_000_000, 64
graph = ops.GraphIndex(bench.powerlaw_graph(n, e, 0).to(dev), n)
x = (torch.randn(n, f, generator=torch.Generator().manual_seed(0)) * 0.25).to(dev)
torch.manual_seed(0)
model = kagnn_amd.GKAN_Nodes(conv, 3, f, f, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2).to(dev)
y = torch.randint(0, 40, (n,), generator=torch.Generator().manual_seed(2)).to(dev)
mask = torch.ones(n, dtype=torch.bool, device=dev)
t, _ = harness.time_model(model, x, graph, y, mask, nb_epochs=steps, warmup=2)
print("ms_per_step", t * 1e3)
