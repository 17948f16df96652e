#!/usr/bin/env python3
"""BASELINE config 2 alone (for rocprofv3): GKAN_Nodes('gin', 3, 128, 64, 40) training step at ogbn-arxiv's shape.
usage: python tools/arxiv_step.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kagnn_amd
from kagnn_amd.harness import time_model
from oracle import kan_oracle as orc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, e = 169343, 1166243
ei = orc.powerlaw_graph(n, e, seed=1).to("cuda")
x = (torch.randn(n, 128) * 0.5).cuda(); y = torch.randint(0, 40, (n,)).cuda(); mask = (torch.rand(n) < 0.5).cuda()
torch.manual_seed(0)
m = kagnn_amd.GKAN_Nodes("gin", 3, 128, 64, 40, grid_size=5, spline_order=3, hidden_layers=2).cuda()
t, _ = time_model(m, x, ei, y, mask, nb_epochs=steps, warmup=3)
print("ms_per_epoch", t * 1e3)
