#!/usr/bin/env python3
"""GPU box: which aten / kagnn ops make up one GKAN_Nodes training step on the headline graph (torch.profiler, device time
per op name per step, with input shapes for the aten elementwise ones).  PYTHONPATH=. python tools/model_step_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, kagnn_amd
from kagnn_amd import harness, ops
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
n, e, f = 1_000_000, 10_000_000, 64
graph = ops.GraphIndex(bench.powerlaw_graph(n, e, 0).to(dev), n)
x = (torch.randn(n, f, generator=torch.Generator().manual_seed(0)) * 0.25).to(dev)
torch.manual_seed(0)
model = kagnn_amd.GKAN_Nodes("gin", 3, f, f, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2).to(dev)
y = torch.randint(0, 40, (n,), generator=torch.Generator().manual_seed(2)).to(dev)
mask = torch.ones(n, dtype=torch.bool, device=dev)
harness.time_model(model, x, graph, y, mask, nb_epochs=2, warmup=2)
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    t, _ = harness.time_model(model, x, graph, y, mask, nb_epochs=STEPS, warmup=0)
print("ms_per_step", t * 1e3)
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, "self_device_time_total", getattr(ev, "self_cuda_time_total", 0))
    if dt > 0:
        rows.append((dt / STEPS, ev.count / STEPS, ev.key, str(ev.input_shapes)[:90]))
for dt, cnt, key, shp in sorted(rows, reverse=True)[:45]:
    print(f"{dt:9.1f} us/step {cnt:6.1f}x  {key[:50]:50s} {shp}")
