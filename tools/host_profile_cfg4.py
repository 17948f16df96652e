#!/usr/bin/env python3
"""cProfile of the host side of the config-4 (ZINC-like mini-batch) training step as kagnn_amd.harness.train_graph_batches runs it
(embedding encoders, 8 distinct batches, fused Adam).  Prints the wall time per step with the device idle-waiting excluded
(the loop never synchronises) and the 40 largest self-times / cumulative times per step."""
import os, sys, cProfile, pstats, runpy, time
sys.argv = [sys.argv[0], "4h"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs_sweep.py"))
import torch
m, batches = ns["m"], ns["batches"]
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
from kagnn_amd import ops as _ops
loss_fn = _ops.l1_loss


def step(d):
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(m(d).squeeze(), d.y.squeeze())
    loss.backward()
    opt.step()


for _ in range(3):
    for d in batches: step(d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    for d in batches: step(d)
host = (time.perf_counter() - t0) / 80
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 80
print(f"host issue time {host * 1e3:.3f} ms/step, wall {wall * 1e3:.3f} ms/step (device-bound if wall >> host)")
# host issue time per phase (no synchronisation inside the loop: what the Python side costs when the device keeps up)
from kagnn_amd import ops
from kagnn_amd.harness import Adam
own = Adam(m.parameters(), lr=1e-3)
for label, opt, mt in (("torch.optim.Adam(fused=True)", opt, True), ("kagnn_amd.harness.Adam", own, True),
                       ("kagnn_amd.harness.Adam, backward on the calling thread (autograd multithreading off)", own, False),
                       ("kagnn_amd.harness.Adam again", own, True),
                       ("kagnn_amd.harness.Adam, backward on the calling thread, again", own, False)):
  torch.autograd.set_multithreading_enabled(mt)
  ph = {"zero_grad": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0}
  for _ in range(10):
      for d in batches:
          t = [time.perf_counter()]
          opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
          out = m(d); t.append(time.perf_counter())
          loss = ops.l1_loss(out.squeeze(), d.y.squeeze()); t.append(time.perf_counter())
          loss.backward(); t.append(time.perf_counter())
          opt.step(); t.append(time.perf_counter())
          for k, (a, b) in zip(ph, zip(t, t[1:])):
              ph[k] += b - a
  torch.cuda.synchronize()
  print(label, "-- host issue time per phase, us/step:", {k: round(v / 80 * 1e6, 1) for k, v in ph.items()}, "sum", round(sum(ph.values()) / 80 * 1e6, 1))
torch.autograd.set_multithreading_enabled(os.environ.get("KAGNN_PROFILE_MT", "0") == "1")     # (default: as harness.train_graph_batches runs it)
opt = own
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    for d in batches: step(d)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(70)
