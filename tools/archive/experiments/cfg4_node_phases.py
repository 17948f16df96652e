"""Host time inside the pieces of graph_ops._KaginModelFn (forward and backward), by wrapping the raw helpers with timers."""
import os, sys, time, runpy, collections
sys.argv = [sys.argv[0], "4h"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs_sweep.py"))
import torch
from kagnn_amd import ops, graph_ops, _lib
from kagnn_amd.harness import Adam
m, batches = ns["m"], ns["batches"]
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t0; cnt[label or name] += 1
    setattr(mod, name, timed)


for name in ("_embedding_sum_fwd_raw", "_gine_stack_fwd_raw", "_segment_pool_raw", "_kan_fwd_raw", "kan_pack_chain", "_kan_bwd_input_raw",
             "_kan_bwd_weight_raw", "_segment_broadcast_raw", "_gine_stack_bwd_raw", "_embedding_sum_bwd_raw", "graph_index", "_gine_stack_plan",
             "_gine_stack_args", "_hook_free"):
    wrap(graph_ops, name)
wrap(_lib, "call", "ctypes library calls (all, incl. the kernel launches inside)")
wrap(torch, "empty", "torch.empty")
wrap(graph_ops, "kagin_regression_forward", "kagin_regression_forward (whole forward)")
opt = Adam(m.parameters(), lr=1e-3)
steps = 0
for rep in range(13):
    if rep == 3:
        acc.clear(); cnt.clear(); steps = 0
    for d in batches:
        opt.zero_grad()
        loss = ops.l1_loss(m(d).squeeze(), d.y.squeeze())
        loss.backward()
        opt.step()
        steps += 1
torch.cuda.synchronize()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:70s} {v / steps * 1e6:8.1f} us/step  ({cnt[k] / steps:.1f} calls/step)")
