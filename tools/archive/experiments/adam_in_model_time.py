"""Why does harness.Adam.step() take ~85 us inside the config-4 step and 23 us on bare tensors?  Times its pieces in place."""
import os, sys, time, runpy
sys.argv = [sys.argv[0], "4h"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs_sweep.py"))
import torch
from kagnn_amd import ops
from kagnn_amd.harness import Adam
m, batches = ns["m"], ns["batches"]
opt = Adam(m.parameters(), lr=1e-3)
acc = {"grads list": 0.0, "checks": 0.0, "tables": 0.0, "library call": 0.0}
n = 0
for rep in range(13):
    for d in batches:
        opt.zero_grad()
        loss = ops.l1_loss(m(d).squeeze(), d.y.squeeze())
        loss.backward()
        t0 = time.perf_counter()
        gs = [p.grad for p in opt.params]
        t1 = time.perf_counter()
        ok = all(g is not None and g.dtype is torch.float32 and g.is_contiguous() for g in gs)
        t2 = time.perf_counter()
        tab = opt._VP(*[g.data_ptr() for g in gs])
        t3 = time.perf_counter()
        opt.steps += 1
        ops._call("kagnn_adam_step", len(gs), opt._p_tab, tab, opt._m_tab, opt._v_tab, opt._n_tab, opt.lr, opt.betas[0], opt.betas[1], opt.eps,
                  opt.weight_decay, opt.steps, ops._stream())
        t4 = time.perf_counter()
        if rep >= 3:
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[k] += v
            n += 1
torch.cuda.synchronize()
print({k: round(v / n * 1e6, 1) for k, v in acc.items()}, "all fast-path:", ok)
