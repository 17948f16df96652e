"""Host issue time of the pieces of KAGINRegression.forward on the config-4 batches (no synchronisation inside the loop)."""
import os, sys, time, runpy
sys.argv = [sys.argv[0], "4h"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs_sweep.py"))
import torch
from kagnn_amd import ops
from kagnn_amd.harness import Adam
m, batches = ns["m"], ns["batches"]
opt = Adam(m.parameters(), lr=1e-3)
names = ["atom_encoder", "bond_encoder", "graph_index", "message_passing (stack node)", "pool", "read-out KAN", "loss", "backward", "optimizer"]
acc = [0.0] * len(names)
for rep in range(13):
    for d in batches:
        opt.zero_grad()
        t = [time.perf_counter()]
        x = m.atom_encoder(d.x); t.append(time.perf_counter())
        ea = m.bond_encoder(d.edge_attr.unsqueeze(1) if d.edge_attr.dim() == 1 else d.edge_attr); t.append(time.perf_counter())
        g = ops.graph_index(d.edge_index, x.size(0), cache=False); t.append(time.perf_counter())
        x = m._message_passing(x, g, ea); t.append(time.perf_counter())
        p = m._pool(x, d); t.append(time.perf_counter())
        out = m.kan(p); t.append(time.perf_counter())
        loss = ops.l1_loss(out.squeeze(), d.y.squeeze()); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        if rep >= 3:
            for k in range(len(names)):
                acc[k] += t[k + 1] - t[k]
torch.cuda.synchronize()
n = 10 * len(batches)
print({k: round(v / n * 1e6, 1) for k, v in zip(names, acc)}, "sum", round(sum(acc) / n * 1e6, 1))
