"""GPU box: per-parameter gradient error of the ZINC-shaped regression models (fixture G8b) against the fp64 oracle, both precision
modes, next to the error of the fixture's own fp32 (reference arithmetic) gradients.  PYTHONPATH=. python tools/zinc_grad_errors.py"""
import numpy as np, torch, kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
T = lambda a, d="cpu": torch.from_numpy(np.asarray(a)).to(d)
z = np.load("tests/golden/g8b_zinc_batch.npz")
DEV = "cuda:0"
class D: pass
for kind in ("kan", "fastkan"):
    pre = f"{kind}.state."
    st = {k[len(pre):]: (T(z[k]).double().requires_grad_(True) if z[k].dtype.kind == "f" else T(z[k])) for k in z.files if k.startswith(pre)}
    p64 = orc.graph_regression_forward(T(z["x"]), T(z["edge_index"]), T(z["edge_attr"]), T(z["batch"]), 256, st, kind, 3)
    (p64.squeeze() - T(z[f"{kind}.y"]).double()).abs().mean().backward()
    rows = {}
    for mode, mname in ((ops.PREC_FP32, "fp32"), (ops.PREC_SPLIT, "split")):
        d = D(); d.x, d.edge_index, d.batch = T(z["x"], DEV), T(z["edge_index"], DEV), T(z["batch"], DEV)
        d.edge_attr, d.num_graphs = T(z["edge_attr"], DEV), 256
        m = kagnn_amd.KAGINRegression(1, 1, 3, 32, 2, 4, 3, 1, 0.0, True) if kind == "kan" else kagnn_amd.FASTKAGINRegression(1, 1, 3, 32, 2, 6, 1, 0.0, True)
        m.atom_encoder = kagnn_amd.graph_models.AtomEncoder(32, [21])
        m.bond_encoder.bond_embedding_list = torch.nn.ModuleList([torch.nn.Embedding(4, 32)])
        m.load_state_dict({k[len(pre):]: T(z[k], DEV) for k in z.files if k.startswith(pre)})
        m = m.to(DEV).train()
        for mod in m.modules():
            if hasattr(mod, "precision"): mod.precision = mode
        pred = m(d)
        torch.nn.L1Loss()(pred.squeeze(), T(z[f"{kind}.y"], DEV)).backward()
        for name, p in m.named_parameters():
            if p.requires_grad and st[name].grad is not None:
                w = st[name].grad; sc = max(1.0, float(w.abs().max()))
                rows.setdefault(name, {})[mname] = float((p.grad.double().cpu() - w).abs().max()) / sc
    for name in rows:
        w = st[name].grad; sc = max(1.0, float(w.abs().max()))
        rows[name]["ref32"] = float((T(z[f"{kind}.grad.{name}"]).double() - w).abs().max()) / sc
    worst = {k: max(r[k] for r in rows.values()) for k in ("fp32", "split", "ref32")}
    print(kind, "worst:", {k: f"{v:.2e}" for k, v in worst.items()})
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["split"])[:6]:
        print("   ", name, {k: f"{v:.1e}" for k, v in r.items()})
