#!/usr/bin/env python3
"""Random graph-regression models and mini-batches (hidden 8..64, 2..5 GINE convolutions, chains of 1..3 KANLinears, grid 3..5,
1..300 graphs of 1..40 nodes with 0..3x as many edges incl. self loops / duplicates / isolated nodes, 1..3 atom and 1..2 bond feature
columns with tables of 2..300 rows, 1 or 3 targets): KAGINRegression's default path -- the whole forward as ONE tape node
(graph_ops._KaginModelFn: stack call, embedding / pool / read-out calls, deferred slab reductions, merged norm kernels) -- against
(a) the same model run as its five kinds of nodes, and with one node per convolution instead of the stack node (all bit for bit:
prediction, loss, every gradient, running statistics) and (b) the
fp64 oracle, with the per-operation composition (aggregate_gine -> pack -> KANLinear -> BatchNorm ...) as the yardstick, through
ops.l1_loss.  usage: python tools/fuzz_graph_models.py [cases] [seed]"""
import copy, os, random, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import kagnn_amd
from kagnn_amd import graph_ops, ops
from oracle import kan_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
bad = nodes_run = calls_run = kinks = 0
for case in range(cases):
    H = rng.choice([8, 16, 24, 32, 40, 48, 64])
    nconv, hl, G = rng.choice([2, 3, 4, 5]), rng.choice([1, 2, 3]), rng.choice([3, 4, 5])
    B = rng.choice([1, 2, 7, 64, 300])
    edgeless = rng.random() < 0.08                   # a batch of single atoms: no edge at all
    targets = rng.choice([1, 1, 3])
    acols, bcols = rng.choice([1, 2, 3]), rng.choice([1, 2])
    adims = [rng.choice([2, 21, 119, 300]) for _ in range(acols)]
    bdims = [rng.choice([2, 4, 6]) for _ in range(bcols)]
    gen = torch.Generator().manual_seed(1000 + case)
    sizes = torch.randint(1, 41, (B,), generator=gen)
    n = int(sizes.sum()); off = torch.cumsum(sizes, 0) - sizes
    src, dst, batch = [], [], []
    for b in range(B):
        nb = int(sizes[b]); eb = 0 if edgeless else int(torch.randint(0, 3 * nb + 1, (1,), generator=gen))
        src.append(torch.randint(0, nb, (eb,), generator=gen) + off[b]); dst.append(torch.randint(0, nb, (eb,), generator=gen) + off[b])
        batch.append(torch.full((nb,), b))
    e = sum(len(s_) for s_ in src)
    d = SimpleNamespace(x=torch.stack([torch.randint(0, v, (n,), generator=gen) for v in adims], 1).to(DEV),
                        edge_index=torch.stack([torch.cat(src), torch.cat(dst)]).to(DEV),
                        edge_attr=torch.stack([torch.randint(0, v, (e,), generator=gen) for v in bdims], 1).to(DEV),
                        batch=torch.cat(batch).to(DEV), num_graphs=B)
    y = torch.randn(B, targets, generator=gen).to(DEV)
    torch.manual_seed(case)
    m0 = kagnn_amd.KAGINRegression(1, 1, nconv, H, hl, G, 3, targets, 0.0, True)
    m0.atom_encoder = kagnn_amd.graph_models.AtomEncoder(H, adims)
    m0.bond_encoder.bond_embedding_list = torch.nn.ModuleList([torch.nn.Embedding(v, H) for v in bdims])
    with torch.no_grad():                            # trained-looking norms (with beta = 0 the pooled sum of a ONE-graph batch is
        for bn in m0.bn:                             # exactly 0 in exact arithmetic: the prediction would be rounding noise)
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0.0, 0.3)
    m0 = m0.to(DEV).train()
    label = f"case {case}: hidden {H} convs {nconv} chain {hl} grid {G} graphs {B} nodes {n} edges {e} atom tables {adims} bond tables {bdims} targets {targets}"
    try:
        res = {}
        for how in ("call", "model", "nodes", "layer", "ops"):
            graph_ops._GINE_MODEL_CALL = how == "call"            # (round 6) the whole model as ONE library call each way
            graph_ops._GINE_MODEL_NODE = how in ("call", "model")
            graph_ops._GINE_STACK_ABI = how in ("call", "model", "nodes")
            graph_ops._GINE_LAYER_ABI = how != "ops"
            m = copy.deepcopy(m0)
            pred = m(d)
            nodes_run += how == "model" and type(pred.grad_fn).__name__ == "_KaginModelFnBackward"
            calls_run += how == "call" and type(pred.grad_fn).__name__ == "_KaginModelCallFnBackward"
            loss = ops.l1_loss(pred, y)
            loss.backward()
            res[how] = ([pred.detach().clone(), loss.detach().clone()] + [p.grad.clone() for p in m.parameters()]
                        + [b_.clone() for b_ in m.buffers() if b_.dtype.is_floating_point])
        for k, (a, b) in enumerate(zip(res["call"], res["model"])):
            if not torch.equal(a, b):
                raise AssertionError(f"tensor {k}: the one-call form differs from the one-node form by {float((a - b).abs().max()):.3e}")
        copy.deepcopy(m)                                 # (a model that has run stays deep-copyable: its call template is plain bytes)
        for k, (a, b, c) in enumerate(zip(res["model"], res["nodes"], res["layer"])):
            if not torch.equal(a, b):
                raise AssertionError(f"tensor {k}: the one-node form differs from the five-node form by {float((a - b).abs().max()):.3e}")
            if not torch.equal(b, c):
                raise AssertionError(f"tensor {k}: the stack node differs from one node per convolution by {float((b - c).abs().max()):.3e}")
        # referee: the fp64 oracle restatement of the model (oracle/kan_oracle.py::graph_regression_forward); yardstick: the error of
        # the per-operation composition against it.  relu(x_j + e_ij) has a kink -- a message element within rounding of zero flips
        # the upstream gradient by ~1e-3 of its scale, and which side an fp32 pipeline lands on is arithmetic-order luck (the
        # fused and composed forms differ in how the batch statistics are rounded) -- so a tensor may be off by 1e-4 of its largest
        # element, or by four times what the composition is off, or by four times what the EXACT gradient moves when every parameter
        # moves by half an fp32 ulp (`moved` below: the discontinuity measured where it is); tensors that are pure cancellation
        # (everything upstream of the pool in a one-graph batch; norms over two distinct rows) are scaled by 3 % of the model's
        # largest gradient.
        st = {k: (v.detach().cpu().double().requires_grad_(True) if v.dtype.is_floating_point else v.cpu()) for k, v in m0.state_dict().items()}
        p64 = orc.graph_regression_forward(d.x.cpu(), d.edge_index.cpu(), d.edge_attr.cpu(), d.batch.cpu(), B, st, "kan", nconv)
        l64 = (p64 - y.cpu().double()).abs().mean()
        l64.backward()
        names = [k for k, _ in m0.named_parameters()]
        want = [p64.detach(), l64.detach()] + [st[k].grad if st[k].grad is not None else torch.zeros_like(st[k]) for k in names]
        # how discontinuous is the exact gradient HERE at fp32 resolution?  The same oracle on parameters moved by half an fp32 ulp
        pg = torch.Generator().manual_seed(7 + case)
        st2 = {k: ((v.detach() * (1.0 + 6e-8 * torch.randn(v.shape, generator=pg, dtype=torch.float64))).requires_grad_(True)
                   if v.dtype.is_floating_point and k in names else v.detach() if v.dtype.is_floating_point else v) for k, v in st.items()}
        q64 = orc.graph_regression_forward(d.x.cpu(), d.edge_index.cpu(), d.edge_attr.cpu(), d.batch.cpu(), B, st2, "kan", nconv)
        m64 = (q64 - y.cpu().double()).abs().mean()
        m64.backward()
        moved = [q64.detach(), m64.detach()] + [st2[k].grad if st2[k].grad is not None else torch.zeros_like(st2[k]) for k in names]
        G = max(float(w.abs().max()) for w in want[2:])
        worst = wf = wo = 0.0
        kinked = False
        for k, w in enumerate(want):
            a, b = res["model"][k].double().cpu(), res["ops"][k].double().cpu()
            scale = max(float(w.abs().max()), 3e-2 * G if k >= 2 else 0.0)
            ef, eo = float((a - w).abs().max()) / scale, float((b - w).abs().max()) / scale
            ep = float((moved[k] - w).abs().max()) / scale
            bound = max(1e-4 if n >= 8 else 1e-3, 4.0 * eo, 4.0 * ep)      # (batch statistics over < 8 rows: rstd amplifies every rounding)
            if ef > bound and e > 0 and ef <= 2e-3:
                # a relu kink that only THIS pipeline's rounding crossed (2M message elements of unit scale: one of them lies
                # within an fp32 ulp of zero in about one pipeline out of five; seen on the composition as often as on the fused
                # form, 1.5e-3 at most).  Counted, and bounded over the whole run below: a systematic error would hit every case.
                kinked = True
                continue
            worst, wf, wo = max(worst, ef / bound), max(wf, ef), max(wo, eo)
            if not ef <= bound:
                raise AssertionError(f"tensor {k} ({'pred' if k == 0 else 'loss' if k == 1 else names[k - 2]}): fused form {ef:.2e} from the fp64 oracle, "
                                     f"composition {eo:.2e} (scale {scale:.2e}, largest gradient {G:.2e})")
        kinks += kinked
        print(f"ok   {label} worst {worst:.3f} of tolerance (largest error vs fp64: fused {wf:.1e}, composition {wo:.1e})"
              + ("  [one relu kink crossed: <= 2e-3]" if kinked else ""), flush=True)
    except Exception as ex:
        bad += 1
        print("FAIL", label, "->", str(ex)[:300], flush=True)
    finally:
        graph_ops._GINE_MODEL_CALL = graph_ops._GINE_MODEL_NODE = graph_ops._GINE_STACK_ABI = graph_ops._GINE_LAYER_ABI = True
if kinks > max(1, cases // 20):
    bad += 1
    print(f"FAIL {kinks} of {cases} cases needed the relu-kink allowance: that is not chance any more")
print(f"failures: {bad}   (cases that ran as one tape node: {nodes_run} of {cases}, as one library call each way: {calls_run}; relu-kink allowance used by {kinks})")
sys.exit(1 if bad else 0)
