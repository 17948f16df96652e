"""Round-2 GPU parity tests: the model / conv classes VERDICT r01 found untested against an oracle, BASELINE.json's
configs at their real shapes, determinism, and the precision report.

* G11 / G12 / G4b / G5b: reference-made fixtures (tests/golden/make_golden.py).
* Cora shape (config 1: N=2708, E=10 556, 1433 -> 32, grid 5, KAN-GCN, 2 layers) and ogbn-arxiv shape (config 2:
  N=169 343, E=1 166 243, 128 -> 64, grid 5, KAN-GIN, 3 layers; config 5: FastKAN hidden 256): whole-model logits and
  EVERY gradient against ``oracle.node_model_forward`` in fp64 (the restatement of models.py:192-203,246-257 that
  tests/test_oracle_golden.py pins to the reference-made G9 / G11 logits).
* config 3's layer (hidden 128, grid 8) at full size: sampled rows vs the oracle + additivity.
"""
import os
import re

import numpy as np
import pytest
import torch

import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
from helpers import (FK_KEYS, KAN_KEYS, T, TOL, assert_close, oracle_kan_linear_fwd_bwd, oracle_node_model_fwd_bwd,
                     prenorm_bias_noise)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = [ops.PREC_FP32, ops.PREC_SPLIT]
MODE_IDS = ["fp32", "split"]


def _set_precision(module, mode):
    for m in module.modules():
        if hasattr(m, "precision"):
            m.precision = mode


# ------------------------------------------------------------------ G4b / G5b / G12: reference-made fixtures
@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_fastkan_wide_layers_golden(golden, mode):
    z = golden("g4b_fastkan_wide")
    for i in range(2):
        fi, fo, ng = [int(v) for v in z[f"shape_{i}"]]
        tag = f"fk_{fi}_{fo}_{ng}"
        layer = kagnn_amd.FastKANLayer(fi, fo, num_grids=ng)
        layer.load_state_dict({n: T(z[f"{tag}.{n}"]) for n in FK_KEYS})
        layer = layer.to(DEV)
        layer.precision = mode
        x = T(z[f"{tag}.x"], DEV).requires_grad_(True)
        y = layer(x)
        y.backward(T(z[f"{tag}.gy"], DEV))
        assert_close(y, z[f"{tag}.y"], what=tag + ".y")
        assert_close(x.grad, z[f"{tag}.gx"], what=tag + ".gx")
        for name, p in layer.named_parameters():
            if p.requires_grad:
                assert_close(p.grad, z[f"{tag}.grad.{name}"], what=f"{tag}.grad.{name}")


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_gin_layers_on_10k_node_powerlaw_graph_golden(golden, mode):
    z = golden("g5b_gin_plaw10k")
    ei = T(z["edge_index"], DEV)
    for tag, conv in (("kan", kagnn_amd.GIKANLayer(8, 12, grid_size=5, spline_order=3, hidden_dim=12, nb_layers=2)),
                      ("fastkan", kagnn_amd.GIFASTKANLayer(8, 12, grid_size=4, hidden_dim=12, nb_layers=2))):
        conv.nn.load_state_dict({n[len(tag) + 1:]: T(z[n]) for n in z.files if n.startswith(tag + ".layers.")})
        conv = conv.to(DEV)
        _set_precision(conv, mode)
        x = T(z[f"{tag}.x"], DEV).requires_grad_(True)
        y = conv(x, ei)
        y.backward(T(z[f"{tag}.gy"], DEV))
        assert_close(y, z[f"{tag}.y"], what=f"plaw10k.{tag}.y")
        assert_close(x.grad, z[f"{tag}.gx"], what=f"plaw10k.{tag}.gx")
        for name, p in conv.nn.named_parameters():
            if p.requires_grad:
                assert_close(p.grad, z[f"{tag}.grad.{name}"], what=f"plaw10k.{tag}.grad.{name}")


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_fastkan_gcn_conv_golden(golden, mode):
    """FASTKAGCNConv (models.py:68-74) like G6: reference FastKANLayer inside the restated gcn_norm / aggregate"""
    z, g7 = golden("g12_fastkan_gcn"), golden("g7_csr")
    for g in ("small", "plaw"):
        pre = f"{g}.fgcn"
        ei = T(g7[f"{g}.edge_index"], DEV)
        conv = kagnn_amd.FASTKAGCNConv(16, 24, grid_size=4)
        conv.lin.load_state_dict({k: T(z[f"{pre}.lin.{k}"]) for k in FK_KEYS})
        conv.bias.data.copy_(T(z[f"{pre}.bias"]))
        conv = conv.to(DEV)
        _set_precision(conv, mode)
        x = T(z[f"{pre}.x"], DEV).requires_grad_(True)
        y = conv(x, ei)
        y.backward(T(z[f"{pre}.gy"], DEV))
        assert_close(y, z[f"{pre}.y"], what=pre + ".y")
        assert_close(x.grad, z[f"{pre}.gx"], what=pre + ".gx")
        assert_close(conv.bias.grad, z[f"{pre}.grad.bias"], what=pre + ".g_bias")
        for name, p in conv.lin.named_parameters():
            if p.requires_grad:
                assert_close(p.grad, z[f"{pre}.grad.lin.{name}"], what=f"{pre}.grad.lin.{name}")


# ------------------------------------------------------------------ G11: GFASTKAN_Nodes + harness
@pytest.mark.parametrize("kind", ["gin", "gcn"])
def test_gfastkan_nodes_harness_step_golden(golden, kind):
    """GFASTKAN_Nodes (config 5's model class, models.py:205-257) loaded from a reference-made state_dict: first
    forward, first-step gradients, both losses of the reference timing loop and the logits after two Adam steps."""
    from kagnn_amd.harness import time_model
    z = golden("g11_fastkan_harness")
    n, e, fin, hid, classes, ng = [int(v) for v in z["cfg"]]
    model = kagnn_amd.GFASTKAN_Nodes(kind, 2, fin, hid, classes, skip=True, grid_size=ng, hidden_layers=2)
    pre = f"{kind}.init."
    init = {n_[len(pre):]: T(z[n_]) for n_ in z.files if n_.startswith(pre)}
    assert set(init) == set(model.state_dict())                  # the reference's state_dict keys, exactly
    model.load_state_dict(init)
    model = model.to(DEV).train()
    x, ei, y, mask = T(z["x"], DEV), T(z["edge_index"], DEV), T(z["y"], DEV), T(z["mask"], DEV)
    logits = model(x, ei)
    assert_close(logits, z[f"{kind}.logits0"], what=f"g11.{kind}.logits0")
    loss = torch.nn.CrossEntropyLoss()(torch.softmax(logits, dim=1)[mask], y[mask])
    loss.backward()
    wants = {n_[len(kind) + 7:]: z[n_] for n_ in z.files if n_.startswith(f"{kind}.grad0.")}
    for name, p in model.named_parameters():
        if p.requires_grad:
            # (a bias in front of BatchNorm: the fixture holds the reference's own fp32 rounding noise around an exact zero)
            assert_close(p.grad, wants[name], what=f"g11.{kind}.grad0.{name}", noise=prenorm_bias_noise(name, wants))
    model.zero_grad()
    model.load_state_dict({k: v.to(DEV) for k, v in init.items()})
    _, losses = time_model(model, x, ei, y, mask, nb_epochs=2, warmup=0)
    assert abs(losses[0] - float(z[f"{kind}.losses"][0])) < 1e-5
    assert abs(losses[1] - float(z[f"{kind}.losses"][1])) < 1e-3       # one Adam step (sign-like update) in between
    assert_close(model(x, ei), z[f"{kind}.logits2"], 5e-3, what=f"g11.{kind}.logits after 2 Adam steps")


# ------------------------------------------------------------------ BASELINE configs at their real shapes
_ORACLE_CACHE: dict = {}


def _oracle_case(key, model, x, ei, gout, arch, kind, layers, spline_order, chunk, dtype):
    """(state, logits, gx, parameter gradients) of the fp64 oracle for a model case that more than one test checks (the
    arxiv-shaped KAN-GIN model: fp32 storage and bf16 gather operands): computed once per session -- it is the minutes of
    this suite on a slow host -- and the SAME initial state goes into every model of the case"""
    hit = _ORACLE_CACHE.get(key)
    if hit is None:
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        hit = (state,) + tuple(oracle_node_model_fwd_bwd(x, ei, state, gout, arch, kind, layers, spline_order, chunk, dtype))
        _ORACLE_CACHE[key] = hit
    model.load_state_dict(hit[0])
    return hit[1], hit[2], hit[3]


def _model_vs_oracle(model, arch, kind, layers, x, ei, gout, label, tol, chunk=None, spline_order=3, ei_dev=None,
                     dtype=torch.float64, cache_key=None, expect_fused_norm=None, mutation_guard=False):
    """``expect_fused_norm``: how many convolutions must have run as the conv + BatchNorm tape node
    (``kagnn_gin_kan_layer_bwd_bn`` -> ``kan_split_dx_kernel<..., BNB>``), i.e. the DEFAULT training path of the 64-wide
    GIN models (module hooks used to switch it off here, so that kernel only met itself)."""
    if cache_key is not None:
        want, gx_want, g_want = _oracle_case(cache_key, model, x, ei, gout, arch, kind, layers, spline_order, chunk, dtype)
    else:
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        want, gx_want, g_want = oracle_node_model_fwd_bwd(x, ei, state, gout, arch, kind, layers, spline_order, chunk, dtype)
    model = model.to(DEV).train()
    l1 = {}                                           # per conv: column-wise sum over the nodes of |d loss / d conv output|
    # Module hooks take a convolution off its fused tape nodes (models.conv_bn_dropout), so they are installed ONLY where
    # the check below needs them: the GCN / GAT flavours, whose conv bias sits in front of the norm (never fused anyway).
    biased = any(n_.startswith("convs.") and n_.count(".") == 2 and n_.endswith(".bias") for n_, _ in model.named_parameters())

    def watch(i):
        def fwd_hook(_m, _inp, out):
            out.register_hook(lambda g: l1.__setitem__(i, float(g.abs().sum(0).max())))
        return fwd_hook
    hooks = [conv.register_forward_hook(watch(i)) for i, conv in enumerate(model.convs)] if biased else []
    xd = x.to(DEV).requires_grad_(True)
    timer = ops.EntryPointTimer()
    ops.set_timer(timer)
    try:
        out = model(xd, ei.to(DEV) if ei_dev is None else ei_dev)
        out.backward(gout.to(DEV))
    finally:
        ops.set_timer(None)
    for h in hooks:
        h.remove()
    calls = {}
    for name, _a, _b in timer.records:
        calls[name] = calls.get(name, 0) + 1
    if expect_fused_norm is not None:
        # (round 4: the same node calls ..._bwd_bn_sums when the norms' backward statistics travel with the gradients)
        assert calls.get("kagnn_gin_kan_layer_bwd_bn", 0) + calls.get("kagnn_gin_kan_layer_bwd_bn_sums", 0) == expect_fused_norm, calls
    assert_close(out, want, tol, what=f"{label}.logits")
    assert_close(xd.grad, gx_want, tol, what=f"{label}.gx")
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        parts = name.split(".")
        if parts[0] == "convs" and parts[-1] == "bias" and len(parts) == 3:
            # a bias in front of BatchNorm (training mode) has an identically ZERO gradient -- the batch mean is
            # subtracted -- so what fp32 computes (here and in the reference) is the rounding noise of a cancelling
            # sum over the nodes: bound it by 1e-5 of the sum of magnitudes instead of comparing noise with 0
            assert float(g_want[name].abs().max()) <= 1e-9 * max(1.0, l1[int(parts[1])])
            assert float(p.grad.abs().max()) <= 1e-5 * l1[int(parts[1])], (name, float(p.grad.abs().max()), l1)
            continue
        assert_close(p.grad, g_want[name], tol, what=f"{label}.grad.{name}", noise=prenorm_bias_noise(name, g_want))
    if mutation_guard:
        # the assertions above must be ABLE to fail on these tensors (gout / n makes them ~1e-5 .. 1e-2): zeros, and a
        # 1e-3 relative perturbation (10x the tolerance), for the input gradient and the smallest family of parameter gradients
        from helpers import must_fail
        must_fail(torch.zeros_like(xd.grad), gx_want, tol, what=f"{label}.gx")
        must_fail(xd.grad * (1.0 + 1e-3), gx_want, tol, what=f"{label}.gx")
        some = [n_ for n_, p_ in model.named_parameters() if p_.requires_grad and n_.endswith(("spline_scaler", "spline_linear.weight"))]
        for n_ in (some[0], some[-1]):
            g = dict(model.named_parameters())[n_].grad
            must_fail(torch.zeros_like(g), g_want[n_], tol, what=f"{label}.grad.{n_}")
            must_fail(g * (1.0 + 1e-3), g_want[n_], tol, what=f"{label}.grad.{n_}")
    return want


def _cora_like(seed=0):
    """Cora's shape (Planetoid: 2708 nodes, 10 556 directed edges = 5278 undirected pairs, 1433 bag-of-words
    features, 7 classes) with NormalizeFeatures-style rows: ~18 non-zeros per row, each row summing to 1."""
    g = torch.Generator().manual_seed(seed)
    n, pairs, fin = 2708, 5278, 1433
    a = torch.randint(0, n, (pairs,), generator=g)
    b = torch.randint(0, n, (pairs,), generator=g)
    ei = torch.cat([torch.stack([a, b]), torch.stack([b, a])], dim=1)
    x = (torch.rand(n, fin, generator=g) < 18.0 / fin).float()
    x[torch.arange(n), torch.randint(0, fin, (n,), generator=g)] = 1.0        # no empty row
    x = x / x.sum(1, keepdim=True)
    return ei, x


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_cora_shaped_kan_gcn_model_vs_oracle(mode):
    """BASELINE config 1's model on the GPU: GKAN_Nodes('gcn', 2 layers, 1433 -> 32, grid 5) -- KANLinear(1433 -> 32)
    is the split-K few-rows path, the read-out KANLinear(1497 -> 7)."""
    ei, x = _cora_like()
    torch.manual_seed(1)
    model = kagnn_amd.GKAN_Nodes("gcn", 2, 1433, 32, 7, skip=True, grid_size=5, spline_order=3, hidden_layers=2)
    _set_precision(model, mode)
    gout = torch.randn(2708, 7, generator=torch.Generator().manual_seed(2))
    # two conv layers + BatchNorm (divides by a batch std of ~1e-2 on these tiny activations) compound the layer
    # error: 1e-4 (the contract) instead of the layer-level 2e-5
    _model_vs_oracle(model, "kan", "gcn", 2, x, ei, gout, f"cora.{MODE_IDS[mode]}", 1e-4, chunk=512)


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_cora_shaped_kan_gin_model_default_path_vs_oracle(mode):
    """Cora's shape through GKAN_Nodes('gin', 2 layers, 1433 -> 32, grid 5), hook-free: both convolutions run as the conv +
    BatchNorm tape node (32 outputs: the other BNB instantiation of the input-gradient kernel; exact-fp32 mode: the
    stand-alone pass inside the same library call), small graph => concatenating read-out."""
    ei, x = _cora_like(seed=5)
    torch.manual_seed(12)
    model = kagnn_amd.GKAN_Nodes("gin", 2, 1433, 32, 7, skip=True, grid_size=5, spline_order=3, hidden_layers=2)
    _set_precision(model, mode)
    gout = torch.randn(2708, 7, generator=torch.Generator().manual_seed(13))
    _model_vs_oracle(model, "kan", "gin", 2, x, ei, gout, f"cora.gin.{MODE_IDS[mode]}", 1e-4, chunk=512, expect_fused_norm=2)


def test_cora_shaped_model_on_the_sparse_adjacency_of_the_gcn_timing_branch():
    """time_model.py:70-80 hands GCNConv a torch sparse matrix (D^-1/2 (A+I) D^-1/2); torch_geometric then
    normalises it AGAIN with add_self_loops (+1 on the existing diagonal).  Model level (the node models
    used to crash on it), eager and HIP-graph harness."""
    from kagnn_amd.harness import time_model
    ei, x = _cora_like(seed=3)
    n = 2708
    a = torch.sparse_coo_tensor(ei, torch.ones(ei.size(1)), (n, n))
    a_hat = (a + torch.sparse_coo_tensor(torch.arange(n).repeat(2, 1), torch.ones(n), (n, n))).coalesce()
    d = torch.sparse.sum(a_hat, dim=1).to_dense().pow(-0.5)
    idx = a_hat.indices()
    adj = torch.sparse_coo_tensor(idx, d[idx[0]] * a_hat.values() * d[idx[1]], (n, n)).coalesce()     # sparse_diag @ A_hat @ sparse_diag
    torch.manual_seed(4)
    model = kagnn_amd.GKAN_Nodes("gcn", 2, 1433, 32, 7, skip=True, grid_size=4, spline_order=3)
    gout = torch.randn(n, 7, generator=torch.Generator().manual_seed(5))
    adj_dev = adj.to(DEV)
    _model_vs_oracle(model, "kan", "gcn", 2, x, adj, gout, "cora.sparse_adj", 1e-4, chunk=512, ei_dev=adj_dev)
    y = torch.randint(0, 7, (n,)).to(DEV)
    mask = (torch.rand(n) < 0.05).to(DEV)
    t_eager, l_eager = time_model(model, x.to(DEV), adj_dev, y, mask, nb_epochs=2, warmup=1)
    assert all(np.isfinite(l_eager))
    with pytest.raises(ValueError, match="sparse"):
        kagnn_amd.GKAN_Nodes("gin", 1, 1433, 8, 7).to(DEV)(x.to(DEV), adj_dev)


def _arxiv_like():
    n, e = 169_343, 1_166_243
    ei = orc.powerlaw_graph(n, e, seed=0)            # directed, not symmetrised (time_model.py never calls to_undirected)
    x = torch.randn(n, 128, generator=torch.Generator().manual_seed(0)) * 0.5
    return n, ei, x


def test_arxiv_shaped_kan_gin_model_vs_oracle():
    """BASELINE config 2's model at ogbn-arxiv's shape, default (split) precision: GKAN_Nodes('gin', 3 layers,
    128 -> 64, grid 5, 40 classes) -- logits, d/dx and every parameter gradient against the fp64 oracle (row-chunked
    with checkpointing: the dense bases of one KANLinear alone are 0.7 GB here)."""
    n, ei, x = _arxiv_like()
    torch.manual_seed(6)
    model = kagnn_amd.GKAN_Nodes("gin", 3, 128, 64, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2)
    gout = torch.randn(n, 40, generator=torch.Generator().manual_seed(7)) / n      # a mean-type loss gradient
    # three conv layers + BatchNorm compound the layer error; parameter gradients are sums over 169k rows
    # hook-free: all three convolutions run as the conv + BatchNorm tape node (kan_split_dx_kernel<..., BNB>), the skip
    # gradients travel through ops.SkipGradient and the read-out is one launch over the four blocks -- the path bench.py's
    # secondary.model_step times
    _model_vs_oracle(model, "kan", "gin", 3, x, ei, gout, "arxiv.kan_gin", 1e-4, chunk=8192, cache_key="arxiv.kan_gin",
                     expect_fused_norm=3, mutation_guard=True)


def test_arxiv_shaped_fastkan_model_vs_oracle():
    """BASELINE config 5: GFASTKAN_Nodes('gin', 3 layers, hidden 256, default num_grids) at ogbn-arxiv's shape"""
    n, ei, x = _arxiv_like()
    torch.manual_seed(8)
    model = kagnn_amd.GFASTKAN_Nodes("gin", 3, 128, 256, 40, skip=True, grid_size=4, hidden_layers=2)
    gout = torch.randn(n, 40, generator=torch.Generator().manual_seed(9)) / n
    # (fp64 oracle: in fp32 -- the reference's own arithmetic -- the oracle itself is 5e-4 off at this depth and width)
    _model_vs_oracle(model, "fastkan", "gin", 3, x, ei, gout, "arxiv.fastkan_gin", 1e-4, chunk=32768, expect_fused_norm=0,
                     mutation_guard=True)


# ------------------------------------------------------------------ config 3's layer at full size (hidden 128, grid 8)
@pytest.mark.parametrize("mode", [ops.PREC_SPLIT, ops.PREC_FP32], ids=["split", "fp32"])
def test_fullsize_kanlinear_hidden128_grid8_samples_and_additivity(mode):
    N, F_, G = 1_000_000, 128, 8
    gen = torch.Generator().manual_seed(13)
    p = orc.init_kan_linear(F_, F_, G, 3, gen)
    layer = kagnn_amd.KANLinear(F_, F_, grid_size=G, spline_order=3)
    layer.load_state_dict(p)
    layer = layer.to(DEV)
    layer.precision = mode
    x = torch.randn(N, F_, generator=gen) * 0.6
    xs = x.to(DEV).requires_grad_(True)
    rows = torch.unique(torch.randint(0, N, (2048,), generator=gen))
    gy = torch.zeros(N, F_)
    gy[rows] = torch.randn(rows.numel(), F_, generator=gen)
    y = layer(xs)
    y.backward(gy.to(DEV))
    y64, gx64, g64 = oracle_kan_linear_fwd_bwd(x[rows], gy[rows], p, 3)
    assert_close(y.detach().cpu()[rows], y64, what="cfg3.y rows")
    assert_close(xs.grad.cpu()[rows], gx64, what="cfg3.gx rows")
    for k in ("base_weight", "spline_weight", "spline_scaler"):
        assert_close(getattr(layer, k).grad, g64[k], what="cfg3.g_" + k)      # gy is zero outside the sample
    off = torch.ones(N, dtype=torch.bool); off[rows] = False
    assert float(xs.grad[off.to(DEV)].abs().max()) == 0.0
    g2 = torch.randn(N, F_, generator=gen).to(DEV)
    half = N // 2 + 17

    def wgrad(gmat):
        for q in layer.parameters():
            q.grad = None
        layer(xs.detach()).backward(gmat)
        return layer.spline_weight.grad.clone(), layer.base_weight.grad.clone()
    full_s, full_b = wgrad(g2)
    ga = g2.clone(); ga[half:] = 0
    gb = g2.clone(); gb[:half] = 0
    a_s, a_b = wgrad(ga)
    b_s, b_b = wgrad(gb)
    assert_close(a_s + b_s, full_s, what="cfg3.dW additivity")
    assert_close(a_b + b_b, full_b, what="cfg3.dWb additivity")


def test_fullsize_gin_layer_hidden128_grid8_checksum():
    """config 3's conv layer on the 1M / 10M graph: column checksum of the aggregation in fp64 and the layer output
    on sampled destination rows against the oracle fed with the device's own aggregate (rows are independent after
    the aggregation)."""
    N, E, F_, G = 1_000_000, 10_000_000, 128, 8
    ei = orc.powerlaw_graph(N, E, seed=0)
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(N, F_, generator=gen) * 0.25
    conv = kagnn_amd.GIKANLayer(F_, F_, grid_size=G, spline_order=3, hidden_dim=F_, nb_layers=2)
    layers = [{k: v.detach().clone() for k, v in l.state_dict().items()} for l in conv.nn.layers]
    conv = conv.to(DEV)
    gi = ops.GraphIndex(ei.to(DEV), N)
    xd = x.to(DEV)
    h0 = ops.aggregate_sum(xd, gi, self_scale=1.0)
    want = xd.double().sum(0) + xd.double().index_select(0, ei[0].to(DEV)).sum(0)
    assert_close(h0.double().sum(0), want, 1e-6, what="cfg3.agg column checksum")
    y = conv(xd, gi)
    rows = torch.unique(torch.randint(0, N, (1024,), generator=gen))
    y64 = orc.kan_forward(h0[rows.to(DEV)].cpu().double(), [{k: v.double() for k, v in l.items()} for l in layers], 3)
    assert_close(y.detach().cpu()[rows], y64, what="cfg3.layer rows")


# ------------------------------------------------------------------ determinism (reference utils.py:25-28 seeds everything)
def test_aggregation_and_layer_are_bit_reproducible_with_hubs():
    n, e, f = 50_000, 600_000, 64
    ei = orc.powerlaw_graph(n, e, seed=4)
    gi = ops.GraphIndex(ei.to(DEV), n)
    assert gi.num_hub_seg > 0 and gi.num_hub_seg_t >= 0
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(1)).to(DEV)
    for transposed in (False, True):
        a = ops._aggregate_raw(x, gi, transposed, 1.0, None, None, None, None, False)
        for _ in range(3):
            assert torch.equal(a, ops._aggregate_raw(x, gi, transposed, 1.0, None, None, None, None, False))
    # hub rows equal the plain edge-order sum to fp32 rounding of a different order, not bitwise -- but every run is
    deg = torch.bincount(ei[1], minlength=n)
    hub = int(deg.argmax())
    want = x[hub].double() + x[ei[0][ei[1] == hub].to(DEV)].double().sum(0)
    assert_close(ops.aggregate_sum(x, gi)[hub], want, what="hub row")
    conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2).to(DEV)
    gy = torch.randn(n, f, generator=torch.Generator().manual_seed(2)).to(DEV)

    def run():
        conv.zero_grad()
        xr = x.clone().requires_grad_(True)
        y = conv(xr, gi)
        y.backward(gy)
        return [y.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
    first = run()
    for _ in range(2):
        for a, b in zip(first, run()):
            assert torch.equal(a, b)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_modules_on_a_non_current_device():
    """kernels must launch on the operands' device, not on torch's current one"""
    torch.cuda.set_device(0)
    dev1 = torch.device("cuda:1")
    layer = kagnn_amd.KANLinear(16, 8).to(dev1)
    x = torch.randn(100, 16, device=dev1, requires_grad=True)
    y = layer(x)
    y.sum().backward()
    ref = kagnn_amd.KANLinear(16, 8)
    ref.load_state_dict(layer.state_dict())
    ref = ref.to(DEV)
    assert_close(y, ref(x.detach().to(DEV)), 1e-6, what="cuda:1 forward")
    with pytest.raises(RuntimeError, match="one device"):
        ops.kan_linear(x.detach().to(DEV), layer.base_weight, layer.spline_weight, layer.spline_scaler, layer._knots(), 5, 3)


# ------------------------------------------------------------------ skip-branch gradient added inside the convolution's backward
@pytest.mark.parametrize("f", [8, 12, 64, 128, 50])
def test_aggregation_with_an_addend_is_the_aggregation_plus_the_addend_bitwise(f):
    """kagnn_aggregate_sum_add: the addend rides in the epilogue (hub rows: after the segment fold) -- exactly the bits of the
    aggregation followed by a separate addition; edge-slot, row and generic kernels, hubs included"""
    n, e = 6000, 90000
    ei = orc.powerlaw_graph(n, e, seed=11)
    g = ops.GraphIndex(torch.cat([ei, ei.flip(0)], dim=1).to(DEV), n)        # (both directions: hubs on either side)
    assert g.num_hub_seg > 0 and g.num_hub_seg_t > 0
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(1)).to(DEV)
    add = torch.randn(n, f + 4, generator=torch.Generator().manual_seed(2)).to(DEV)[:, :f]       # (a strided addend)
    for transposed in (False, True):
        want = ops._aggregate_raw(x, g, transposed, 1.25, None, None, None, None, False) + add
        got = ops._aggregate_raw(x, g, transposed, 1.25, None, None, None, None, False, addend=add)
        assert torch.equal(got, want)
    dis = g.gcn_dis
    bias = torch.randn(f, device=DEV)
    want = ops._aggregate_raw(x, g, False, 1.0, None, dis, dis, bias, True) + add
    got = ops._aggregate_raw(x, g, False, 1.0, None, dis, dis, bias, True, addend=add)
    assert torch.equal(got, want)


@pytest.mark.parametrize("dropout", [0.0, 0.3])
def test_skip_gradient_handed_to_the_next_convolution_gives_the_tape_sums_bits(monkeypatch, dropout):
    """GKAN_Nodes with skip connections on a large-graph code path: the read-out's gradient of h_l travels to conv l+1's
    backward outside the tape (ops.SkipGradient, added in the transposed aggregation's epilogue).  Same bits as letting
    autograd sum the two gradients -- every parameter gradient and the input gradient"""
    from kagnn_amd import models as M
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS", 0)
    monkeypatch.setattr(M, "_LAZY_NORM", False)        # (the composed node of the third variant cannot fold the norms: compare like with like;
                                                       # the folded form has its own test in test_gpu_epilogue.py)
    n, e, f = 9000, 80000, 64
    g = ops.GraphIndex(orc.powerlaw_graph(n, e, seed=4).to(DEV), n)
    x = (torch.randn(n, f, generator=torch.Generator().manual_seed(3)) * 0.3).to(DEV)
    y = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(5)).to(DEV)
    torch.manual_seed(9)
    model = kagnn_amd.GKAN_Nodes("gin", 3, f, f, 10, skip=True, grid_size=5, spline_order=3, hidden_layers=2, dropout=dropout).to(DEV)
    res = []
    for carry, abi in ((True, True), (False, True), (True, False)):       # (the last: the convolution node composed from the per-op entry points)
        monkeypatch.setattr(M, "_SKIP_GRADIENT", carry)
        monkeypatch.setattr(ops, "_LAYER_ABI", abi)
        model.zero_grad()
        xr = x.clone().requires_grad_(True)
        torch.manual_seed(77)                      # the dropout masks
        out = model(xr, g)
        assert type(out.grad_fn).__name__ == "_KANLinearPartsFnBackward"
        ops.softmax_cross_entropy(out, y).backward()
        res.append([out.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in model.parameters()])
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)
    # a second backward through a retained graph hands the gradients over again
    monkeypatch.setattr(M, "_SKIP_GRADIENT", True)
    monkeypatch.setattr(ops, "_LAYER_ABI", True)
    model.zero_grad()
    out = model(x, g)
    loss = ops.softmax_cross_entropy(out, y)
    loss.backward(retain_graph=True)
    first = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    loss.backward()
    for a, p in zip(first, model.parameters()):
        assert torch.equal(a, p.grad)
    # frozen convolutions: nothing is handed over, the read-out alone trains
    for c in model.convs:
        c.requires_grad_(False)
    for b in model.bns:
        b.requires_grad_(False)
    model.zero_grad()
    ops.softmax_cross_entropy(model(x, g), y).backward()
    assert all(p.grad is None for c in model.convs for p in c.parameters())
    assert model.lay_out.spline_weight.grad is not None


def test_node_model_folds_against_the_unfused_paths_over_random_configurations(monkeypatch):
    """GKAN_Nodes('gin') with the three round-3 folds on (read-out over column blocks in one launch, skip gradients handed to
    the convolutions, convolution + norm as one node) against the same model with all three off, over random widths / depths /
    row counts / dropout: same loss and gradients (the one-launch read-out adds its chunks in a different order than the
    per-block sums: 2e-5, everything else is bit-identical)"""
    import copy
    import random
    from kagnn_amd import models as M
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS", 0)
    monkeypatch.setattr(M, "_LAZY_NORM", False)        # (round 4's fold changes the rounding of every layer input: its own test
                                                       # over random configurations is in test_gpu_epilogue.py, at its own tolerance)
    rng = random.Random(20260929)
    for case in range(10):
        f_in = rng.choice([64, 128, 40, 64])
        hidden = rng.choice([64, 32, 16, 64, 128])
        mp, hl = rng.choice([1, 2, 3]), rng.choice([1, 2])
        n = rng.choice([700, 3001, 9000])
        p_drop = rng.choice([0.0, 0.0, 0.25])
        classes = rng.choice([7, 40])
        g = ops.GraphIndex(orc.powerlaw_graph(n, 8 * n, seed=case).to(DEV), n)
        x = (torch.randn(n, f_in, generator=torch.Generator().manual_seed(case)) * 0.4).to(DEV)
        y = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(case + 1)).to(DEV)
        torch.manual_seed(case)
        model0 = kagnn_amd.GKAN_Nodes("gin", mp, f_in, hidden, classes, skip=True, grid_size=5, spline_order=3, hidden_layers=hl,
                                      dropout=p_drop).to(DEV)
        res = []
        for on in (True, False):
            monkeypatch.setattr(M, "_SKIP_GRADIENT", on)
            monkeypatch.setattr(M, "_FUSED_NORM_BACKWARD", on)
            monkeypatch.setattr(ops, "_PARTS_ONE_LAUNCH", on)
            model = copy.deepcopy(model0)
            xr = x.clone().requires_grad_(case % 2 == 0)
            torch.manual_seed(1000 + case)                 # dropout masks
            loss = ops.softmax_cross_entropy(model(xr, g), y)
            loss.backward()
            res.append([loss.detach().clone()] + ([xr.grad.clone()] if xr.requires_grad else [])
                       + [p.grad.clone() for p in model.parameters()] + [b.clone() for b in model.buffers() if b.dtype.is_floating_point])
        label = f"case {case}: f_in {f_in} hidden {hidden} mp {mp} chain {hl} n {n} dropout {p_drop}"
        for k, (a, b) in enumerate(zip(*res)):
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1e-30), (label, k, float((a - b).abs().max()), scale)


# ------------------------------------------------------------------ convolution + BatchNorm1d as one tape node
@pytest.mark.parametrize("f_in,hidden,layers,x_grad", [(64, 64, 2, True), (64, 64, 2, False), (48, 32, 2, True), (64, 64, 1, True),
                                                        (64, 64, 1, False), (40, 16, 2, True), (64, 128, 2, True), (64, 64, 3, True)])
def test_conv_and_norm_as_one_tape_node_give_the_two_nodes_bits(monkeypatch, f_in, hidden, layers, x_grad):
    _conv_and_norm_node_case(monkeypatch, f_in, hidden, layers, x_grad, 7001, 60000)


@pytest.mark.parametrize("n,e", [(2, 3), (130, 900), (257, 2000), (3000, 1)])
def test_conv_and_norm_as_one_tape_node_on_few_rows(monkeypatch, n, e):
    """few rows: the input-gradient launch spreads its feature tiles over blockIdx.y (every block transforms the rows it
    loads, block 0 stores them); two rows is the norm's minimum"""
    _conv_and_norm_node_case(monkeypatch, 64, 64, 2, True, n, e)


def test_conv_and_norm_as_one_tape_node_in_the_exact_fp32_mode(monkeypatch):
    """exact-fp32 kernels: the library call runs the stand-alone normalisation backward, then the fp32 chain"""
    _conv_and_norm_node_case(monkeypatch, 64, 64, 2, True, 5000, 40000, mode=ops.PREC_FP32)


def _conv_and_norm_node_case(monkeypatch, f_in, hidden, layers, x_grad, n, e, mode=None):
    """models.conv_bn_dropout on a KAN-GIN convolution + training-mode BatchNorm1d: ONE tape node whose backward applies the
    norm's element-wise backward inside the last input-gradient kernel (kagnn_gin_kan_layer_bwd_bn; 32 / 64 outputs) or runs
    the stand-alone pass inside the same library call (other widths, single-layer chains without an input gradient).  Same
    bits as the convolution node followed by the norm node: output, running statistics, every gradient"""
    from kagnn_amd import models as M
    g = ops.GraphIndex(orc.powerlaw_graph(n, e, seed=6).to(DEV), n)
    x = (torch.randn(n, f_in, generator=torch.Generator().manual_seed(3)) * 0.4).to(DEV)
    gh = torch.randn(n, hidden, generator=torch.Generator().manual_seed(4)).to(DEV)
    import copy
    torch.manual_seed(5)
    conv0 = kagnn_amd.GIKANLayer(f_in, hidden, grid_size=5, spline_order=3, hidden_dim=hidden, nb_layers=layers).to(DEV)
    bn0 = kagnn_amd.BatchNorm1d(hidden).to(DEV)
    if mode is not None:
        _set_precision(conv0, mode)
    with torch.no_grad():
        bn0.weight.uniform_(0.5, 1.5); bn0.bias.uniform_(-0.3, 0.3)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(M, "_FUSED_NORM_BACKWARD", fused)
        conv, bn = copy.deepcopy(conv0), copy.deepcopy(bn0)      # (two constructions from one seed differ in the last bit: the init's least-squares fit)
        drop = torch.nn.Dropout(0.0)
        xr = x.clone().requires_grad_(x_grad)
        h = M.conv_bn_dropout(conv, bn, drop, xr, g)
        assert type(h.grad_fn).__name__ == ("_GinKanBnLayerFnBackward" if fused else "_BatchNormFnBackward")
        h.backward(gh)
        res.append([h.detach().clone(), bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()]
                   + ([xr.grad.clone()] if x_grad else []) + [p.grad.clone() for p in list(conv.parameters()) + list(bn.parameters())])
    names = (["h", "running_mean", "running_var", "num_batches_tracked"] + (["gx"] if x_grad else [])
             + [k for k, _ in conv.named_parameters()] + ["bn.weight", "bn.bias"])
    for k, a, b in zip(names, *res):
        assert torch.equal(a, b), f"{k}: max |diff| {(a.float() - b.float()).abs().max().item():.3e} of {b.float().abs().max().item():.3e}"


def test_conv_and_norm_node_steps_aside_for_hooks_dropout_and_eval(monkeypatch):
    from kagnn_amd import models as M
    n, e, f = 3000, 20000, 64
    g = ops.GraphIndex(orc.powerlaw_graph(n, e, seed=2).to(DEV), n)
    x = torch.randn(n, f, device=DEV).requires_grad_(True)
    conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2).to(DEV)
    bn = kagnn_amd.BatchNorm1d(f).to(DEV)
    name = lambda h: type(h.grad_fn).__name__
    assert name(M.conv_bn_dropout(conv, bn, torch.nn.Dropout(0.0), x, g)) == "_GinKanBnLayerFnBackward"
    assert name(M.conv_bn_dropout(conv, bn, torch.nn.Dropout(0.5), x, g)) == "_BatchNormFnBackward"       # active dropout rides in the norm's pass
    seen = []
    hk = conv.register_forward_hook(lambda m, i, o: seen.append(o.shape))
    assert name(M.conv_bn_dropout(conv, bn, torch.nn.Dropout(0.0), x, g)) == "_BatchNormFnBackward" and seen      # a hook wants the convolution's output
    hk.remove()

    class Custom(kagnn_amd.GIKANLayer):
        def forward(self, x, edge_index):
            return super().forward(x, edge_index) * 2.0
    sub = Custom(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2).to(DEV)
    assert name(M.conv_bn_dropout(sub, bn, torch.nn.Dropout(0.0), x, g)) == "_BatchNormFnBackward"      # an overriding forward is called
    bn.eval()
    assert name(M.conv_bn_dropout(conv, bn, torch.nn.Dropout(0.0), x, g)) == "_BatchNormFnBackward"
    bn.train()
    # a single-layer chain that widens (16 -> 64): its output is wider than every layer input -- two nodes, and they train
    wide = kagnn_amd.GIKANLayer(16, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=1).to(DEV)
    x16 = torch.randn(n, 16, device=DEV).requires_grad_(True)
    h = M.conv_bn_dropout(wide, bn, torch.nn.Dropout(0.0), x16, g)
    assert name(h) == "_BatchNormFnBackward"
    h.sum().backward()
    assert x16.grad is not None and torch.isfinite(x16.grad).all()
    before = int(bn.num_batches_tracked)
    M.conv_bn_dropout(conv, bn, torch.nn.Dropout(0.0), x, g)
    assert int(bn.num_batches_tracked) == before + 1


# ------------------------------------------------------------------ fused layer node + bf16 gather operands (config 2)
def test_fused_gin_kan_node_equals_the_composed_ops_bitwise(monkeypatch):
    """ops.gin_kan_layer (one tape node: aggregate + KAN chain, chain-packed weights) runs the same kernels in the same
    order as aggregate_sum -> KAN.forward: identical bits, forward and backward, both precision modes"""
    from kagnn_amd import models as M
    n, e, f = 20000, 150000, 64
    ei = orc.powerlaw_graph(n, e, seed=8)
    g = ops.GraphIndex(ei.to(DEV), n)
    x = (torch.randn(n, f, generator=torch.Generator().manual_seed(3)) * 0.3).to(DEV)
    gy = torch.randn(n, f, generator=torch.Generator().manual_seed(4)).to(DEV)
    for mode in MODES:
        torch.manual_seed(5)
        conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2).to(DEV)
        _set_precision(conv, mode)
        res = []
        for fused, abi in ((True, True), (True, False), (False, False)):     # library layer call / composed node / separate ops
            monkeypatch.setattr(M, "_FUSED_LAYER", fused)
            monkeypatch.setattr(ops, "_LAYER_ABI", abi)
            conv.zero_grad()
            xr = x.clone().requires_grad_(True)
            y = conv(xr, g)
            y.backward(gy)
            res.append([y.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in conv.parameters()])
        for other in res[1:]:
            for a, b in zip(res[0], other):
                assert torch.equal(a, b)
    # narrow first layers: the library call runs the aggregation INSIDE the first KANLinear's forward kernel
    # (kan_sparse_fwd_kernel<..., AGG>: north_star's producer -> consumer fusion; hub rows fixed up on a compact copy) --
    # still the same bits as aggregate_sum -> KAN.forward, for the three summation orders (in <= 8, <= 16, <= 32), with hubs
    # (this graph's top in-degree is ~500, threshold 96), an odd row count and a first layer not a multiple of 8 wide
    monkeypatch.setenv("KAGNN_FUSE_AGG", "1")           # (read per call; off by default: measured slower, api.hip)
    n3 = 20011
    ei3 = orc.powerlaw_graph(n3, 150000, seed=10)
    g3 = ops.GraphIndex(ei3.to(DEV), n3)
    assert g3.num_hub_seg > 0
    for fin, hid, fout in ((8, 64, 64), (16, 32, 64), (32, 64, 40), (24, 64, 64), (12, 64, 64)):
        torch.manual_seed(11)
        conv = kagnn_amd.GIKANLayer(fin, fout, grid_size=5, spline_order=3, hidden_dim=hid, nb_layers=2).to(DEV)
        x3 = (torch.randn(n3, fin, generator=torch.Generator().manual_seed(12)) * 0.4).to(DEV)
        gy3 = torch.randn(n3, fout, generator=torch.Generator().manual_seed(13)).to(DEV)
        res = []
        for fused, abi in ((True, True), (False, False)):
            monkeypatch.setattr(M, "_FUSED_LAYER", fused)
            monkeypatch.setattr(ops, "_LAYER_ABI", abi)
            conv.zero_grad()
            xr = x3.clone().requires_grad_(True)
            y = conv(xr, g3)
            y.backward(gy3)
            res.append([y.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in conv.parameters()])
        for a, b in zip(*res):
            assert torch.equal(a, b), (fin, hid, fout)
        if fin in (24, 12, 32):
            # ... and the fused kernel against the fp64 ORACLE, not only against its two-launch twin:
            # ragged first-layer widths, hub rows, odd row count
            lay = [{k: v.detach().cpu().double() for k, v in l.state_dict().items()} for l in conv.nn.layers]
            y64, gx64, g64 = orc.kan_gin_layer_fwd_bwd(x3.cpu().double(), ei3, lay, 3, gy3.cpu().double())
            assert_close(res[0][0], y64, what=f"fused aggregation->KANLinear kernel y in={fin}")
            assert_close(res[0][1], gx64, what=f"fused aggregation->KANLinear kernel gx in={fin}")
            for li, layer in enumerate(conv.nn.layers):
                for k in ("base_weight", "spline_weight", "spline_scaler"):
                    assert_close(getattr(layer, k).grad, g64[li][k], what=f"fused aggregation kernel L{li}.{k} in={fin}")
    monkeypatch.delenv("KAGNN_FUSE_AGG")
    # a chain the one-launch pack does not cover (3 layers, ragged widths, grid 8 => 11 coefficients) and an odd row count
    n2 = 7001
    ei2 = orc.powerlaw_graph(n2, 40000, seed=9)
    g2 = ops.GraphIndex(ei2.to(DEV), n2)
    conv = kagnn_amd.GIKANLayer(24, 40, grid_size=8, spline_order=3, hidden_dim=33, nb_layers=3).to(DEV)
    x2 = (torch.randn(n2, 24, generator=torch.Generator().manual_seed(6)) * 0.4).to(DEV)
    gy2 = torch.randn(n2, 40, generator=torch.Generator().manual_seed(7)).to(DEV)
    res = []
    for fused, abi in ((True, True), (False, False)):
        monkeypatch.setattr(M, "_FUSED_LAYER", fused)
        monkeypatch.setattr(ops, "_LAYER_ABI", abi)
        conv.zero_grad()
        xr = x2.clone().requires_grad_(True)
        y = conv(xr, g2)
        y.backward(gy2)
        res.append([y.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in conv.parameters()])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    layers = [{k: v.detach().cpu().double() for k, v in l.state_dict().items()} for l in conv.nn.layers]
    y64, gx64, _ = orc.kan_gin_layer_fwd_bwd(x2.cpu().double(), ei2, layers, 3, gy2.cpu().double())
    assert_close(res[0][0], y64, what="3-layer ragged chain y")
    assert_close(res[0][1], gx64, what="3-layer ragged chain gx")


def test_library_stage_timer_sees_the_kernels_inside_the_layer_calls():
    """kagnn_stage_timer_* (what bench.py reads its per-kernel times from on the product's one-call-per-convolution path):
    while enabled every stage INSIDE kagnn_gin_kan_layer_fwd / _bwd is bracketed by HIP events; `only` restricts the records
    to one stage; disabled (the default) nothing is recorded."""
    n, e, f = 20000, 150000, 64
    ei = orc.powerlaw_graph(n, e, seed=8)
    g = ops.GraphIndex(ei.to(DEV), n)
    torch.manual_seed(5)
    conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2).to(DEV)
    x = (torch.randn(n, f, generator=torch.Generator().manual_seed(3)) * 0.3).to(DEV).requires_grad_(True)
    gy = torch.randn(n, f, generator=torch.Generator().manual_seed(4)).to(DEV)

    def step():
        x.grad = None
        conv.zero_grad()
        conv(x, g).backward(gy)
    step()
    assert ops.LibraryStageTimer.collect() == {}                    # off by default
    with ops.LibraryStageTimer(None):
        step()
        step()
    got = ops.LibraryStageTimer.collect()
    assert {k: v["launches"] for k, v in got.items()} == {"kagnn_aggregate_sum": 4, "kagnn_kan_pack_batch": 2, "kagnn_kan_linear_fwd": 4,
                                                           "kagnn_kan_linear_bwd_weight": 4, "kagnn_kan_linear_bwd_input": 4}, got
    assert all(v["total_ms"] > 0.0 and abs(v["avg_ms"] * v["launches"] - v["total_ms"]) < 1e-9 for v in got.values())
    with ops.LibraryStageTimer("kagnn_aggregate_sum"):
        step()
    only = ops.LibraryStageTimer.collect()
    assert list(only) == ["kagnn_aggregate_sum"] and only["kagnn_aggregate_sum"]["launches"] == 2
    step()
    assert ops.LibraryStageTimer.collect() == {}


@pytest.mark.parametrize("f", [64, 128, 8, 24, 256, 12, 40])
def test_bf16_aggregation_vs_oracle_on_the_rounded_rows(f):
    """kagnn_aggregate_sum_bf16: fp32 accumulation of bf16 rows is EXACT arithmetic on the rounded inputs up to fp32
    summation order -- compare with the fp64 oracle fed the same rounded rows (fp32 output: 2e-6; bf16 output: one
    rounding, 2^-8 relative), hubs and isolated nodes included, both directions, GIN and GCN forms"""
    n, e = 20000, 200000
    ei = orc.powerlaw_graph(n, e, seed=1)
    g = ops.GraphIndex(ei.to(DEV), n)
    assert g.num_hub_seg > 0
    gen = torch.Generator().manual_seed(f)
    xb = torch.randn(n, f, generator=gen).to(torch.bfloat16)
    x64 = xb.double()
    want = orc.sum_aggregate(x64, ei) + 1.5 * x64
    got = ops._aggregate_raw(xb.to(DEV), g, False, 1.5, None, None, None, None, False)
    assert got.dtype == torch.float32
    assert_close(got, want, 2e-6, what=f"bf16 gather, fp32 sums F={f}")
    got16 = ops._aggregate_raw(xb.to(DEV), g, False, 1.5, None, None, None, None, False, out_dtype=torch.bfloat16)
    assert got16.dtype == torch.bfloat16
    err = (got16.double().cpu() - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -8 + 1e-30).all()), float((err / want.abs().clamp_min(1e-30)).max())
    want_t = orc.sum_aggregate(x64, ei.flip(0)) + 1.5 * x64                         # transposed direction
    got_t = ops._aggregate_raw(xb.to(DEV), g, True, 1.5, None, None, None, None, False)
    assert_close(got_t, want_t, 2e-6, what=f"bf16 gather transposed F={f}")
    # GCN form: in/out scales, bias, self loops skipped
    dis = g.gcn_dis
    bias = torch.randn(f, generator=gen)
    ei2, w2 = orc.gcn_norm(ei, n, torch.float64)
    want_g = orc.sum_aggregate(x64, ei2, n, w2) + bias.double()
    got_g = ops._aggregate_raw(xb.to(DEV), g, False, 1.0, None, dis, dis, bias.to(DEV), True)
    assert_close(got_g, want_g, 2e-6, what=f"bf16 gather GCN form F={f}")
    # bit-reproducible (hub rows included)
    assert torch.equal(got, ops._aggregate_raw(xb.to(DEV), g, False, 1.5, None, None, None, None, False))


def test_bf16_mode_gin_kan_layer_vs_oracle(monkeypatch):
    """KAGNN_ACT=bf16 on the KAN-GIN layer: (a) against the fp64 oracle fed the bf16-ROUNDED input the forward is as
    tight as the fp32 mode (the KAN chain is untouched); (b) against the fp64 oracle on the ORIGINAL input everything
    is within 4e-3 (bf16 has 8 significant bits: 2^-9 per gathered element) -- the tolerance of this build-defined
    mode; (c) the gradient of a bf16 input comes back as bf16."""
    monkeypatch.setenv("KAGNN_ACT", "bf16")
    n, e, f = 30000, 300000, 64
    ei = orc.powerlaw_graph(n, e, seed=2)
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(n, f, generator=gen) * 0.25
    gy = torch.randn(n, f, generator=gen)
    torch.manual_seed(7)
    conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2)
    layers = [{k: v.detach().clone().double() for k, v in l.state_dict().items()} for l in conv.nn.layers]
    conv = conv.to(DEV)
    g = ops.GraphIndex(ei.to(DEV), n)
    xd = x.to(DEV).requires_grad_(True)
    y = conv(xd, g)
    y.backward(gy.to(DEV))
    assert xd.grad.dtype == torch.float32
    y_r, _, _ = orc.kan_gin_layer_fwd_bwd(x.to(torch.bfloat16).double(), ei, layers, 3, gy.double())
    assert_close(y, y_r, what="bf16 mode: forward on the rounded input")
    # ... and BOTH directions against the oracle that rounds the two gathered matrices where the mode does: the parity
    # statement of the mode.  Outputs and parameter gradients at the fp32 tolerance; the input gradient in L2 (an element of
    # d loss / d h0 that fp32 and fp64 arithmetic put on different sides of a bf16 tie is off by one bf16 ulp = 3.9e-3)
    from helpers import bf16_gather_oracle
    xr = x.double().requires_grad_(True)
    ps = [{k: (v.clone().requires_grad_(True) if k != "grid" else v) for k, v in p.items()} for p in layers]

    def chain(h):
        for p in ps:
            h = orc.kan_linear_forward(h, p["base_weight"], p["spline_weight"], p["spline_scaler"], p["grid"], 3)
        return h
    with bf16_gather_oracle():
        y_q = orc.gin_conv(xr, ei, chain)
        y_q.backward(gy.double())
    assert_close(y, y_q.detach(), what="bf16 mode vs the rounding oracle: y")
    for li, layer in enumerate(conv.nn.layers):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            assert_close(getattr(layer, k).grad, ps[li][k].grad, what=f"bf16 mode vs the rounding oracle: L{li}.{k}", elementwise=False)
    gx_l2 = float((xd.grad.double().cpu() - xr.grad).norm() / xr.grad.norm())
    assert gx_l2 <= 1e-4, gx_l2
    assert_close(xd.grad, xr.grad, 6e-3, what="bf16 mode vs the rounding oracle: gx (1.5 bf16 ulp)", elementwise=False)
    y64, gx64, g64 = orc.kan_gin_layer_fwd_bwd(x.double(), ei, layers, 3, gy.double())
    # (max-norm only: sums of ~10 rounded terms with cancellation have no per-element relative bound)
    assert_close(y, y64, 4e-3, what="bf16 mode y", elementwise=False)
    assert_close(xd.grad, gx64, 8e-3, what="bf16 mode gx", elementwise=False)
    for li, layer in enumerate(conv.nn.layers):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            # parameter gradients: sums over 30 000 rows of products with the rounded activations
            assert_close(getattr(layer, k).grad, g64[li][k], 1e-2, what=f"bf16 mode L{li}.{k}", elementwise=False)
    xb = x.to(torch.bfloat16).to(DEV).requires_grad_(True)
    conv.zero_grad()
    conv(xb, g).backward(gy.to(DEV))
    assert xb.grad.dtype == torch.bfloat16
    assert_close(xb.grad.float(), gx64, 8e-3, what="bf16 mode gx (bf16 rows)", elementwise=False)
    # the mode really runs the bf16 kernels (composed form: the per-op entry points are visible to the timer)
    monkeypatch.setattr(ops, "_LAYER_ABI", False)
    timer = ops.EntryPointTimer()
    ops.set_timer(timer)
    conv(xd.detach().requires_grad_(True), g).sum().backward()
    ops.set_timer(None)
    assert timer.summary()["kagnn_aggregate_sum_bf16"]["launches"] == 2 and "kagnn_aggregate_sum" not in timer.summary()


def test_bf16_mode_node_model_runs_and_stays_close(monkeypatch):
    """GKAN_Nodes / GFASTKAN_Nodes under KAGNN_ACT=bf16 against their own fp32 run"""
    n, e = 20000, 160000
    ei = orc.powerlaw_graph(n, e, seed=3).to(DEV)
    x = (torch.randn(n, 32, generator=torch.Generator().manual_seed(8)) * 0.5).to(DEV)
    for cls in (kagnn_amd.GKAN_Nodes, kagnn_amd.GFASTKAN_Nodes):
        torch.manual_seed(9)
        model = cls("gin", 2, 32, 16, 6, skip=True, grid_size=4).to(DEV).train()
        monkeypatch.setenv("KAGNN_ACT", "fp32")
        ref = model(x, ei)
        monkeypatch.setenv("KAGNN_ACT", "bf16")
        out = model(x, ei)
        out.sum().backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.requires_grad)
        assert_close(out, ref, 2e-2, what=f"{cls.__name__} bf16 vs fp32 logits", elementwise=False)     # two BatchNorms amplify the 2^-9 input rounding


def test_bf16_mode_first_layer_wider_than_the_bf16_aggregation(monkeypatch):
    """a first layer wider than the bf16 aggregation's 512-column limit (1024 -> 64) under KAGNN_ACT=bf16 --
    the forward falls back to fp32 gathers; the backward used to have dX write bf16 rows and then fail in
    kagnn_aggregate_sum_bf16.  Both directions must run and equal the fp32 mode."""
    n, e = 3000, 20000
    ei = orc.powerlaw_graph(n, e, seed=5).to(DEV)
    x = (torch.randn(n, 1024, generator=torch.Generator().manual_seed(6)) * 0.3).to(DEV)
    gy = torch.randn(n, 64, generator=torch.Generator().manual_seed(7)).to(DEV)
    torch.manual_seed(8)
    conv = kagnn_amd.GIKANLayer(1024, 64, grid_size=5, spline_order=3, hidden_dim=64, nb_layers=2).to(DEV)
    res = {}
    for act in ("fp32", "bf16"):
        monkeypatch.setenv("KAGNN_ACT", act)
        conv.zero_grad()
        xd = x.clone().requires_grad_(True)
        y = conv(xd, ei)
        y.backward(gy)
        res[act] = (y.detach(), xd.grad.float(), conv.nn.layers[0].spline_weight.grad.clone())
    for a, b, what in zip(res["bf16"], res["fp32"], ("y", "gx", "g_spline_weight")):
        assert_close(a, b, 1e-6, what=f"bf16 mode, 1024-wide first layer: {what}", elementwise=False)


def test_fused_layer_refuses_a_graph_built_for_another_node_count():
    """the one-call layer ABI indexes rowptr / x by x.size(0); a GraphIndex of another size must raise, not read
    out of bounds."""
    ei = orc.powerlaw_graph(500, 3000, seed=1).to(DEV)
    g = ops.GraphIndex(ei, 500)
    conv = kagnn_amd.GIKANLayer(16, 16, grid_size=5, spline_order=3, hidden_dim=16, nb_layers=2).to(DEV)
    with pytest.raises(ValueError, match="nodes"):
        conv(torch.randn(400, 16, device=DEV), g)


# Tolerances of the build-defined bf16 gather mode against the UNROUNDED fp64 oracle, relative to the
# largest element of the reference.  Derivation: a gathered element carries one round-to-nearest-even to 8 significant bits,
# relative error <= u = 2^-9 = 1.95e-3.  A row sum over ~10 neighbours keeps at most u relative to the sum of magnitudes;
# training-mode BatchNorm rescales by 1 / std and so preserves the error relative to the layer's own range, amplified by
# kappa = range / (c std) of the convolution output (1 .. 1.7 measured per layer); L stacked convolutions add up to first
# order: |d logits| <= L * kappa * u * max|logits| = 3 * 1.7 * 1.95e-3 = 1e-2 for the 3-layer model.  Input gradient: the
# same on the way back plus the forward perturbation of the activations it is evaluated at: 2x.  Parameter gradients (sums
# over all rows of products with perturbed activations AND perturbed upstream gradients): 2.5x.  Measured over 5 seeds
# (test_bf16_mode_tolerances_hold_with_headroom_over_seeds, gpurun_out/bf16_seed_errors.json): logits <= 4.3e-3, i.e. a
# factor 2.3 below the bound.
# Round 5: parameter gradients are now measured relative to EACH gradient's own largest element (they were relative to max(1, .),
# i.e. absolute for these ~1e-4 tensors -- VERDICT r04 weak 1).  On that scale the mode costs the gradients that are themselves
# cancelling sums the most: spline_scaler (sum over c of gW * spline_weight) and the norms' bias gradients reach 4.2..5.9e-2 over
# five seeds (gpurun_out/bf16_seed_errors.json: per_param), everything else stays below 3.9e-2.  A measured property of the
# mode, not a derived bound: 1.2e-1 leaves the factor 2 the head-room test asks for.  Parity proper is the comparison against the
# bf16-ROUNDING oracle below (same tensors: <= 6.2e-3).
BF16_TOL = {"logits": 1e-2, "gx": 2.5e-2, "params": 1.2e-1}
# ... and against the oracle that rounds the same two matrices per convolution to bf16 (helpers.bf16_gather_oracle).  At LAYER
# level that is the parity statement proper and it is tight (test_bf16_mode_gin_kan_layer_vs_oracle: y 2e-6, parameter
# gradients 1e-5, input gradient 1e-4 in L2).  At MODEL level it cannot be: where fp32 and fp64 arithmetic land on different
# sides of a bf16 tie (~6e-4 of the elements after a KAN chain + BatchNorm) the two paths differ by ONE bf16 ulp = 3.9e-3
# of that element; measured: logits 1e-4 in L2 -- 17x below the unrounded oracle -- while gradients, sums of mixed-sign terms
# over 170 000 rows, amplify the same flips to 1e-3 .. 2e-3 (tools/archive/debug/bf16_model_dbg.py lists them per parameter).
BF16_ROUNDED_TOL = {"logits": 1e-3, "gx": 8e-3, "params": 8e-3}
BF16_ROUNDED_L2 = {"logits": 5e-4, "gx": 5e-3, "params": 1e-2}


def _bf16_model_errors(model, n, ei, x, gout, want, gx_want, g_want, expect_fused=None):
    model = model.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    timer = ops.EntryPointTimer()
    ops.set_timer(timer)
    try:
        out = model(xd, ei.to(DEV))
        out.backward(gout.to(DEV))
    finally:
        ops.set_timer(None)
    if expect_fused is not None:      # the default (fused conv + norm) path
        assert sum(1 for r in timer.records if r[0] in ("kagnn_gin_kan_layer_bwd_bn", "kagnn_gin_kan_layer_bwd_bn_sums")) == expect_fused
    rel = lambda a, b: float((a.detach().double().cpu() - b).abs().max() / max(1e-30, float(b.abs().max())))      # relative to the largest element (these gradients are ~1e-6: max(1, .) would hide them)
    l2 = lambda a, b: float((a.detach().double().cpu() - b).norm() / max(1e-30, float(b.norm())))
    errs = {"logits": rel(out, want), "gx": rel(xd.grad, gx_want), "params": 0.0,
            "logits_l2": l2(out, want), "gx_l2": l2(xd.grad, gx_want), "params_l2": 0.0}
    for name, p in model.named_parameters():
        parts = name.split(".")
        if not p.requires_grad or (parts[0] == "convs" and parts[-1] == "bias" and len(parts) == 3):
            continue                                   # (a bias in front of BatchNorm: identically zero gradient, see _model_vs_oracle)
        # parameter gradients: relative to the gradient's OWN largest element, as everywhere else in this suite since round 5
        # (helpers.assert_close; it was max(1, .), an absolute bound for these ~1e-4 tensors)
        r = rel(p.grad, g_want[name])
        errs.setdefault("per_param", {})[name] = r
        errs["params"] = max(errs["params"], r)
        errs["params_l2"] = max(errs["params_l2"], l2(p.grad, g_want[name]))
    return errs


def test_bf16_mode_arxiv_shaped_kan_gin_model_vs_oracle(monkeypatch):
    """BASELINE config 2 AS WORDED ("ogbn-arxiv KAN-GIN 3-layer hidden=64 grid=5 bf16"): GKAN_Nodes('gin', 3, 128 -> 64, 40
    classes) at ogbn-arxiv's shape under KAGNN_ACT=bf16 -- logits, d/dx and every parameter gradient (a) against the fp64
    oracle that rounds the gathered matrices where the mode does (parity proper, BF16_ROUNDED_TOL) and (b) against the fp64
    oracle of the SAME model on unrounded matrices (what the mode costs, BF16_TOL: derived above)."""
    from helpers import bf16_gather_oracle
    monkeypatch.setenv("KAGNN_ACT", "bf16")
    n, ei, x = _arxiv_like()
    torch.manual_seed(6)
    model = kagnn_amd.GKAN_Nodes("gin", 3, 128, 64, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2)
    gout = torch.randn(n, 40, generator=torch.Generator().manual_seed(7)) / n
    # (same model, input and output gradient as test_arxiv_shaped_kan_gin_model_vs_oracle: one oracle run serves both)
    want, gx_want, g_want = _oracle_case("arxiv.kan_gin", model, x, ei, gout, "kan", "gin", 3, 3, 8192, torch.float64)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with bf16_gather_oracle():
        want_r, gx_r, g_r = oracle_node_model_fwd_bwd(x, ei, state, gout, "kan", "gin", 3, 3, 8192, torch.float64)
    errs_r = _bf16_model_errors(model, n, ei, x, gout, want_r, gx_r, g_r, expect_fused=3)
    model.zero_grad()
    errs = _bf16_model_errors(model, n, ei, x, gout, want, gx_want, g_want)
    from helpers import check
    for k in BF16_TOL:
        check(errs_r[k] <= BF16_ROUNDED_TOL[k], f"bf16 arxiv vs the bf16-rounding oracle {k}", errs_r)
        check(errs_r[k + "_l2"] <= BF16_ROUNDED_L2[k], f"bf16 arxiv vs the bf16-rounding oracle, L2 {k}", errs_r)
        check(errs[k] <= BF16_TOL[k], f"bf16 arxiv vs the unrounded oracle {k}", errs)


def test_bf16_mode_tolerances_hold_with_headroom_over_seeds(monkeypatch):
    """the mode's tolerances must not be numbers fitted to one seed.  Five model initialisations of the
    config-2 model (3 x KAN-GIN 128 -> 64, 40 classes) on a 30 000-node power-law graph, each against BOTH oracles; the worst
    case over the seeds must leave a factor 2 to BF16_TOL.  The per-seed errors go to gpurun_out/bf16_seed_errors.json."""
    import json, os
    from helpers import bf16_gather_oracle
    monkeypatch.setenv("KAGNN_ACT", "bf16")
    n, e = 30_000, 210_000
    ei = orc.powerlaw_graph(n, e, seed=21)
    x = torch.randn(n, 128, generator=torch.Generator().manual_seed(22)) * 0.5
    gout = torch.randn(n, 40, generator=torch.Generator().manual_seed(23)) / n
    report = {}
    for seed in (1, 2, 3, 4, 5):
        torch.manual_seed(seed)
        model = kagnn_amd.GKAN_Nodes("gin", 3, 128, 64, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2)
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        plain = oracle_node_model_fwd_bwd(x, ei, state, gout, "kan", "gin", 3, 3, 8192, torch.float64)
        with bf16_gather_oracle():
            rounded = oracle_node_model_fwd_bwd(x, ei, state, gout, "kan", "gin", 3, 3, 8192, torch.float64)
        errs_r = _bf16_model_errors(model, n, ei, x, gout, *rounded, expect_fused=3)
        model.zero_grad()
        errs = _bf16_model_errors(model, n, ei, x, gout, *plain)
        report[seed] = {"vs_unrounded_oracle": errs, "vs_bf16_rounding_oracle": errs_r}
    keys = list(BF16_TOL) + [k + "_l2" for k in BF16_TOL]
    worst = {k: max(r["vs_unrounded_oracle"][k] for r in report.values()) for k in keys}
    worst_r = {k: max(r["vs_bf16_rounding_oracle"][k] for r in report.values()) for k in keys}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bf16_seed_errors.json"), "w") as fh:
        json.dump({"per_seed": report, "worst_vs_unrounded": worst, "worst_vs_rounding_oracle": worst_r,
                   "tolerance_vs_unrounded": BF16_TOL, "tolerance_vs_rounding_oracle": BF16_ROUNDED_TOL}, fh, indent=1)
    from helpers import check
    for k in BF16_TOL:
        check(worst[k] <= 0.5 * BF16_TOL[k], f"bf16 seeds: head-room below 2x {k}", worst)
        check(worst_r[k] <= BF16_ROUNDED_TOL[k], f"bf16 seeds vs the bf16-rounding oracle {k}", worst_r)
        check(worst_r[k + "_l2"] <= BF16_ROUNDED_L2[k], f"bf16 seeds vs the bf16-rounding oracle, L2 {k}", worst_r)


# ------------------------------------------------------------------ torch.library registration
@pytest.mark.parametrize("arch,kind", [("kan", "gin"), ("kan", "gcn"), ("fastkan", "gin"), ("fastkan", "gcn")])
def test_models_trace_into_one_graph_of_kagnn_ops(arch, kind):
    """torch.compile(fullgraph=True) must not graph-break on the ctypes calls: under dynamo the layers route to the
    torch.library ops of kagnn_amd/library.py.  Same kernels => same results as eager, forward and backward
    (backend aot_eager: the point is the opaque-op registration + autograd formulas, not a code generator)."""
    import torch._dynamo
    from kagnn_amd import library                                    # registers kagnn::*
    assert "kagnn::kan_linear" in str(torch.ops.kagnn.kan_linear.default._schema)
    torch._dynamo.reset()
    n, e, fin, hid, classes = 3000, 24000, 24, 16, 5
    ei = orc.powerlaw_graph(n, e, seed=7)
    torch.manual_seed(11)
    cls = kagnn_amd.GKAN_Nodes if arch == "kan" else kagnn_amd.GFASTKAN_Nodes
    model = cls(kind, 2, fin, hid, classes, skip=True, grid_size=4).to(DEV).train()
    g = ops.GraphIndex(ei.to(DEV), n)
    x = (torch.randn(n, fin, generator=torch.Generator().manual_seed(1)) * 0.5).to(DEV)
    gout = torch.randn(n, classes, generator=torch.Generator().manual_seed(2)).to(DEV)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(fn):
        model.load_state_dict(state)                                 # same BatchNorm running statistics going in
        model.zero_grad()
        xr = x.clone().requires_grad_(True)
        out = fn(xr, g)
        out.backward(gout)
        return ([out.detach(), xr.grad] + [p.grad.clone() for p in model.parameters() if p.grad is not None],
                {k: v.clone() for k, v in model.state_dict().items() if "running" in k})
    eager, eager_stats = run(model)
    compiled = torch.compile(model, backend="aot_eager", fullgraph=True)
    traced, traced_stats = run(compiled)
    for a, b in zip(eager, traced):
        # (eager takes the batch statistics from the convolution's epilogue, traced code from the statistics pass)
        assert_close(b, a, 4e-6, what=f"compiled vs eager {arch}/{kind}")
    for k in eager_stats:
        assert_close(traced_stats[k], eager_stats[k], 1e-6, what=f"compiled {k}")
    # the traced graph holds the opaque ops
    seen = []

    def spy(gm, example_inputs):
        seen.extend(str(nd.target) for nd in gm.graph.nodes if nd.op == "call_function")
        return gm.forward
    torch._dynamo.reset()
    torch.compile(model, backend=spy, fullgraph=True)(x, g)
    assert any("kagnn" in t for t in seen), seen[:20]


# ------------------------------------------------------------------ precision report
def test_precision_report_split_vs_fp32_vs_reference_fp32():
    """errors against the fp64 oracle, side by side: the HIP path in exact-fp32 mode, in split mode, and the
    reference's own fp32 arithmetic (the oracle run in fp32, bit-identical to ekan.py on CPU).  The split path must
    stay in the reference's own error class."""
    import json, os
    gen = torch.Generator().manual_seed(31)
    report = {}
    for (n, fi, fo, G) in [(4096, 64, 64, 5), (4096, 128, 128, 8)]:
        p = orc.init_kan_linear(fi, fo, G, 3, gen)
        x = torch.randn(n, fi, generator=gen) * 0.8
        gy = torch.randn(n, fo, generator=gen)
        y64, gx64, g64 = oracle_kan_linear_fwd_bwd(x, gy, p, 3)
        y32, gx32, g32 = oracle_kan_linear_fwd_bwd(x, gy, p, 3, dtype=torch.float32)

        def errs(y, gx, g):
            rel = lambda a, b: float((a.double().cpu() - b).abs().max() / float(b.abs().max()))
            return {"y": rel(y, y64), "gx": rel(gx, gx64), **{"g_" + k: rel(g[k], g64[k]) for k in g64}}
        row = {"reference_fp32": errs(y32, gx32, g32)}
        for mode, name in zip(MODES, MODE_IDS):
            layer = kagnn_amd.KANLinear(fi, fo, grid_size=G, spline_order=3)
            layer.load_state_dict(p)
            layer = layer.to(DEV)
            layer.precision = mode
            xd = x.to(DEV).requires_grad_(True)
            y = layer(xd)
            y.backward(gy.to(DEV))
            row["hip_" + name] = errs(y.detach(), xd.grad, {k: getattr(layer, k).grad for k in g64})
        report[f"KANLinear({fi},{fo},G={G}) N={n}"] = row
        for k, ref_err in row["reference_fp32"].items():
            assert row["hip_split"][k] <= max(8 * ref_err, 4e-6), (k, row)
            assert row["hip_fp32"][k] <= max(8 * ref_err, 4e-6), (k, row)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, "precision_report.json"), "w"), indent=1)


# ------------------------------------------------------------------ folds on vs off over random model configurations
def test_fuzz_models_folds_on_vs_off():
    """tools/fuzz_models.py as part of the suite: 40 random GKAN_Nodes configurations (widths 32..128, 1-4 layers, 700..70 001 rows,
    dropout, skip on / off, hubs both ways), the default path (norms folded, statistics travelling with the gradients, one-launch
    read-out) against every fold switched off -- loss, input gradient, every parameter gradient and buffer.  This is the defence of
    the switch surface: a fold that only an A/B flag used to exercise is exercised here."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_models.py"), "40", "0"], cwd=root, capture_output=True, text=True,
                       timeout=900)
    tail = r.stdout[-3000:]
    assert r.returncode == 0 and "failures: 0" in r.stdout, tail + r.stderr[-2000:]
    m = re.search(r"folded norms: (\d+), statistics made by producers: (\d+)", r.stdout)
    assert m and int(m.group(1)) >= 20 and int(m.group(2)) >= 20, tail       # the folds really ran


def test_fuzz_graph_models_one_node_vs_five_nodes_vs_oracle():
    """tools/fuzz_graph_models.py inside the suite: 30 random graph-regression models / mini-batches (hidden 8..64, 2..5 convolutions,
    1..300 graphs, edgeless batches, tables of 2..300 rows): the whole forward as one tape node is bit-identical to its five-node
    form and as close to the fp64 oracle as the per-operation composition (or 1e-4)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_graph_models.py"), "30", "0"], cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"ran as one tape node: (\d+) of 30, as one library call each way: (\d+)", r.stdout)
    assert m and int(m.group(1)) >= 25 and int(m.group(2)) >= 25, r.stdout[-1000:]      # (round 6: + kagnn_kagin_model_fwd / _bwd, bit-identical)


# ------------------------------------------------------------------ the harness optimiser (optuna_zinc.py:49,62: torch.optim.Adam)
@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_harness_adam_is_torch_adam(weight_decay):
    """kagnn_amd.harness.Adam (kagnn_adam_step: one launch per 64 tensors) against torch.optim.Adam run in float64 on the same
    gradients: 40 tensors of 1 .. 100 003 elements, 6 steps, a tensor without a gradient is left alone"""
    from kagnn_amd.harness import Adam
    gen = torch.Generator().manual_seed(9)
    sizes = [1, 2, 63, 64, 65, 4096, 100_003] + [int(v) for v in torch.randint(1, 5000, (33,), generator=gen)]
    ps = [torch.nn.Parameter(torch.randn(n, generator=gen).to(DEV)) for n in sizes]
    ps[5] = torch.nn.Parameter(torch.randn(64, 64, generator=gen).to(DEV))
    ref = [torch.nn.Parameter(p.detach().double().clone()) for p in ps]
    frozen = 11
    mine = Adam(ps, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
    theirs = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
    for step in range(6):
        mine.zero_grad(); theirs.zero_grad()
        for k, (p, r) in enumerate(zip(ps, ref)):
            if k == frozen:
                continue
            g = torch.randn(p.shape, generator=gen).to(DEV) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=gen)))
            p.grad = g
            r.grad = g.double()
        mine.step(); theirs.step()
        for k, (p, r) in enumerate(zip(ps, ref)):
            assert_close(p.detach(), r.detach(), 2e-6, what=f"adam step {step} tensor {k}", elementwise=False)
    assert torch.equal(ps[frozen].detach().double(), ref[frozen].detach())
    with pytest.raises(TypeError, match="no CPU path"):
        Adam([torch.nn.Parameter(torch.zeros(3))])
    with pytest.raises(NotImplementedError, match="amsgrad"):
        Adam(ps, amsgrad=True)


def test_harness_adam_counts_steps_per_tensor_like_torch():
    """ torch.optim.Adam keeps one step count PER PARAMETER and does not advance it while that parameter has no gradient:
    a tensor that sits out steps 2-3 gets the bias correction of ITS third step at the optimiser's fifth.  A step with no gradient
    anywhere is not a step."""
    from kagnn_amd.harness import Adam
    gen = torch.Generator().manual_seed(10)
    ps = [torch.nn.Parameter(torch.randn(n, generator=gen).to(DEV)) for n in (5, 300, 4097)]
    ref = [torch.nn.Parameter(p.detach().double().clone()) for p in ps]
    mine = Adam(ps, lr=1e-2)
    theirs = torch.optim.Adam(ref, lr=1e-2)
    mine.zero_grad(); theirs.zero_grad()
    mine.step(); theirs.step()                               # nothing has a gradient: neither optimiser moves or counts
    for step in range(7):
        mine.zero_grad(); theirs.zero_grad()
        for k, (p, r) in enumerate(zip(ps, ref)):
            if (k == 1 and step in (2, 3)) or (k == 2 and step == 5):
                continue
            g = torch.randn(p.shape, generator=gen).to(DEV)
            p.grad, r.grad = g, g.double()
        mine.step(); theirs.step()
        for k, (p, r) in enumerate(zip(ps, ref)):
            assert_close(p.detach(), r.detach(), 2e-6, what=f"adam step {step} tensor {k}", elementwise=False)


# ------------------------------------------------------------------ the graph-level training loop (optuna_zinc.py:56-66)
def test_train_graph_batches_is_the_reference_loop():
    """harness.train_graph_batches = `for data in loader: zero_grad; loss = L1(model(data).squeeze(), data.y); backward; step` with
    Adam (graph_regression/optuna_zinc.py:56-66) over distinct mini-batches (a CSR per batch, the GINE stack as one tape node,
    embedding encoders): with torch's fused Adam passed in, the same losses and the same final parameters as that loop written out
    by hand, bit for bit; with the package's own Adam (the default) the same trajectory to rounding; the loss goes down."""
    from types import SimpleNamespace
    from kagnn_amd.harness import train_graph_batches
    B, H = 32, 32
    batches = []
    for k in range(4):
        g = torch.Generator().manual_seed(50 + k)
        sizes = torch.randint(10, 30, (B,), generator=g)
        n = int(sizes.sum()); off = torch.cumsum(sizes, 0) - sizes
        src, dst, batch = [], [], []
        for b in range(B):
            nb = int(sizes[b]); eb = 2 * nb + 3
            src.append(torch.randint(0, nb, (eb,), generator=g) + off[b]); dst.append(torch.randint(0, nb, (eb,), generator=g) + off[b])
            batch.append(torch.full((nb,), b))
        e = sum(len(s_) for s_ in src)
        x = torch.randint(0, 21, (n, 1), generator=g)
        batches.append(SimpleNamespace(x=x.to(DEV), edge_index=torch.stack([torch.cat(src), torch.cat(dst)]).to(DEV),
                                       edge_attr=torch.randint(0, 4, (e,), generator=g).to(DEV), batch=torch.cat(batch).to(DEV), num_graphs=B,
                                       y=(x.float().mean() + torch.randn(B, generator=g) * 0.1).to(DEV)))

    def make():
        torch.manual_seed(3)
        m = kagnn_amd.KAGINRegression(1, 1, 3, H, 2, 4, 3, 1, 0.0, True)
        m.atom_encoder = kagnn_amd.graph_models.AtomEncoder(H, [21])
        m.bond_encoder.bond_embedding_list = torch.nn.ModuleList([torch.nn.Embedding(4, H)])
        return m.to(DEV)
    m1 = make()
    # (a second seeded construction is NOT the same model to the bit: the spline-weight init is a CPU lstsq, as in the reference's
    # curve2coeff, whose last bit depends on buffer alignment -- tools/archive/experiments/train_determinism.py; the copy below is)
    initial = {k: v.clone() for k, v in m1.state_dict().items()}
    t, means = train_graph_batches(m1, batches, nb_epochs=6, warmup=0, lr=2e-3, optimizer=torch.optim.Adam(m1.parameters(), lr=2e-3, fused=True))
    assert t > 0 and all(np.isfinite(means)) and means[-1] < means[0], means
    # the default optimiser (kagnn_amd.harness.Adam: the same rule, one library call) follows the same trajectory to rounding
    m3 = make()
    m3.load_state_dict(initial)
    _, means3 = train_graph_batches(m3, batches, nb_epochs=6, warmup=0, lr=2e-3)
    assert np.allclose(means3[:3], means[:3], rtol=1e-4, atol=0) and np.allclose(means3, means, rtol=2e-2, atol=0), (means3, means)
    m2 = make()
    m2.load_state_dict(initial)
    opt = torch.optim.Adam(m2.parameters(), lr=2e-3, fused=True)
    m2.train()
    want = []
    for _ in range(6):
        tot = 0.0
        for d in batches:
            opt.zero_grad()
            loss = torch.nn.L1Loss()(m2(d).squeeze(), d.y)
            loss.backward()
            opt.step()
            tot += float(loss) * d.num_graphs
        want.append(tot / (len(batches) * B))
    assert np.allclose(means, want, rtol=1e-6, atol=0), (means, want)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_run_reference_runs_the_standin_timing_script_on_the_gpu(tmp_path):
    """`python -m kagnn_amd.run_reference tests/standin/time_model_standin.py`: the script with time_model.py:13-15's import lines, unedited,
    trains the four KAN / FastKAN node models on the HIP path (VERDICT r05 missing 5)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "report.json"
    r = subprocess.run([sys.executable, "-m", "kagnn_amd.run_reference", os.path.join(root, "tests", "standin", "time_model_standin.py"), str(out)],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(out.read_text())
    assert rep["device"] == "cuda" and len(rep["classes"]) == 4
    for k, v in rep["classes"].items():
        assert "error" not in v and v["module"] == "kagnn_amd.models", (k, v)
        assert all(np.isfinite(v["losses"])) and v["losses"][-1] < v["losses"][0], (k, v["losses"])


def test_prefetched_graph_index_is_the_same_index():
    """ops.prefetch_graph_index builds a later mini-batch's CSR + transpose on a side stream; graph_index(cache=False) adopts it: the same six
    arrays, bit for bit, as the direct build; an edge list that is never consumed does not pile up; a big graph is not prefetched."""
    g = torch.Generator().manual_seed(9)
    n, e = 5000, 12000
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    want = ops.GraphIndex(ei, n)
    assert ops.prefetch_graph_index(ei, n) and len(ops._prefetched) == 1
    got = ops.graph_index(ei, n, cache=False)
    assert len(ops._prefetched) == 0
    for name in ("rowptr", "col", "perm", "rowptr_t", "col_t", "perm_t"):
        assert torch.equal(getattr(got, name), getattr(want, name)), name
    x = torch.randn(n, 16, generator=g).to(DEV)
    assert torch.equal(ops.aggregate_sum(x, got), ops.aggregate_sum(x, want))
    for k in range(7):                                       # never consumed: bounded
        ops.prefetch_graph_index(torch.randint(0, n, (2, e + k), generator=g).to(DEV), n)
    assert len(ops._prefetched) <= 4
    ops.clear_graph_cache()
    big = torch.randint(0, 100000, (2, 200000), generator=g).to(DEV)
    assert not ops.prefetch_graph_index(big, 100000) and len(ops._prefetched) == 0
    ops.flush_graph_checks()


@pytest.mark.parametrize("workload", ["headline", "config3", "fastkan", "model"])
def test_bench_single_gpu_line_carries_the_contract(workload):
    """`python bench.py --workload X` at a small size: ONE JSON line with the driver's contract fields, the roofline block, the memory
    contract, and -- for the headline -- the cpu_baseline block and the secondary figures (model step, other layers, graph-level step)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--nodes", "30000", "--edges", "300000", "--steps", "3",
           "--warmup", "1", "--no-traffic", "--cpu-sample", "3000", "--cpu-sample-only"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "peak_device_GB", "device_memory"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "edges/s" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith(workload) and d["value"] > 0 and d["complete"] is True
    units = 3 * 300000 if workload == "model" else 300000
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] < 1.5 and roof["peak"] > 0 and "traffic" in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert d["peak_device_GB"] > 0 and d["device_memory"]["step_peak_over_resident_GB"] >= 0
    if workload == "headline":
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
        sec = d["secondary"]
        assert sec["model_step"]["ms_per_step"] > 0 and sec["graph_level_step"]["ms_per_step"] > 0 and len(sec["graph_level_step"]["repeats_ms_per_step"]) == 3
        assert set(sec["other_layers"]) == {"config3", "fastkan"} and d["fp32_mode_ms_per_step"] > d["ms_per_step"]
