"""The fused convolution epilogue of SURVEY.md 8(f) rank 1 -- `x = dropout(bn(conv(x)))`, reference
node_classification_clean/models.py:198-201 -- through the C ABI:

* kagnn_kan_linear_fwd_moments / kagnn_gin_kan_layer_fwd: the column moments (mean, sum of squared deviations) that the
  forward kernel's epilogue leaves behind equal the fp64 statistics of the very y it wrote, on ragged shapes, every
  kernel family (fused epilogue and the one-pass fallback) and an output whose mean dwarfs its spread;
* kagnn_batchnorm_fwd fed those moments == torch.nn.functional.batch_norm (fp64, CPU) incl. running statistics;
* the in-kernel dropout: keep rate, per-column / per-row uniformity, independence of neighbours, determinism in the seed,
  exact `bn(x) * mask / (1 - p)` values, and a backward equal to autograd through that expression with the SAME mask;
* the node models with the fused epilogue on and off agree, and train with dropout.
"""
import numpy as np
import math

import pytest
import torch
import torch.nn.functional as F

import kagnn_amd
from kagnn_amd import models, ops
from oracle import kan_oracle as orc
from helpers import TOL, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(fin, fout, grid, order, seed, mode=None):
    torch.manual_seed(seed)
    layer = kagnn_amd.KANLinear(fin, fout, grid_size=grid, spline_order=order)
    layer.precision = mode
    return layer.to(DEV)


def _fwd_with_moments(layer, x):
    mode = layer.precision if layer.precision is not None else ops.default_precision()
    return ops._kan_fwd_raw(x, layer.base_weight.contiguous(), layer.spline_weight.contiguous(),
                            layer.spline_scaler.contiguous(), layer._knots(), layer.grid_size, layer.spline_order, mode,
                            moments=True)


def _check_moments(y, mom, what):
    y64 = y.detach().double().cpu()
    n = y64.size(0)
    mean = y64.mean(0)
    m2 = ((y64 - mean) ** 2).sum(0)
    got_mean, got_m2 = mom[0].double().cpu(), mom[1].double().cpu()
    scale = y64.abs().max().clamp_min(1e-30)
    assert float((got_mean - mean).abs().max()) <= 2e-6 * float(scale), (what, float((got_mean - mean).abs().max()), float(scale))
    # M2: relative, plus the floor set by the fp32 representation of the column mean itself
    floor = n * (4 * 2.0 ** -24 * mean.abs()) ** 2
    bad = (got_m2 - m2).abs() > 2e-5 * m2 + floor + 1e-30
    assert not bool(bad.any()), (what, float(((got_m2 - m2).abs() / (m2 + floor + 1e-30)).max()))
    assert bool((got_m2 >= 0).all())


@pytest.mark.parametrize("n", [2, 31, 257, 5000, 100003])
@pytest.mark.parametrize("shape", [(64, 64, 5, 3), (16, 40, 5, 3), (128, 128, 8, 3), (20, 7, 4, 2), (64, 64, 3, 3)],
                         ids=["64x64g5", "16x40g5", "128x128g8", "20x7k2", "64x64g3"])
def test_forward_moments_equal_the_statistics_of_the_output(n, shape):
    fin, fout, grid, order = shape
    layer = _layer(fin, fout, grid, order, seed=n % 97)
    x = (torch.randn(n, fin, generator=torch.Generator().manual_seed(n)) * 0.6).to(DEV)
    y, _, mom = _fwd_with_moments(layer, x)
    y_plain = layer(x)
    assert torch.equal(y, y_plain), "the moments variant must write the same y"
    _check_moments(y, mom, f"{shape} n={n}")


def test_forward_moments_fp32_mode_and_few_row_split_launch():
    """exact-fp32 kernels and the launch split over the feature chunks (Cora: 2708 x 1433) take the one-pass fallback"""
    layer = _layer(64, 64, 5, 3, seed=1, mode=ops.PREC_FP32)
    x = (torch.randn(4099, 64, generator=torch.Generator().manual_seed(3)) * 0.5).to(DEV)
    y, _, mom = _fwd_with_moments(layer, x)
    assert torch.equal(y, layer(x))
    _check_moments(y, mom, "fp32 mode")
    layer = _layer(1433, 32, 5, 3, seed=2)
    x = (torch.rand(2708, 1433, generator=torch.Generator().manual_seed(4)) < 0.02).float().to(DEV)
    y, _, mom = _fwd_with_moments(layer, x)
    assert torch.equal(y, layer(x))
    _check_moments(y, mom, "cora-shaped split launch")


def test_forward_moments_when_the_mean_dwarfs_the_spread():
    """constant spline coefficients: every output is ~ in * w0 (partition of unity) with a spread of a few ulps --
    sum / sum-of-squares statistics would cancel; the pairwise update does not"""
    layer = _layer(64, 64, 5, 3, seed=5)
    with torch.no_grad():
        layer.spline_weight.fill_(0.75)
        layer.spline_scaler.fill_(1.0)
        layer.base_weight.mul_(1e-3)
    x = (torch.rand(50001, 64, generator=torch.Generator().manual_seed(5)) * 1.6 - 0.8).to(DEV)
    y, _, mom = _fwd_with_moments(layer, x)
    assert float(y.mean()) > 40.0 and float(y.std()) < 0.05
    _check_moments(y, mom, "large mean")
    rv = torch.ones(64, device=DEV)
    rm = torch.zeros(64, device=DEV)
    out = ops.batch_norm(y, None, None, rm, rv, True, 0.1, 1e-5, moments=mom)
    want = F.batch_norm(y.double().cpu(), None, None, None, None, True, 0.1, 1e-5)
    # Tolerance DERIVED, not fitted: the normalised value is (y - mean) * rstd with y ~ 48 and rstd ~ 1e2, so what shows is the fp32
    # rounding of the column MEAN times rstd.  The mean is a chain of pairwise merges (Chan) -- ceil(n / 256) workgroup rows folded
    # one after the other in the finish kernel, each fold rounding to half an ulp of the mean: a random walk of sqrt(merges) / 2 ulps
    # (sum / sum-of-squares statistics would be off by whole units here).  Allowed: exactly that estimate; observed: about a fifth
    # of it (the folds are hierarchical -- wave, workgroup, finish -- so fewer of them are sequential than the estimate assumes)
    yd = y.double()
    mean_abs = float(yd.mean(0).abs().max())
    rstd_max = float((yd.var(0, unbiased=False) + 1e-5).rsqrt().max())
    merges = -(-y.size(0) // 256)
    ulp = 2.0 ** (math.floor(math.log2(mean_abs)) - 23)
    bound = rstd_max * ulp * 0.5 * math.sqrt(merges)
    scale = float(want.abs().max())                   # (assert_close scales by the reference's own maximum)
    assert_close(out, want, bound / scale, what="bn of a near-constant column", elementwise=False)


@pytest.mark.parametrize("n,f", [(5000, 64), (100003, 64), (777, 40), (30000, 128)])
def test_batch_norm_from_moments_matches_torch(n, f):
    layer = _layer(64, f, 5, 3, seed=7)
    gen = torch.Generator().manual_seed(n)
    x = (torch.randn(n, 64, generator=gen) * 0.5).to(DEV)
    y, _, mom = _fwd_with_moments(layer, x)
    w, b = torch.randn(f, generator=gen), torch.randn(f, generator=gen)
    gy = torch.randn(n, f, generator=gen)
    res = []
    for use in (True, False):
        rm, rv = torch.full((f,), 0.3, device=DEV), torch.full((f,), 2.0, device=DEV)
        yd = y.clone().requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        out = ops.batch_norm(yd, wd, bd, rm, rv, True, 0.1, 1e-5, moments=mom if use else None)
        out.backward(gy.to(DEV))
        res.append((out, yd.grad, wd.grad, bd.grad, rm, rv))
    y64 = y.double().cpu().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    rm64, rv64 = torch.full((f,), 0.3, dtype=torch.float64), torch.full((f,), 2.0, dtype=torch.float64)
    want = F.batch_norm(y64, rm64, rv64, w64, b64, True, 0.1, 1e-5)
    want.backward(gy.double())
    for tag, r in zip(("moments", "plain"), res):
        assert_close(r[0], want, what=f"bn[{tag}] y")
        assert_close(r[1], y64.grad, what=f"bn[{tag}] gx")
        assert_close(r[2], w64.grad, what=f"bn[{tag}] g_weight")
        assert_close(r[3], b64.grad, what=f"bn[{tag}] g_bias")
        assert_close(r[4], rm64, what=f"bn[{tag}] running_mean")
        assert_close(r[5], rv64, what=f"bn[{tag}] running_var")


def test_layer_abi_and_composed_ops_return_the_same_moments(monkeypatch):
    n, e, f = 20000, 150000, 64
    ei = orc.powerlaw_graph(n, e, seed=3)
    g = ops.GraphIndex(ei.to(DEV), n)
    torch.manual_seed(11)
    conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2).to(DEV)
    x = (torch.randn(n, f, generator=torch.Generator().manual_seed(1)) * 0.25).to(DEV)
    res = []
    for abi in (True, False):
        monkeypatch.setattr(ops, "_LAYER_ABI", abi)
        y, mom = conv.forward_with_moments(x, g)
        res.append((y, mom))
        _check_moments(y, mom, f"layer abi={abi}")
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][0], conv(x, g))
    # FastKAN chains have no fused statistics: (y, None)
    fk = kagnn_amd.GIFASTKANLayer(f, f, grid_size=4, hidden_dim=f, nb_layers=2).to(DEV)
    y, mom = fk.forward_with_moments(x, g)
    assert mom is None and torch.equal(y, fk(x, g))


# ------------------------------------------------------------------ dropout inside the normalising pass
@pytest.mark.parametrize("p", [0.1, 0.5, 0.85])
def test_fused_dropout_statistics_values_and_backward(p):
    n, f = 200000, 64
    gen = torch.Generator().manual_seed(int(p * 100))
    x = torch.randn(n, f, generator=gen).to(DEV)
    w, b = (torch.rand(f, generator=gen) + 0.5).to(DEV), torch.randn(f, generator=gen).to(DEV)
    plain = ops.batch_norm(x, w, b, None, None, True, 0.1, 1e-5)
    xd = x.clone().requires_grad_(True)
    wd, bd = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.batch_norm(xd, wd, bd, None, None, True, 0.1, 1e-5, dropout_p=p, seed=1234)
    keep = y != 0
    assert not bool((plain == 0).any())
    # values: exactly bn(x) / (1 - p) where kept
    want = torch.where(keep, plain * (1.0 / (1.0 - p)), torch.zeros_like(plain))
    assert float((y - want).abs().max()) <= 1e-6 * float(want.abs().max())
    # keep rate overall / per column / per row, and independence of neighbours along both axes
    k = keep.float()
    q = 1.0 - p
    sd = (p * q) ** 0.5
    assert abs(float(k.mean()) - q) < 5 * sd / (n * f) ** 0.5 + 2.0 ** -16
    assert float((k.mean(0) - q).abs().max()) < 6 * sd / n ** 0.5 + 2.0 ** -16
    assert float((k.mean(1) - q).abs().max()) < 6.5 * sd / f ** 0.5 + 2.0 ** -16
    kc = k - q
    for a, c in ((kc[:, :-1], kc[:, 1:]), (kc[:-1], kc[1:]), (kc[:, :-4], kc[:, 4:]), (kc[:-64], kc[64:])):
        corr = float((a * c).mean()) / (p * q)
        assert abs(corr) < 5.0 / (a.numel()) ** 0.5, corr
    # determinism in the seed; another seed is another mask
    assert torch.equal(y, ops.batch_norm(x, w, b, None, None, True, 0.1, 1e-5, dropout_p=p, seed=1234))
    other = ops.batch_norm(x, w, b, None, None, True, 0.1, 1e-5, dropout_p=p, seed=1235) != 0
    assert abs(float((other & keep).float().mean()) - q * q) < 0.01
    # backward == autograd through bn(x) * mask / (1 - p) with the same mask
    gy = torch.randn(n, f, generator=gen).to(DEV)
    y.backward(gy)
    x2 = x.clone().requires_grad_(True)
    w2, b2 = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ops.batch_norm(x2, w2, b2, None, None, True, 0.1, 1e-5) * (k * (1.0 / (1.0 - p)))
    ref.backward(gy)
    for got, exp, what in ((xd.grad, x2.grad, "gx"), (wd.grad, w2.grad, "g_weight"), (bd.grad, b2.grad, "g_bias")):
        assert_close(got, exp.double().cpu(), what=f"fused dropout p={p} {what}", elementwise=False)
    # eval mode: no dropout
    rm, rv = torch.zeros(f, device=DEV), torch.ones(f, device=DEV)
    assert torch.equal(ops.batch_norm(x, w, b, rm, rv, False, 0.1, 1e-5, dropout_p=p, seed=9),
                       ops.batch_norm(x, w, b, rm, rv, False, 0.1, 1e-5))


def test_fused_dropout_edge_probabilities_and_ragged_widths():
    x = torch.randn(1001, 13, generator=torch.Generator().manual_seed(0)).to(DEV)       # F % 4 != 0: scalar path
    assert torch.equal(ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5, dropout_p=0.0),
                       ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5))
    assert float(ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5, dropout_p=1.0, seed=3).abs().max()) == 0.0
    y = ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5, dropout_p=0.5, seed=3)
    assert abs(float((y != 0).float().mean()) - 0.5) < 0.03
    with pytest.raises(ValueError):
        ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5, dropout_p=1.5)
    torch.manual_seed(5)
    a = ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5, dropout_p=0.5)
    torch.manual_seed(5)
    assert torch.equal(a, ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5, dropout_p=0.5))   # torch.manual_seed reproduces
    assert not torch.equal(a, ops.batch_norm(x, None, None, None, None, True, 0.1, 1e-5, dropout_p=0.5))


# ------------------------------------------------------------------ the node models
@pytest.mark.parametrize("kind", ["gin", "gcn"])
def test_node_model_with_and_without_the_fused_epilogue(kind, monkeypatch):
    n, e, fin = 30000, 200000, 48
    ei = orc.powerlaw_graph(n, e, seed=4).to(DEV)
    gen = torch.Generator().manual_seed(8)
    x = (torch.randn(n, fin, generator=gen) * 0.5).to(DEV)
    gout = torch.randn(n, 10, generator=gen).to(DEV)
    torch.manual_seed(3)
    model = kagnn_amd.GKAN_Nodes(kind, 3, fin, 64, 10, skip=True, grid_size=5, spline_order=3, hidden_layers=2,
                                 dropout=0.0).to(DEV).train()
    res = []
    for fused in (True, False):
        monkeypatch.setattr(models, "_FUSED_EPILOGUE", fused)
        model.zero_grad()
        for bn in model.bns:
            bn.reset_running_stats()
        out = model(x, ei)
        out.backward(gout)
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                    [bn.running_var.clone() for bn in model.bns]))
    assert_close(res[0][0], res[1][0].double().cpu(), what=f"{kind} model logits fused vs unfused epilogue")
    for k in res[0][1]:
        assert_close(res[0][1][k], res[1][1][k].double().cpu(), 2e-4, what=f"{kind} model grad {k} fused vs unfused", elementwise=False)
    for a, c in zip(res[0][2], res[1][2]):
        assert_close(a, c.double().cpu(), what="running_var")


def test_node_model_trains_with_fused_dropout():
    """p = 0.3: the mask comes from the normalising kernel (no aten dropout launches), differs between steps, eval mode
    is deterministic, and 25 optimiser steps reduce the loss"""
    n, e, fin, classes = 20000, 150000, 32, 7
    ei = orc.powerlaw_graph(n, e, seed=5).to(DEV)
    gen = torch.Generator().manual_seed(9)
    y = torch.randint(0, classes, (n,), generator=gen)
    x = (torch.randn(n, fin, generator=gen) * 0.3 + F.one_hot(y, fin).float()).to(DEV)
    y = y.to(DEV)
    torch.manual_seed(4)
    model = kagnn_amd.GKAN_Nodes("gin", 2, fin, 64, classes, skip=True, grid_size=5, hidden_layers=2, dropout=0.3).to(DEV).train()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        a = model(x, ei)
        torch.cuda.synchronize()
    names = [ev.key for ev in prof.key_averages()]
    assert not any("dropout" in k.lower() or "bernoulli" in k.lower() for k in names), names
    assert any("bn_apply_from_moments_kernel" in k for k in names) and not any("bn_colsum_kernel<0>" in k for k in names), names
    b = model(x, ei)
    assert not torch.equal(a, b)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    losses = []
    for _ in range(25):
        opt.zero_grad()
        loss = F.cross_entropy(model(x, ei), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] - 0.4 and all(np.isfinite(losses)), losses
    model.eval()
    assert torch.equal(model(x, ei), model(x, ei))


# ------------------------------------------------------------------ the norm's forward pass folded into its consumers (round 4)
@pytest.mark.parametrize("f", [64, 128, 32, 8, 40])
def test_aggregation_of_a_folded_norm_equals_the_aggregation_of_the_normalised_rows(f):
    """kagnn_aggregate_sum_affine: a * (self * x_i + sum_j x_j) + (self + deg_i) * b  against the aggregation of the matrix
    a * x + b written out -- hub rows (segment partial sums scaled in the merge), isolated rows (deg 0: self term only), both
    directions"""
    n, e = 20011, 150000
    ei = orc.powerlaw_graph(n, e, seed=10)
    g = ops.GraphIndex(ei.to(DEV), n)
    assert g.num_hub_seg > 0
    gen = torch.Generator().manual_seed(f)
    x = (torch.randn(n, f, generator=gen) * 3.0 + 5.0).to(DEV)
    aff = torch.stack([torch.rand(f, generator=gen) + 0.5, torch.randn(f, generator=gen)]).to(DEV)
    h = torch.addcmul(aff[1], x, aff[0])
    for tr in (False, True):
        want = ops._aggregate_raw(h, g, tr, 1.0, None, None, None, None, False)
        got = ops.aggregate_sum_affine(x, g, 1.0, aff, transposed=tr)
        assert_close(got, want.double().cpu(), 2e-6, what=f"folded-norm aggregation F={f} transposed={tr}")
    want64 = orc.sum_aggregate(h.double().cpu(), ei) + 1.5 * h.double().cpu()
    assert_close(ops.aggregate_sum_affine(x, g, 1.5, aff), want64, 2e-6, what=f"folded-norm aggregation vs oracle F={f}")


def test_read_out_over_folded_norm_blocks_equals_the_materialised_blocks():
    """kagnn_kan_linear_fwd_parts_affine / _bwd_input_affine / _bwd_weight_affine: the skip read-out over blocks that exist
    only as (y, per-column affine) against the same layer over the blocks written out -- output, the gradient with respect to
    the NORMALISED block (ops.AffineRows' convention) and every parameter gradient"""
    n, out = 5003, 40
    gen = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    layer = kagnn_amd.KANLinear(192, out, grid_size=5, spline_order=3).to(DEV)
    x0 = (torch.randn(n, 64, generator=gen) * 0.5).to(DEV)
    ys = [(torch.randn(n, 64, generator=gen) * 2.0 + 1.0).to(DEV) for _ in range(2)]
    affs = [torch.stack([torch.rand(64, generator=gen) * 0.5 + 0.2, torch.randn(64, generator=gen) * 0.3]).to(DEV) for _ in range(2)]
    gy = torch.randn(n, out, generator=gen).to(DEV)
    # reference: the blocks written out
    hs = [torch.addcmul(a[1], y, a[0]).requires_grad_(True) for y, a in zip(ys, affs)]
    x0a = x0.clone().requires_grad_(True)
    ref = layer.forward_parts([x0a] + hs)
    ref.backward(gy)
    want = [ref.detach().clone(), x0a.grad.clone()] + [h.grad.clone() for h in hs] + [p.grad.clone() for p in layer.parameters()]
    layer.zero_grad()
    yr = [y.clone().requires_grad_(True) for y in ys]
    x0b = x0.clone().requires_grad_(True)
    timer = ops.EntryPointTimer()
    ops.set_timer(timer)
    try:
        got_y = layer.forward_parts([x0b] + [ops.AffineRows(y, a) for y, a in zip(yr, affs)])
        got_y.backward(gy)
    finally:
        ops.set_timer(None)
    names = [r[0] for r in timer.records]
    assert "kagnn_kan_linear_fwd_parts_affine" in names and names.count("kagnn_kan_linear_bwd_input_affine") == 2 \
        and names.count("kagnn_kan_linear_bwd_weight_affine") == 2, names
    got = [got_y.detach(), x0b.grad] + [y.grad for y in yr] + [p.grad for p in layer.parameters()]
    for a, b, what in zip(got, want, ["y", "gx0", "g_h1 (w.r.t. the normalised block)", "g_h2", "g_base", "g_spline", "g_scaler"]):
        assert_close(a, b.double().cpu(), 4e-6, what="folded-norm read-out: " + what)


def test_node_model_with_folded_norms_equals_the_model_with_normalising_passes(monkeypatch):
    """GKAN_Nodes (3 x KAN-GIN hidden 64, skip read-out, 40 classes) on a graph large enough for the one-launch read-out: with
    the norms folded into their consumers (default: no normalised matrix is written, no bn_apply pass) against
    KAGNN_LAZY_NORM=0 -- logits, input gradient, every parameter gradient, the running statistics"""
    from kagnn_amd import models as M
    n, e = 131072 + 5, 900000
    ei = orc.powerlaw_graph(n, e, seed=2).to(DEV)
    g = ops.GraphIndex(ei, n)
    x = (torch.randn(n, 64, generator=torch.Generator().manual_seed(1)) * 0.5).to(DEV)
    gout = (torch.randn(n, 40, generator=torch.Generator().manual_seed(2)) / n).to(DEV)
    torch.manual_seed(3)
    model = kagnn_amd.GKAN_Nodes("gin", 3, 64, 64, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2).to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    res = []
    for lazy in (True, False):
        monkeypatch.setattr(M, "_LAZY_NORM", lazy)
        model.load_state_dict(state)
        model.zero_grad()
        xr = x.clone().requires_grad_(True)
        timer = ops.EntryPointTimer()
        ops.set_timer(timer)
        try:
            out = model(xr, g)
            out.backward(gout)
        finally:
            ops.set_timer(None)
        names = [r[0] for r in timer.records]
        assert names.count("kagnn_batchnorm_stats_affine") == (3 if lazy else 0), names
        assert names.count("kagnn_batchnorm_fwd") == (0 if lazy else 3), names
        assert names.count("kagnn_gin_kan_layer_fwd_affine") == (2 if lazy else 0), names
        res.append(([out.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None],
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    names = ["logits", "gx"] + [k for k, p in model.named_parameters() if p.grad is not None]
    for a, b, what in zip(res[0][0], res[1][0], names):
        scale = max(1e-30, float(b.abs().max()))
        # (2e-5 of the largest element; the norms' bias gradients are cancelling sums over 131k rows of terms ~1e3 times their
        # result -- both forms carry that noise: 1e-4)
        tol = 1e-4 if what.startswith("bns.") else 2e-5
        assert float((a - b).abs().max()) <= tol * scale, (what, float((a - b).abs().max()) / scale)
    for k in res[0][1]:                                           # running statistics (layers > 0 see inputs that differ by rounding)
        a, b = res[0][1][k].float(), res[1][1][k].float()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), k


def test_folded_norms_over_random_model_configurations(monkeypatch):
    """the same comparison (norms folded into their consumers vs normalising passes) over random widths / depths / row counts /
    class counts / chain lengths, incl. configurations the fold does not cover (narrow hidden widths, > 64 classes, dropout
    on: the model must fall back per layer or as a whole and still agree).  Tolerance: 1e-4 of each tensor's largest element
    with a floor at 1e-6 of the largest gradient in the model (cancelling sums such as the norms' bias gradients)."""
    import copy
    import random
    from kagnn_amd import models as M
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS", 0)
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS_ONE_LAUNCH", 0)
    rng = random.Random(4)
    folded = 0
    for case in range(10):
        f_in = rng.choice([64, 128])
        hidden = rng.choice([64, 64, 128, 32])
        mp, hl = rng.choice([1, 2, 3]), rng.choice([1, 2])
        n = rng.choice([700, 3001, 9000])
        p_drop = rng.choice([0.0, 0.0, 0.0, 0.25])
        classes = rng.choice([7, 40, 64, 70])
        g = ops.GraphIndex(orc.powerlaw_graph(n, 8 * n, seed=case).to(DEV), n)
        x = (torch.randn(n, f_in, generator=torch.Generator().manual_seed(case)) * 0.4).to(DEV)
        y = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(case + 1)).to(DEV)
        torch.manual_seed(case)
        model0 = kagnn_amd.GKAN_Nodes("gin", mp, f_in, hidden, classes, skip=True, grid_size=5, spline_order=3, hidden_layers=hl,
                                      dropout=p_drop).to(DEV)
        res = []
        for lazy in (True, False):
            monkeypatch.setattr(M, "_LAZY_NORM", lazy)
            model = copy.deepcopy(model0)
            xr = x.clone().requires_grad_(case % 2 == 0)
            torch.manual_seed(1000 + case)
            timer = ops.EntryPointTimer()
            ops.set_timer(timer)
            try:
                loss = ops.softmax_cross_entropy(model(xr, g), y)
                loss.backward()
            finally:
                ops.set_timer(None)
            if lazy:
                folded += sum(1 for r in timer.records if r[0] == "kagnn_batchnorm_stats_affine")
            res.append([loss.detach().clone()] + ([xr.grad.clone()] if xr.requires_grad else [])
                       + [p.grad.clone() for p in model.parameters()] + [b.clone() for b in model.buffers() if b.dtype.is_floating_point])
        label = f"case {case}: f_in {f_in} hidden {hidden} mp {mp} chain {hl} n {n} dropout {p_drop} classes {classes}"
        floor = 1e-6 * max(float(t.abs().max()) for t in res[1][1:])
        for k, (a, b) in enumerate(zip(*res)):
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= max(1e-4 * scale, floor), (label, k, float((a - b).abs().max()), scale)
    assert folded >= 5                                  # (the covered configurations really took the folded path)


# ------------------------------------------------------------------ the norm's backward statistics travelling with the gradient
def _node_model_grads(model, x, g, gout, state):
    model.load_state_dict(state)
    model.zero_grad()
    xr = x.clone().requires_grad_(True)
    timer = ops.EntryPointTimer()
    ops.set_timer(timer)
    try:
        with ops.LibraryStageTimer(None):
            out = model(xr, g)
            out.backward(gout)
            torch.cuda.synchronize()
        stages = ops.LibraryStageTimer.collect()
    finally:
        ops.set_timer(None)
    names = [r[0] for r in timer.records]
    tensors = [out.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
    return tensors, names, stages


def test_norm_backward_statistics_come_with_the_gradient_from_the_next_layers_aggregation(monkeypatch):
    """GKAN_Nodes (3 x KAN-GIN hidden 64, skip read-out) with the norms folded: the gradient arriving at layer l's norm is
    written by layer l+1's transposed aggregation (+ the skip gradient), whose epilogue now also leaves sum g and sum g * xhat
    (kagnn_gin_kan_layer_bwd_bn_sums; hub rows from the merge kernel), and the last norm's gradient comes from the read-out's
    input-gradient kernel, which leaves the same two sums (kagnn_kan_linear_bwd_input_affine_sums) -- no norm runs a statistics pass.
    Against KAGNN_FOLD_NORM_STATS=0 (the pass over g and y): every gradient, on a power-law graph with hub rows and a row count
    that is not a multiple of the 16-row workgroups."""
    n, e = 131072 + 13, 650_000
    ei = orc.powerlaw_graph(n, e, seed=7)
    ei = torch.cat([ei, ei.flip(0)], dim=1).to(DEV)              # both directions power-law: the TRANSPOSED structure has hub rows too
    g = ops.GraphIndex(ei, n)
    assert g.num_hub_seg_t > 0 and g.num_hub_seg > 0
    x = (torch.randn(n, 64, generator=torch.Generator().manual_seed(1)) * 0.5).to(DEV)
    gout = (torch.randn(n, 40, generator=torch.Generator().manual_seed(2)) / n).to(DEV)
    torch.manual_seed(5)
    model = kagnn_amd.GKAN_Nodes("gin", 3, 64, 64, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2).to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    GIVEN, PASS = "kagnn_batchnorm_bwd statistics given (in ..._layer_bwd_bn)", "kagnn_batchnorm_bwd statistics (in ..._layer_bwd_bn)"
    res = {}
    for fold in (True, False):
        monkeypatch.setattr(ops, "_FOLD_NORM_STATS", fold)
        tensors, names, stages = _node_model_grads(model, x, g, gout, state)
        launches = {k: v["launches"] for k, v in stages.items()}
        if fold:
            assert names.count("kagnn_gin_kan_layer_bwd_bn_sums") == 3 and names.count("kagnn_gin_kan_layer_bwd_bn") == 0, names
            # (the last norm's gradient comes from the read-out's input-gradient kernel, which makes its sums too:
            # kagnn_kan_linear_bwd_input_affine_sums -- no statistics pass is left)
            assert names.count("kagnn_kan_linear_bwd_input_affine_sums") == 1, names
            assert launches.get(GIVEN) == 3 and PASS not in launches, launches
            assert launches.get("kagnn_batchnorm_bwd statistics fold") == 3, launches
        else:
            assert names.count("kagnn_gin_kan_layer_bwd_bn") == 3 and GIVEN not in launches and launches.get(PASS) == 3, (names, launches)
        res[fold] = tensors
    names = ["logits", "gx"] + [k for k, p in model.named_parameters() if p.grad is not None]
    for a, b, what in zip(res[True], res[False], names):
        scale = max(1e-30, float(b.abs().max()))
        tol = 1e-4 if what.startswith("bns.") else 2e-5          # (cancelling sums over 131k rows: see the folded-norm test above)
        assert float((a - b).abs().max()) <= tol * scale, (what, float((a - b).abs().max()) / scale)
    assert torch.equal(res[True][0], res[False][0])               # the forward does not change


def test_norm_backward_statistics_are_not_taken_when_the_gradient_is_not_the_parked_tensor(monkeypatch):
    """the side channel only applies when the gradient a norm receives IS the tensor the next layer's aggregation wrote.  With
    the skip gradients travelling on the tape (KAGNN_SKIP_GRADIENT off) autograd sums the read-out's and the convolution's
    gradients of h_l into a NEW tensor: the parked sums describe only one addend and must be dropped -- same results as without
    the fold, and the statistics pass runs for every norm."""
    from kagnn_amd import models as M
    n, e = 20011, 180000
    g = ops.GraphIndex(orc.powerlaw_graph(n, e, seed=8).to(DEV), n)
    x = (torch.randn(n, 64, generator=torch.Generator().manual_seed(3)) * 0.5).to(DEV)
    gout = (torch.randn(n, 40, generator=torch.Generator().manual_seed(4)) / n).to(DEV)
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS", 0)
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS_ONE_LAUNCH", 0)
    monkeypatch.setattr(M, "_SKIP_GRADIENT", False)
    torch.manual_seed(6)
    model = kagnn_amd.GKAN_Nodes("gin", 3, 64, 64, 40, skip=True, grid_size=5, spline_order=3, hidden_layers=2).to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    PASS = "kagnn_batchnorm_bwd statistics (in ..._layer_bwd_bn)"
    res = {}
    for fold in (True, False):
        monkeypatch.setattr(ops, "_FOLD_NORM_STATS", fold)
        tensors, names, stages = _node_model_grads(model, x, g, gout, state)
        assert stages[PASS]["launches"] == 3, stages.keys()
        res[fold] = tensors
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)                                  # (the aggregation's result rows do not depend on the statistics side)


@pytest.mark.parametrize("hidden", [40, 128, 200])
def test_norm_backward_statistics_fold_at_other_widths(hidden, monkeypatch):
    """the statistics side of the aggregation at 40 / 128 / 200 columns (lane groups of 16 / 32 / 64 per row, partly idle lanes),
    hub rows in both directions, few rows: fold on vs off, every gradient"""
    from kagnn_amd import models as M
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS", 0)
    monkeypatch.setattr(M, "_SPLIT_READOUT_MIN_ROWS_ONE_LAUNCH", 0)
    n, e = 5003, 40000
    ei = orc.powerlaw_graph(n, e, seed=hidden)
    g = ops.GraphIndex(torch.cat([ei, ei.flip(0)], dim=1).to(DEV), n)
    assert g.num_hub_seg_t > 0
    x = (torch.randn(n, 64, generator=torch.Generator().manual_seed(hidden)) * 0.5).to(DEV)
    gout = (torch.randn(n, 10, generator=torch.Generator().manual_seed(hidden + 1)) / n).to(DEV)
    torch.manual_seed(hidden)
    model = kagnn_amd.GKAN_Nodes("gin", 3, 64, hidden, 10, skip=True, grid_size=5, spline_order=3, hidden_layers=1).to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    res = {}
    for fold in (True, False):
        monkeypatch.setattr(ops, "_FOLD_NORM_STATS", fold)
        tensors, names, stages = _node_model_grads(model, x, g, gout, state)
        if fold:
            # (hidden > 64: the first convolution's output is wider than its inputs, it stays on two tape nodes -- one fold fewer)
            want = 2 if hidden <= 64 else 1
            assert names.count("kagnn_gin_kan_layer_bwd_bn_sums") >= want, names
            assert stages.get("kagnn_batchnorm_bwd statistics fold", {}).get("launches") == want, stages.keys()
        res[fold] = tensors
    names = ["logits", "gx"] + [k for k, p in model.named_parameters() if p.grad is not None]
    floor = 1e-6 * max(float(t.abs().max()) for t in res[False][1:])
    for a, b, what in zip(res[True], res[False], names):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= max(1e-4 * scale, floor), (what, float((a - b).abs().max()), scale)
