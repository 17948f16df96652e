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
    assert_close(out, want, 2e-3, what="bn of a near-constant column", elementwise=False)   # rstd ~ 1e2 amplifies fp32 ulps of y


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
    assert any("bn_apply_kernel" in k for k in names) and not any("bn_colsum_kernel<0>" in k for k in names), names
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
