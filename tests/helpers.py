"""Shared helpers of the parity tests: tolerances, state loading, oracle drivers.

Tolerances.  BASELINE.json's contract is "within 1e-4 fp32" (``CONTRACT``).  The tests pin an order of magnitude
tighter: ``TOL`` = 2e-5 relative to ``max(1, max|reference|)`` for every element -- about 10x the error both
precision modes actually show on the layer-level cases (``profiles/r02_parity_errors.json``), so a regression of the
fp16 hi/lo split path (a dropped product, a wrong scale exponent: >= 2^-11) cannot hide behind the contract figure.
On top of the max-norm bound every element at or above ``REL_FLOOR`` x scale must agree to ``CONTRACT`` RELATIVE to
its own magnitude.  Tests that need more room (errors compounded through BatchNorm / several layers / Adam) pass an
explicit ``tol`` and say why.
"""
import json
import os

import numpy as np
import torch

from oracle import kan_oracle as orc

CONTRACT = 1e-4   # BASELINE.json north_star: "within 1e-4 fp32"
TOL = 2e-5        # what the tests hold both precision modes to (max-norm, relative to max(1, max|reference|))
REL_FLOOR = 0.1   # elements >= REL_FLOOR * scale are also checked relative to their own magnitude (CONTRACT)
KAN_KEYS = ("base_weight", "spline_weight", "spline_scaler", "grid")
FK_KEYS = ("layernorm.weight", "layernorm.bias", "rbf.grid", "spline_linear.weight",
           "base_linear.weight", "base_linear.bias")

ERROR_LOG = []    # (what, observed error / scale, tolerance): dumped by conftest.py at session end


def T(a, device=None):
    t = torch.from_numpy(np.asarray(a))
    return t if device is None else t.to(device)


def assert_close(got, want, tol=TOL, what="", elementwise=True):
    got = got.detach().double().cpu()
    want = T(want).double() if not torch.is_tensor(want) else want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.equal(torch.isnan(got), torch.isnan(want)), f"{what}: NaN pattern differs"
    if not want.numel():
        return 0.0
    g, w = torch.nan_to_num(got), torch.nan_to_num(want)
    scale = max(1.0, float(w.abs().max()))
    diff = (g - w).abs()
    err = float(diff.max())
    ERROR_LOG.append((what, err / scale, tol))
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:.0e} * {scale:.3g}"
    big = w.abs() >= REL_FLOOR * scale
    if elementwise and bool(big.any()):
        rel = float((diff[big] / w.abs()[big]).max())
        assert rel <= max(CONTRACT, 5 * tol), f"{what}: element-wise relative error {rel:.3e} on a large element"
    return err / scale


def dump_error_log(path):
    if not ERROR_LOG:
        return
    worst = {}
    for what, err, tol in ERROR_LOG:
        key = what
        if key not in worst or err > worst[key][0]:
            worst[key] = (err, tol)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"n_checks": len(ERROR_LOG), "max_ratio_to_tol": max(e / t for _, e, t in ERROR_LOG if t > 0),
                   "worst_per_label": {k: {"err": v[0], "tol": v[1]} for k, v in sorted(worst.items(), key=lambda kv: -kv[1][0])[:200]}},
                  f, indent=1)


def load_kanlinear(layer, z, prefix, device):
    sd = {k: T(z[f"{prefix}{k}"]) for k in KAN_KEYS}
    layer.load_state_dict(sd)
    return layer.to(device)


def oracle_kan_linear_fwd_bwd(x, gy, p, k, dtype=torch.float64):
    """fp64 'truth' through the oracle (same algorithm as the reference, wider arithmetic)."""
    x = x.detach().cpu().to(dtype).requires_grad_(True)
    ps = {n: (v.detach().cpu().to(dtype).requires_grad_(True) if n != "grid" else v.detach().cpu().to(dtype))
          for n, v in p.items()}
    y = orc.kan_linear_forward(x, ps["base_weight"], ps["spline_weight"], ps["spline_scaler"], ps["grid"], k)
    y.backward(gy.detach().cpu().to(dtype))
    return y.detach(), x.grad, {n: ps[n].grad for n in ("base_weight", "spline_weight", "spline_scaler")}


def oracle_node_model_fwd_bwd(x, edge_index, state, gout, arch, conv_type, mp_layers, spline_order=3, chunk=None,
                              dtype=torch.float64):
    """logits, d/dx and every parameter gradient of a GKAN_Nodes / GFASTKAN_Nodes model through the oracle
    (``oracle.node_model_forward``) in ``dtype``; ``gout`` is the upstream gradient of the logits."""
    frozen = ("grid", "rbf.grid", "eps", "running_mean", "running_var", "num_batches_tracked")
    st, leaves = {}, {}
    for k, v in state.items():
        v = v.detach().cpu()
        if v.is_floating_point():
            v = v.to(dtype)
        if k.endswith(frozen):
            st[k] = v
        else:
            leaves[k] = v.clone().requires_grad_(True)
            st[k] = leaves[k]
    xr = x.detach().cpu().to(dtype).requires_grad_(True)
    ei = edge_index
    if isinstance(ei, torch.Tensor) and ei.is_sparse:
        ei = ei.cpu().to(dtype)
    out = orc.node_model_forward(xr, ei, st, arch, conv_type, mp_layers, spline_order, chunk=chunk)
    out.backward(gout.detach().cpu().to(dtype))
    return out.detach(), xr.grad, {k: v.grad for k, v in leaves.items()}


# ---------------------------------------------------------------------------------- the bf16 gather mode, restated
class _RoundRowsBf16(torch.autograd.Function):
    """what KAGNN_ACT=bf16 does to a matrix the aggregation gathers on the way FORWARD: one round-to-nearest-even to bf16
    (the gradient passes through unchanged: the HIP path hands the transposed aggregation's output straight to x.grad)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradBf16(torch.autograd.Function):
    """... and on the way BACK: d loss / d h0, the matrix the transposed aggregation gathers, is stored as bf16"""

    @staticmethod
    def forward(ctx, h):
        return h.view_as(h)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class bf16_gather_oracle:
    """context manager: inside it ``oracle.gin_conv`` rounds the two gathered matrices of every KAN-GIN convolution to bf16
    exactly where the build-defined KAGNN_ACT=bf16 mode does -- everything else stays fp64.  The HIP
    path must agree with THIS oracle at (nearly) the fp32 tolerance: that is the parity statement for the mode; its distance
    from the unrounded oracle is a property of the mode, reported separately."""

    def __enter__(self):
        self._orig = orc.gin_conv

        def gin_conv(x, edge_index, nn_fn, eps=0.0):
            xr = _RoundRowsBf16.apply(x)
            return nn_fn(_RoundGradBf16.apply(orc.sum_aggregate(xr, edge_index) + (1.0 + eps) * xr))
        orc.gin_conv = gin_conv
        return self

    def __exit__(self, *exc):
        orc.gin_conv = self._orig
        return False
