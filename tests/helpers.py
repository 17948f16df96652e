"""Shared helpers of the parity tests: tolerances, state loading, oracle drivers."""
import numpy as np
import torch

from oracle import kan_oracle as orc

TOL = 1e-4   # BASELINE.json north_star: "within 1e-4 fp32" -- relative to max(1, max|reference|)
KAN_KEYS = ("base_weight", "spline_weight", "spline_scaler", "grid")
FK_KEYS = ("layernorm.weight", "layernorm.bias", "rbf.grid", "spline_linear.weight",
           "base_linear.weight", "base_linear.bias")


def T(a, device=None):
    t = torch.from_numpy(np.asarray(a))
    return t if device is None else t.to(device)


def assert_close(got, want, tol=TOL, what=""):
    got = got.detach().double().cpu()
    want = T(want).double() if not torch.is_tensor(want) else want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.equal(torch.isnan(got), torch.isnan(want)), f"{what}: NaN pattern differs"
    g, w = torch.nan_to_num(got), torch.nan_to_num(want)
    scale = max(1.0, float(w.abs().max())) if w.numel() else 1.0
    err = float((g - w).abs().max()) if w.numel() else 0.0
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:.0e} * {scale:.3g}"
    return err / scale


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
