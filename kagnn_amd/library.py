"""``torch.library`` registration of the hot-path entry points ("registered via torch.library so
autograd / torch.compile see opaque ops").

Eager code keeps going through the ``torch.autograd.Function`` glue of ``kagnn_amd.ops`` (a ``custom_op`` call costs
tens of microseconds of dispatcher work, which matters on Cora-sized graphs); when dynamo is tracing
(``torch.compiler.is_compiling()``) the public functions of ``ops`` route here instead, so a compiled
``GKAN_Nodes`` / ``GFASTKAN_Nodes`` traces into ONE graph whose nodes are ``kagnn::*`` ops -- no graph breaks on
ctypes calls.  Both paths end in the same raw functions, i.e. the same ``libkagnn_hip.so`` kernels.

Ops (all fp32, CUDA only -- there is no CPU kernel to register):
  kagnn::aggregate_sum            self-adjoint: its backward is the same op on the transposed CSR
  kagnn::kan_linear (+ _bwd_input, _bwd_weight)
  kagnn::fastkan_layer (+ _bwd)
  kagnn::batch_norm (+ _bwd)      functional; the running statistics are updated by traced aten ops next to it
  kagnn::segment_pool / kagnn::segment_broadcast
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops as _ops

_lib = torch.library


def _maybe(t: Tensor) -> Optional[Tensor]:
    """custom ops cannot return Optional tensors: absent gradients travel as 0-element tensors"""
    return None if t.numel() == 0 else t


def _empty(like: Tensor) -> Tensor:
    return like.new_empty((0,))


# ---------------------------------------------------------------------------------------- aggregation
@_lib.custom_op("kagnn::aggregate_sum", mutates_args=())
def aggregate_sum(x: Tensor, rowptr: Tensor, col: Tensor, edge_weight: Optional[Tensor], in_scale: Optional[Tensor],
                  out_scale: Optional[Tensor], bias: Optional[Tensor], hub_seg: Optional[Tensor], num_hub_seg: int,
                  hub_threshold: int, self_scale: float, skip_self: bool) -> Tensor:
    with _ops._device_of(x):
        return _ops._aggregate_csr(_ops._rows(x), rowptr, col, hub_seg, num_hub_seg, hub_threshold, self_scale,
                                   edge_weight, in_scale, out_scale, bias, skip_self)


@aggregate_sum.register_fake
def _(x, rowptr, col, edge_weight, in_scale, out_scale, bias, hub_seg, num_hub_seg, hub_threshold, self_scale, skip_self):
    return x.new_empty(x.shape)


# ---------------------------------------------------------------------------------------- KANLinear
@_lib.custom_op("kagnn::kan_linear", mutates_args=())
def kan_linear(x: Tensor, base_weight: Optional[Tensor], spline_weight: Tensor, spline_scaler: Optional[Tensor],
               knots: Tensor, grid_size: int, spline_order: int, mode: int) -> Tuple[Tensor, Tensor]:
    with _ops._device_of(x):
        bw = None if base_weight is None else base_weight.contiguous()
        sc = None if spline_scaler is None else spline_scaler.contiguous()
        return _ops._kan_fwd_raw(_ops._rows(x), bw, spline_weight.contiguous(), sc, knots, grid_size, spline_order, mode)


@kan_linear.register_fake
def _(x, base_weight, spline_weight, spline_scaler, knots, grid_size, spline_order, mode):
    _, db = _ops._sizes("kagnn_kan_pack_bytes", int(x.size(1)), int(spline_weight.size(0)), grid_size, spline_order, mode,
                        outputs=2)                       # a pure function of the integer shape: no device needed
    return x.new_empty((x.size(0), spline_weight.size(0))), x.new_empty((max(int(db), 16),), dtype=torch.uint8)


@_lib.custom_op("kagnn::kan_linear_bwd_input", mutates_args=())
def kan_linear_bwd_input(x: Tensor, gy: Tensor, knots: Tensor, pack_dx: Tensor, out_features: int, grid_size: int,
                         spline_order: int, mode: int) -> Tensor:
    with _ops._device_of(x):
        x = _ops._rows(x)
        return _ops._kan_bwd_input_raw(x, _ops._rows(gy), knots, pack_dx, x.size(1), out_features, grid_size, spline_order, mode)


@kan_linear_bwd_input.register_fake
def _(x, gy, knots, pack_dx, out_features, grid_size, spline_order, mode):
    return x.new_empty(x.shape)


@_lib.custom_op("kagnn::kan_linear_bwd_weight", mutates_args=())
def kan_linear_bwd_weight(x: Tensor, gy: Tensor, knots: Tensor, spline_weight: Tensor, spline_scaler: Optional[Tensor],
                          grid_size: int, spline_order: int, mode: int, has_base: bool) -> Tuple[Tensor, Tensor, Tensor]:
    with _ops._device_of(x):
        x = _ops._rows(x)
        fout, fin = spline_weight.size(0), spline_weight.size(1)
        sc = None if spline_scaler is None else spline_scaler.contiguous()
        gbw, gsw, gsc = _ops._kan_bwd_weight_raw(x, _ops._rows(gy), knots, spline_weight.contiguous(), sc, fin, fout,
                                                 grid_size, spline_order, mode, has_base)
        return (gbw if gbw is not None else _empty(x)), gsw, (gsc if gsc is not None else _empty(x))


@kan_linear_bwd_weight.register_fake
def _(x, gy, knots, spline_weight, spline_scaler, grid_size, spline_order, mode, has_base):
    fout, fin = spline_weight.size(0), spline_weight.size(1)
    return (x.new_empty((fout, fin)) if has_base else x.new_empty((0,)), x.new_empty(spline_weight.shape),
            x.new_empty((fout, fin)) if spline_scaler is not None else x.new_empty((0,)))


def _kan_setup(ctx, inputs, output):
    x, bw, sw, sc, knots, G, K, mode = inputs
    ctx.save_for_backward(x, sw, sc, knots, output[1])
    ctx.meta = (sw.size(0), G, K, mode, bw is not None)


def _kan_backward(ctx, gy, _gpack):
    x, sw, sc, knots, pack_d = ctx.saved_tensors
    fout, G, K, mode, has_base = ctx.meta
    gx = gbw = gsw = gsc = None
    if ctx.needs_input_grad[0]:
        gx = kan_linear_bwd_input(x, gy, knots, pack_d, fout, G, K, mode)
    if any(ctx.needs_input_grad[1:4]):
        gbw, gsw, gsc = kan_linear_bwd_weight(x, gy, knots, sw, sc, G, K, mode, has_base)
        gbw, gsc = (gbw if has_base else None), (gsc if sc is not None else None)
    return gx, gbw, gsw, gsc, None, None, None, None


kan_linear.register_autograd(_kan_backward, setup_context=_kan_setup)


# ---------------------------------------------------------------------------------------- FastKAN layer
@_lib.custom_op("kagnn::fastkan_layer", mutates_args=())
def fastkan_layer(x: Tensor, ln_weight: Optional[Tensor], ln_bias: Optional[Tensor], spline_weight: Tensor,
                  base_weight: Optional[Tensor], base_bias: Optional[Tensor], centers: Tensor, denominator: float,
                  ln_eps: float, mode: int) -> Tuple[Tensor, Tensor]:
    with _ops._device_of(x):
        y, stats = _ops._fastkan_fwd_raw(_ops._rows(x), ln_weight, ln_bias, spline_weight, base_weight, base_bias, centers,
                                         denominator, ln_eps, mode)
        return y, (stats if stats is not None else _empty(x))


@fastkan_layer.register_fake
def _(x, ln_weight, ln_bias, spline_weight, base_weight, base_bias, centers, denominator, ln_eps, mode):
    return (x.new_empty((x.size(0), spline_weight.size(0))),
            x.new_empty((x.size(0), 2)) if ln_weight is not None else x.new_empty((0,)))


@_lib.custom_op("kagnn::fastkan_layer_bwd", mutates_args=())
def fastkan_layer_bwd(x: Tensor, gy: Tensor, ln_weight: Optional[Tensor], ln_bias: Optional[Tensor], spline_weight: Tensor,
                      base_weight: Optional[Tensor], centers: Tensor, stats: Tensor, denominator: float, ln_eps: float,
                      mode: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    with _ops._device_of(x):
        out = _ops._fastkan_bwd_raw(_ops._rows(x), _ops._rows(gy), ln_weight, ln_bias, spline_weight, base_weight, centers,
                                    _maybe(stats), denominator, ln_eps, mode)
        return tuple(t if t is not None else _empty(x) for t in out)


@fastkan_layer_bwd.register_fake
def _(x, gy, ln_weight, ln_bias, spline_weight, base_weight, centers, stats, denominator, ln_eps, mode):
    fin, fout = x.size(1), spline_weight.size(0)
    e = x.new_empty((0,))
    ln = ln_weight is not None
    has_b = base_weight is not None
    return (x.new_empty(x.shape), x.new_empty((fin,)) if ln else e, x.new_empty((fin,)) if ln else e,
            x.new_empty(spline_weight.shape), x.new_empty((fout, fin)) if has_b else e, x.new_empty((fout,)) if has_b else e)


def _fk_setup(ctx, inputs, output):
    x, lw, lb, sw, bw, bb, centers, den, eps, mode = inputs
    ctx.save_for_backward(x, lw, lb, sw, bw, centers, output[1])
    ctx.meta = (den, eps, mode, bb is not None)


def _fk_backward(ctx, gy, _gstats):
    x, lw, lb, sw, bw, centers, stats = ctx.saved_tensors
    den, eps, mode, has_bb = ctx.meta
    gx, glw, glb, gsw, gbw, gbb = fastkan_layer_bwd(x, gy, lw, lb, sw, bw, centers, stats, den, eps, mode)
    return (gx, glw if lw is not None else None, glb if lw is not None else None, gsw, gbw if bw is not None else None,
            gbb if has_bb else None, None, None, None, None)


fastkan_layer.register_autograd(_fk_backward, setup_context=_fk_setup)


# ---------------------------------------------------------------------------------------- BatchNorm1d
@_lib.custom_op("kagnn::batch_norm", mutates_args=())
def batch_norm(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], running_mean: Optional[Tensor],
               running_var: Optional[Tensor], training: bool, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """functional form (an op that mutates cannot carry an autograd formula): in training mode the running statistics
    are NOT touched here -- ``batch_norm_traced`` below updates them with two traced aten ops from the returned batch
    mean / rstd; in eval mode they are only read"""
    with _ops._device_of(x):
        rm, rv = (None, None) if training else (running_mean, running_var)
        return _ops._batchnorm_fwd_raw(_ops._rows(x), weight, bias, rm, rv, training, 0.0, eps)


@batch_norm.register_fake
def _(x, weight, bias, running_mean, running_var, training, eps):
    return x.new_empty(x.shape), x.new_empty((x.size(1),)), x.new_empty((x.size(1),))


@_lib.custom_op("kagnn::batch_norm_bwd", mutates_args=())
def batch_norm_bwd(x: Tensor, gy: Tensor, weight: Optional[Tensor], mean: Tensor, rstd: Tensor, training: bool,
                   want_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    with _ops._device_of(x):
        gx, gw, gb = _ops._batchnorm_bwd_raw(_ops._rows(x), _ops._rows(gy), weight, mean, rstd, training, True, want_bias)
        return gx, (gw if gw is not None else _empty(x)), (gb if gb is not None else _empty(x))


@batch_norm_bwd.register_fake
def _(x, gy, weight, mean, rstd, training, want_bias):
    f = x.size(1)
    return (x.new_empty(x.shape), x.new_empty((f,)) if weight is not None else x.new_empty((0,)),
            x.new_empty((f,)) if want_bias else x.new_empty((0,)))


def _bn_setup(ctx, inputs, output):
    x, w, b, rm, rv, training, eps = inputs
    ctx.save_for_backward(x, w, output[1], output[2])
    ctx.meta = (training, b is not None)


def _bn_backward(ctx, gy, _gm, _gr):
    x, w, mean, rstd = ctx.saved_tensors
    training, has_bias = ctx.meta
    gx, gw, gb = batch_norm_bwd(x, gy, w, mean, rstd, training, has_bias)
    return gx, (gw if w is not None else None), (gb if has_bias else None), None, None, None, None


batch_norm.register_autograd(_bn_backward, setup_context=_bn_setup)


def batch_norm_traced(x, weight, bias, running_mean, running_var, training, momentum, eps):
    y, mean, rstd = batch_norm(x, weight, bias, running_mean, running_var, bool(training), float(eps))
    if training and running_mean is not None and running_var is not None:
        n = x.size(0)
        with torch.no_grad():            # torch.nn.BatchNorm1d: running_var tracks the UNBIASED batch variance
            var = (1.0 / (rstd * rstd) - eps) * (n / max(n - 1, 1))
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1.0 - momentum).add_(var, alpha=momentum)
    return y


# ---------------------------------------------------------------------------------------- pooling
@_lib.custom_op("kagnn::segment_pool", mutates_args=())
def segment_pool(x: Tensor, seg_ptr: Tensor, mean: bool) -> Tensor:
    with _ops._device_of(x):
        return _ops._segment_pool_raw(_ops._rows(x), seg_ptr, mean)


@segment_pool.register_fake
def _(x, seg_ptr, mean):
    return x.new_empty((seg_ptr.numel() - 1, x.size(1)))


@_lib.custom_op("kagnn::segment_broadcast", mutates_args=())
def segment_broadcast(gout: Tensor, seg_ptr: Tensor, num_rows: int, mean: bool) -> Tensor:
    with _ops._device_of(gout):
        return _ops._segment_broadcast_raw(_ops._rows(gout), seg_ptr, num_rows, mean)


@segment_broadcast.register_fake
def _(gout, seg_ptr, num_rows, mean):
    return gout.new_empty((num_rows, gout.size(1)))


def _pool_setup(ctx, inputs, output):
    x, seg, mean = inputs
    ctx.seg, ctx.mean, ctx.n = seg, mean, x.size(0)


def _pool_backward(ctx, g):
    return segment_broadcast(g, ctx.seg, ctx.n, ctx.mean), None, None


segment_pool.register_autograd(_pool_backward, setup_context=_pool_setup)


# ---------------------------------------------------------------------------------------- graph-level wrappers
def aggregate(x, g, self_scale, edge_weight, in_scale, out_scale, bias, skip_self):
    """``ops.aggregate_sum`` for traced code: the op is applied on the by-destination CSR; its autograd formula (below)
    applies the same op on the transposed CSR with the scales swapped."""
    return _AggregateBoth.apply(x, bias, g, float(self_scale), edge_weight, in_scale, out_scale, bool(skip_self))


class _AggregateBoth(torch.autograd.Function):
    """thin tape node around the opaque op: it only chooses which side of the GraphIndex the op sees"""

    @staticmethod
    def forward(ctx, x, bias, g, self_scale, edge_weight, in_scale, out_scale, skip_self):
        ctx.g, ctx.self_scale, ctx.skip_self = g, self_scale, skip_self
        ctx.in_scale, ctx.out_scale = in_scale, out_scale
        ctx.ew_t = None
        ew = None
        if edge_weight is not None:
            w = edge_weight.to(torch.float32)
            ctx.ew_t = w[g.perm_t.long()].contiguous()
            ew = w[g.perm.long()].contiguous()
        return aggregate_sum(x, g.rowptr, g.col, ew, in_scale, out_scale, bias, g.hub_seg if g.num_hub_seg else None,
                             g.num_hub_seg, g.hub_threshold, self_scale, skip_self)

    @staticmethod
    def backward(ctx, gout):
        g = ctx.g
        gx = gb = None
        if ctx.needs_input_grad[0]:
            gx = aggregate_sum(gout, g.rowptr_t, g.col_t, ctx.ew_t, ctx.out_scale, ctx.in_scale, None,
                               g.hub_seg_t if g.num_hub_seg_t else None, g.num_hub_seg_t, g.hub_threshold,
                               ctx.self_scale, ctx.skip_self)
        if ctx.needs_input_grad[1]:
            gb = gout.sum(0)
        return gx, gb, None, None, None, None, None, None
