"""Feature-sharded KAN-GIN layer over the GPUs of one node (RCCL through torch.distributed).

The reference has no multi-GPU code at all; this is the scheme BASELINE.json's
north_star asks for: every rank keeps the whole graph structure (CSR, replicated) and 1/P of the
feature columns of every activation, plus the matching input-feature slice of each KANLinear's
``base_weight / spline_weight / spline_scaler``.  Neighbour aggregation is column-independent, the
basis expansion is per scalar, and the contraction over input features splits into per-rank partial
sums ``[N, out]``; ONE collective per KANLinear closes it:

    forward :  y[:, my columns] = reduce_scatter(partial)  (sum over ranks, scattered along `out`)
    backward:  d partial = all_gather(d y[:, my columns])          (no other communication)

Parameters are sharded, so their gradients are local.  ``local_ops`` exists so the communication
logic can be exercised on CPU with gloo by the test-suite (which injects the oracle there); the
default -- and the only thing the product uses -- is ``kagnn_amd.ops`` (HIP kernels, no fallback).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import nn
from torch.autograd import Function

from . import ops as _hip_ops
from .ekan import KANLinear
from .models import GIKANLayer


class _ReduceScatterColumns(Function):
    """y_shard = (sum over ranks of partial)[:, lo:hi] as ONE reduce-scatter (each rank receives only its
    ``out/P`` columns: half the wire traffic of all-reduce + slice); backward all-gathers the column shards."""

    @staticmethod
    def forward(ctx, partial, group, lo, hi):
        world = dist.get_world_size(group)
        n, out = partial.shape
        w = out // world
        assert hi - lo == w and lo == dist.get_rank(group) * w
        # rank-major blocks [P][N][out/P] so that block p is what rank p keeps
        blocks = partial.detach().view(n, world, w).permute(1, 0, 2).contiguous().view(world * n, w)
        y = torch.empty((n, w), dtype=partial.dtype, device=partial.device)
        dist.reduce_scatter_tensor(y, blocks, op=dist.ReduceOp.SUM, group=group)
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, g_shard):
        world = dist.get_world_size(ctx.group)
        g = g_shard.contiguous()
        n, w = g.shape
        buf = torch.empty((world * n, w), dtype=g.dtype, device=g.device)     # rank-major concat
        dist.all_gather_into_tensor(buf, g, group=ctx.group)
        return buf.view(world, n, w).permute(1, 0, 2).reshape(n, world * w), None, None, None


class ShardedKANLinear(nn.Module):
    """Input-feature slice ``[lo, hi)`` of a KANLinear: ``forward`` returns this rank's partial
    sums over its features for ALL outputs."""

    def __init__(self, full: KANLinear, rank: int, world: int):
        super().__init__()
        if full.in_features % world or full.out_features % world:
            raise ValueError("in_features and out_features must be divisible by the world size")
        self.grid_size, self.spline_order = full.grid_size, full.spline_order
        self.out_features = full.out_features
        w = full.in_features // world
        self.lo, self.hi = rank * w, (rank + 1) * w
        ow = full.out_features // world
        self.out_lo, self.out_hi = rank * ow, (rank + 1) * ow
        self.base_weight = nn.Parameter(full.base_weight.detach()[:, self.lo:self.hi].clone())
        self.spline_weight = nn.Parameter(full.spline_weight.detach()[:, self.lo:self.hi].clone())
        self.spline_scaler = (nn.Parameter(full.spline_scaler.detach()[:, self.lo:self.hi].clone())
                              if full.enable_standalone_scale_spline else None)
        self.register_buffer("knots", full.grid[0].detach().clone())
        self.precision = full.precision

    def forward(self, x_shard: torch.Tensor, local_ops) -> torch.Tensor:
        return local_ops.kan_linear(x_shard, self.base_weight, self.spline_weight, self.spline_scaler,
                                    self.knots, self.grid_size, self.spline_order, self.precision)


class ShardedGIKANLayer(nn.Module):
    """``GIKANLayer`` (sum-aggregate + KAN chain) on 1/P of the feature columns per rank."""

    def __init__(self, conv: GIKANLayer, group=None, local_ops=None):
        super().__init__()
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.local_ops = _hip_ops if local_ops is None else local_ops
        self.eps = float(conv.eps)
        self.layers = nn.ModuleList(ShardedKANLinear(l, self.rank, self.world) for l in conv.nn.layers)

    def shard_columns(self, t: torch.Tensor) -> torch.Tensor:
        w = t.size(1) // self.world
        return t[:, self.rank * w:(self.rank + 1) * w].contiguous()

    def forward(self, x_shard: torch.Tensor, graph) -> torch.Tensor:
        h = self.local_ops.aggregate_sum(x_shard, graph, self_scale=1.0 + self.eps)
        for layer in self.layers:
            h = _ReduceScatterColumns.apply(layer(h, self.local_ops), self.group, layer.out_lo, layer.out_hi)
        return h
