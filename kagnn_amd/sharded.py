"""Feature-sharded KAN-GIN layer over the GPUs of one node (RCCL through torch.distributed).

The reference has no multi-GPU code at all; this is the scheme BASELINE.json's
north_star asks for: every rank keeps the whole graph structure (CSR, replicated) and 1/P of the
feature columns of every activation, plus the matching input-feature slice of each KANLinear's
``base_weight / spline_weight / spline_scaler``.  Neighbour aggregation is column-independent, the
basis expansion is per scalar, and the contraction over input features splits into per-rank partial
sums ``[N, out]``; ONE collective per KANLinear closes it:

    forward :  y[:, my columns] = reduce_scatter(partial)  (sum over ranks, scattered along `out`)
    backward:  d partial = all_gather(d y[:, my columns])          (no other communication)

Parameters are sharded, so their gradients are local.

``TransposedShardedGIKANLayer`` is the variant that scales at narrow widths.  The reduce-scatter above moves
``N*out*4*(P-1)/P`` bytes per rank and KANLinear -- at the metric's width (F=64, N=1M) four collectives of
224 MB against 2.9 ms of single-GPU compute.  The aggregation is the only part that NEEDS whole columns; the
KAN chain is row-independent.  So: aggregate on column shards ``[N, F/P]``, one all-to-all turns them into
row shards ``[N/P, F]`` (each rank sends ``N*F*4/P^2`` bytes to every peer: 28 MB per rank in total at P=8),
the whole KAN chain runs on the row shard with replicated weights, one all-to-all turns the result back into
column shards for the next convolution / BatchNorm.  Backward mirrors it; parameter gradients (a few hundred
KB) are summed with an all-reduce.  8x less wire traffic and no partial-sum buffers.

``local_ops`` exists so the communication
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


# ----------------------------------------------------------------------------------------------------------
# column shards <-> row shards (all-to-all)
def _row_splits(n: int, world: int):
    per = (n + world - 1) // world
    return [max(0, min(per, n - r * per)) for r in range(world)]


def _cols_to_rows(x_cols: torch.Tensor, group) -> torch.Tensor:
    """[N, w] (my columns, all rows) -> [n_me, P*w] (my rows, all columns; column order = rank order)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n, w = x_cols.shape
    splits = _row_splits(n, world)
    n_me = splits[rank]
    recv = torch.empty((world * n_me, w), dtype=x_cols.dtype, device=x_cols.device)
    dist.all_to_all_single(recv, x_cols.contiguous(), output_split_sizes=[n_me] * world,
                           input_split_sizes=splits, group=group)
    return recv.view(world, n_me, w).permute(1, 0, 2).reshape(n_me, world * w)


def _rows_to_cols(x_rows: torch.Tensor, n: int, group) -> torch.Tensor:
    """[n_me, P*w] (my rows, all columns) -> [N, w] (my columns, all rows)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_me, f = x_rows.shape
    w = f // world
    splits = _row_splits(n, world)
    assert n_me == splits[rank]
    send = x_rows.view(n_me, world, w).permute(1, 0, 2).contiguous().view(world * n_me, w)
    recv = torch.empty((n, w), dtype=x_rows.dtype, device=x_rows.device)
    dist.all_to_all_single(recv, send, output_split_sizes=splits, input_split_sizes=[n_me] * world, group=group)
    return recv


class _ColsToRows(Function):
    @staticmethod
    def forward(ctx, x_cols, group):
        ctx.group, ctx.n = group, x_cols.size(0)
        return _cols_to_rows(x_cols.detach(), group)

    @staticmethod
    def backward(ctx, g_rows):
        return _rows_to_cols(g_rows.contiguous(), ctx.n, ctx.group), None


class _RowsToCols(Function):
    @staticmethod
    def forward(ctx, x_rows, n, group):
        ctx.group = group
        return _rows_to_cols(x_rows.detach().contiguous(), n, group)

    @staticmethod
    def backward(ctx, g_cols):
        return _cols_to_rows(g_cols.contiguous(), ctx.group), None, None


class _SumGradAcrossRanks(Function):
    """identity on a (replicated) parameter; its gradient is summed over the ranks' row shards"""

    @staticmethod
    def forward(ctx, p, group):
        ctx.group = group
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


class TransposedShardedGIKANLayer(nn.Module):
    """``GIKANLayer`` with the aggregation on column shards and the KAN chain on row shards (see the module
    docstring).  Same interface as ``ShardedGIKANLayer``: column shard in, column shard out."""

    def __init__(self, conv: GIKANLayer, group=None, local_ops=None, sync_in_backward: bool = True):
        """``sync_in_backward=True``: every parameter's gradient is all-reduced inside autograd (one small
        collective per parameter tensor -- transparent, like DDP without buckets).  ``False``: gradients stay
        rank-local until ``sync_gradients()`` sums ALL of them with ONE flat all-reduce (what ``bench.py`` does)."""
        super().__init__()
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.local_ops = _hip_ops if local_ops is None else local_ops
        self.sync_in_backward = sync_in_backward
        self.eps = float(conv.eps)
        for l in conv.nn.layers:
            if l.in_features % self.world or l.out_features % self.world:
                raise ValueError("layer widths must be divisible by the world size")
        import copy
        self.layers = nn.ModuleList(copy.deepcopy(l) for l in conv.nn.layers)      # replicated parameters

    def sync_gradients(self) -> None:
        """sum the parameter gradients over the ranks with a single flat all-reduce (for ``sync_in_backward=False``)"""
        grads = [p.grad for p in self.parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def shard_columns(self, t: torch.Tensor) -> torch.Tensor:
        w = t.size(1) // self.world
        return t[:, self.rank * w:(self.rank + 1) * w].contiguous()

    def forward(self, x_shard: torch.Tensor, graph) -> torch.Tensor:
        n = x_shard.size(0)
        h = self.local_ops.aggregate_sum(x_shard, graph, self_scale=1.0 + self.eps)
        h = _ColsToRows.apply(h, self.group)
        for layer in self.layers:
            sc = layer.spline_scaler if layer.enable_standalone_scale_spline else None
            g = self.group
            wrap = (lambda p: _SumGradAcrossRanks.apply(p, g)) if self.sync_in_backward else (lambda p: p)
            h = self.local_ops.kan_linear(h, wrap(layer.base_weight), wrap(layer.spline_weight),
                                          None if sc is None else wrap(sc),
                                          layer.grid[0].contiguous(), layer.grid_size, layer.spline_order,
                                          layer.precision)
        return _RowsToCols.apply(h, n, self.group)
