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

Round 6 -- SURVEY.md 8(e)'s other exchanges: ``ShardedGIFASTKANLayer`` (the FastKAN flavour: LayerNorm is the one reduction over the
sharded axis, exchanged as 2 floats per row each way -- local (mean, M2) gathered and merged in rank order on the way forward, the two
row sums of its backward all-reduced beside the weight gradient), ``ShardedNodeModel`` (``GKAN_Nodes`` / ``GFASTKAN_Nodes`` on column
shards: shard-local BatchNorm1d, the skip read-out as an input-sharded layer closed by ONE all-reduce of the ``[N, classes]`` partial
sums); config 4's replicas with one flat gradient all-reduce live in ``kagnn_amd.harness`` (``GradientReplicas``).

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


class _Comm:
    """Per-layer bookkeeping of the chunked, overlapped collectives.

    Row chunks make the exchange of chunk i run beside the KAN kernels of chunk i+1 ("chunk along N and
    overlap with the GEMM of the next node tile").  c10d makes a collective's stream wait for everything enqueued so far
    on the CURRENT stream, so every collective is launched under a side stream that waits only for the event of the
    tensor it sends; the compute stream waits for a chunk's result only where it is consumed."""

    def __init__(self, group):
        self.group = group
        self.side = None
        self.pending = []          # forward: (work, keep-alive tensors) of the reduce-scatters in flight
        self.gathered = {}         # backward: chunk index -> (work, buffer) of the all-gathers launched ahead

    def launch(self, fn, ready_after: torch.Tensor):
        """run ``fn()`` (which issues ONE async collective and returns its work handle) so that it depends only on
        ``ready_after`` having been produced"""
        if not ready_after.is_cuda:
            return fn()
        if self.side is None:
            self.side = torch.cuda.Stream(device=ready_after.device)
        ev = torch.cuda.Event()
        ev.record()                                   # `ready_after` was just produced on the current stream
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            return fn()


class _ReduceScatterChunk(Function):
    """rows [r0, r1) of one KANLinear's partial sums: y_shard = (sum over ranks)[:, my columns], launched asynchronously;
    the backward picks up the all-gather of this chunk's gradient rows that ``_Finish.backward`` launched ahead."""

    @staticmethod
    def forward(ctx, partial, comm, chunk):
        group = comm.group
        world = dist.get_world_size(group)
        n, out = partial.shape
        w = out // world
        # rank-major blocks [P][n][out/P] so that block p is what rank p keeps
        blocks = partial.detach().view(n, world, w).permute(1, 0, 2).contiguous().view(world * n, w)
        y = torch.empty((n, w), dtype=partial.dtype, device=partial.device)
        work = comm.launch(lambda: dist.reduce_scatter_tensor(y, blocks, op=dist.ReduceOp.SUM, group=group, async_op=True), blocks)
        comm.pending.append((work, blocks, y))
        ctx.comm, ctx.chunk, ctx.world = comm, chunk, world
        return y

    @staticmethod
    def backward(ctx, g_shard):
        comm, world = ctx.comm, ctx.world
        hit = comm.gathered.pop(ctx.chunk, None)
        if hit is None:                               # (not launched ahead: e.g. the output was consumed chunk-wise)
            g = g_shard.contiguous()
            buf = torch.empty((world * g.size(0), g.size(1)), dtype=g.dtype, device=g.device)
            work = comm.launch(lambda: dist.all_gather_into_tensor(buf, g, group=comm.group, async_op=True), g)
            hit = (work, buf, g)
        work, buf, _ = hit
        work.wait()                                   # CUDA: the compute stream waits for THIS chunk only
        n = buf.size(0) // world
        return buf.view(world, n, -1).permute(1, 0, 2).reshape(n, -1), None, None


class _Finish(Function):
    """forward: wait for every reduce-scatter of the layer (CUDA: the compute stream waits), THEN concatenate the row
    chunks; backward (runs FIRST on the way back): the all-gathers of ALL row chunks of the incoming gradient are
    launched at once on the side stream -- chunk i's gather then runs beside the backward kernels of chunk i+1."""

    @staticmethod
    def forward(ctx, comm, bounds, *parts):
        for work, _, _ in comm.pending:
            work.wait()
        comm.pending.clear()
        ctx.comm, ctx.bounds = comm, bounds
        return parts[0].view_as(parts[0]) if len(parts) == 1 else torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, g):
        comm = ctx.comm
        world = dist.get_world_size(comm.group)
        g = g.contiguous()
        pieces = []
        for i, (r0, r1) in enumerate(ctx.bounds):
            gc = g[r0:r1]
            buf = torch.empty((world * (r1 - r0), g.size(1)), dtype=g.dtype, device=g.device)
            work = comm.launch(lambda gc=gc, buf=buf: dist.all_gather_into_tensor(buf, gc, group=comm.group, async_op=True), g)
            comm.gathered[i] = (work, buf, gc)
            pieces.append(gc)
        return (None, None, *pieces)


class _P2PStep:
    """one use of a KANLinear's peer-to-peer exchange (one forward call and the backward that belongs to it): which of the two
    partial-sum buffers it writes, the output / gathered-gradient tensors the pull kernels fill, the row chunks and the
    events the compute stream waits on"""

    def __init__(self, xch, n, out, bounds, device):
        self.xch, self.n, self.out, self.bounds = xch, n, out, bounds
        self.w = out // xch.world
        self.part = xch.part[xch.fwd_uses & 1]
        xch.fwd_uses += 1
        self.y = torch.empty((n, self.w), dtype=torch.float32, device=device)
        self.g_full = None
        self.g_ready = []


class _P2PPullChunk(Function):
    """rows [r0, r1) of one KANLinear's exchange in its direct peer-to-peer form (``comm="p2p"``).  The KAN forward has just
    written this rank's partial sums of those rows into its peer-mapped buffer; on the layer's SIDE stream: wait for that
    kernel only, rank barrier (every rank's rows are complete), ONE kernel that reads every peer's column block of the rows
    and sums in rank order (``kagnn_p2p_reduce_scatter``: no rank-major staging copy, no ring).  The compute stream goes on
    with the KAN kernel of the next chunk -- the pull of chunk i runs beside the compute of chunk i+1.
    Backward: hands out the rows of the gathered gradient that ``_P2PJoin.backward`` requested ahead, after waiting for THIS
    chunk's pull only."""

    @staticmethod
    def forward(ctx, partial, step, chunk):
        from . import p2p
        xch = step.xch
        r0, r1 = step.bounds[chunk]
        ev = torch.cuda.Event()
        ev.record()                                   # the KAN kernel that produced these rows
        with torch.cuda.stream(xch.side):
            xch.side.wait_event(ev)
            p2p.rank_barrier(xch.group)
            p2p.reduce_scatter(step.part, step.n, step.out, rows=(r0, r1), y=step.y)
        ctx.step, ctx.chunk = step, chunk
        return step.y[r0:r1]

    @staticmethod
    def backward(ctx, _g_piece):
        step = ctx.step
        r0, r1 = step.bounds[ctx.chunk]
        torch.cuda.current_stream().wait_event(step.g_ready[ctx.chunk])
        return step.g_full[r0:r1], None, None


class _P2PJoin(Function):
    """forward: the compute stream waits for the pulls of all row chunks and hands out the assembled ``[n, out/P]`` shard.
    backward (runs FIRST on the way back): this rank's gradient shard goes into one of its two peer-mapped gradient buffers,
    then on the side stream one rank barrier and, chunk by chunk, ONE kernel that pulls all ranks' shards of the rows into
    the final ``[rows, out]`` layout (``kagnn_p2p_all_gather``: no un-permute) -- chunk i+1 is pulled while the dX / dW
    kernels of chunk i run."""

    @staticmethod
    def forward(ctx, step, *parts):
        done = torch.cuda.Event()
        done.record(step.xch.side)
        torch.cuda.current_stream().wait_event(done)
        ctx.step = step
        return step.y.view_as(step.y)

    @staticmethod
    def backward(ctx, g):
        from . import p2p
        step = ctx.step
        xch, n, w = step.xch, step.n, step.w
        buf = xch.grad[xch.bwd_uses & 1]
        xch.bwd_uses += 1
        buf.local[: n * w].view(n, w).copy_(g)
        step.g_full = torch.empty((n, step.out), dtype=torch.float32, device=g.device)
        ev = torch.cuda.Event()
        ev.record()
        step.g_ready = []
        with torch.cuda.stream(xch.side):
            xch.side.wait_event(ev)
            p2p.rank_barrier(xch.group)
            for r0, r1 in step.bounds:
                p2p.all_gather(buf, n, w, rows=(r0, r1), g=step.g_full)
                e = torch.cuda.Event()
                e.record()
                step.g_ready.append(e)
        return (None, *[g[r0:r1] for r0, r1 in step.bounds])


class _P2PExchange:
    """the peer-mapped buffers of one KANLinear of the feature-sharded layer: partial sums [n, out] and gradient shards
    [n, out/P], TWO of each, used alternately (allocated, and their IPC handles exchanged, once per (layer, row count)).

    Why two: an exchange is  write own buffer -> rank barrier -> pull from the peers.  Nothing after the pull tells
    a rank that its PEERS have finished reading its buffer, so writing the same buffer again is only safe after a later
    barrier -- which a chain with ONE exchange per step (``nb_layers=1`` under ``no_grad``) does not have.  With alternating
    buffers the next write of buffer b comes two uses later, and the use in between contains a barrier that every rank
    enters only after its own pulls of the earlier use (same side stream, stream order) -- and this rank's compute stream
    joins that side stream before the later write."""

    def __init__(self, n: int, out: int, device, group, side):
        from . import p2p
        self.group, self.n, self.out, self.side = group, n, out, side
        self.world = dist.get_world_size(group)
        self.part = [p2p.PeerBuffers(n * out, device, group) for _ in range(2)]
        self.grad = [p2p.PeerBuffers(n * (out // self.world), device, group) for _ in range(2)]
        self.fwd_uses = 0
        self.bwd_uses = 0


def _chunk_bounds(n: int, chunks: int):
    if n <= 0:
        return [(0, 0)]                                          # an empty shard still runs its (empty) collectives
    chunks = max(1, min(chunks, n))
    per = -(-n // chunks)
    per = -(-per // 256) * 256 if n >= 4096 else per            # whole 256-row kernel tiles
    return [(r, min(n, r + per)) for r in range(0, n, per)]


class ShardedKANLinear(nn.Module):
    """Input-feature slice of a KANLinear: ``forward`` returns this rank's partial sums over its features for ALL outputs.
    ``columns`` (a LongTensor; default: the contiguous block ``[rank*w, (rank+1)*w)``) names the input features this rank
    holds -- the skip read-out of a sharded node model owns one block of every part of ``[x | h1 | ... | hL]``.
    ``scatter_out=False``: the partial sums are closed by an all-reduce instead of a reduce-scatter, so ``out_features`` need
    not be divisible by the world size (40 classes over 8 ranks)."""

    def __init__(self, full: KANLinear, rank: int, world: int, columns: Optional[torch.Tensor] = None, scatter_out: bool = True):
        super().__init__()
        if columns is None and full.in_features % world:
            raise ValueError("in_features and out_features must be divisible by the world size")
        if scatter_out and full.out_features % world:
            raise ValueError("in_features and out_features must be divisible by the world size")
        self.grid_size, self.spline_order = full.grid_size, full.spline_order
        self.out_features = full.out_features
        if columns is None:
            w = full.in_features // world
            self.lo, self.hi = rank * w, (rank + 1) * w
            columns = torch.arange(self.lo, self.hi)
        columns = columns.to(full.base_weight.device)
        self.register_buffer("columns", columns.clone(), persistent=False)
        if scatter_out:
            ow = full.out_features // world
            self.out_lo, self.out_hi = rank * ow, (rank + 1) * ow
        self.base_weight = nn.Parameter(full.base_weight.detach()[:, columns].clone())
        self.spline_weight = nn.Parameter(full.spline_weight.detach()[:, columns].clone())
        self.spline_scaler = (nn.Parameter(full.spline_scaler.detach()[:, columns].clone())
                              if full.enable_standalone_scale_spline else None)
        self.register_buffer("knots", full.grid[0].detach().clone())
        self.precision = full.precision

    def forward(self, x_shard: torch.Tensor, local_ops, packed=None, out=None) -> torch.Tensor:
        kw = {}
        if packed is not None:
            kw["packed"] = packed                     # this step's weight packs, made once for all row chunks
        if out is not None:
            kw["out"] = out                           # the partial sums written straight into a (peer-mapped) buffer
        return local_ops.kan_linear(x_shard, self.base_weight, self.spline_weight, self.spline_scaler,
                                    self.knots, self.grid_size, self.spline_order, self.precision, **kw)


class ShardedGIKANLayer(nn.Module):
    """``GIKANLayer`` (sum-aggregate + KAN chain) on 1/P of the feature columns per rank.

    ``chunks``: row chunks per KANLinear whose reduce-scatter (forward) / all-gather (backward) overlap with the KAN
    kernels of the neighbouring chunk; ``None`` = 4 from 256k rows up, else 1."""

    def __init__(self, conv: GIKANLayer, group=None, local_ops=None, chunks: Optional[int] = None, comm: str = "rccl"):
        """``comm="rccl"`` (default): reduce-scatter / all-gather through ``torch.distributed`` (RCCL), row-chunked and
        overlapped.  ``comm="p2p"``: the direct form of SURVEY.md 8(e) -- every rank maps its peers' exchange buffers
        (hipIpc, ``kagnn_amd/p2p.py``) and ONE kernel per exchange pulls the column blocks over xGMI
        (``kagnn_p2p_reduce_scatter`` / ``kagnn_p2p_all_gather``); the KAN forward writes its partial sums straight into the
        peer-mapped buffer and the gathered gradient lands in its final layout: no staging passes, no ring.
        ``comm="rccl_c"``: the RCCL form with the whole exchange inside the library -- one call per KANLinear each way
        (``kagnn_sharded_kan_linear_fwd / _bwd`` of include/kagnn_rccl.h on an ``ncclComm_t`` of this layer's own): the same
        row chunks and overlap, no Python between the chunks, ONE weight-gradient pass over all rows."""
        super().__init__()
        if comm not in ("rccl", "rccl_c", "p2p"):
            raise ValueError("comm must be 'rccl', 'rccl_c' or 'p2p'")
        self.comm = comm
        self._ccomm = None                 # comm="rccl_c": this layer's own ncclComm_t (kagnn_amd/rccl.py), made on first use
        self._xch = {}
        self._side = None                  # ONE side stream for all exchanges of this layer (their order is part of the buffer-reuse argument)
        self.group = group
        self.chunks = chunks
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.local_ops = _hip_ops if local_ops is None else local_ops
        self.eps = float(conv.eps)
        if comm == "p2p":
            # checked here, not in the middle of a forward: what the pull kernels and kan_linear(out=) need
            if self.world > 16:
                raise ValueError("comm='p2p' supports at most 16 ranks (kagnn_p2p_*: peer pointer table)")
            if self.local_ops is not _hip_ops:
                raise ValueError("comm='p2p' needs the HIP ops (kan_linear(out=) writes into the peer-mapped buffer)")
            for l in conv.nn.layers:
                if l.out_features % self.world or (l.out_features // self.world) % 4:
                    raise ValueError(f"comm='p2p': out_features / world = {l.out_features}/{self.world} must be a multiple of 4 "
                                     "(16-byte pulls); use comm='rccl' for this layer")
                if l.grid_size + l.spline_order > 16:
                    raise ValueError("comm='p2p': layers with more than 16 coefficients run as summed groups and cannot write "
                                     "into a caller's buffer; use comm='rccl'")
        self.layers = nn.ModuleList(ShardedKANLinear(l, self.rank, self.world) for l in conv.nn.layers)

    def shard_columns(self, t: torch.Tensor) -> torch.Tensor:
        w = t.size(1) // self.world
        return t[:, self.rank * w:(self.rank + 1) * w].contiguous()

    def _exchange(self, li: int, n: int, out: int, device) -> "_P2PExchange":
        """the peer-mapped buffers of layer ``li`` at row count ``n`` (collective: every rank reaches this at the same layer and
        row count).  One row count is kept per layer: a new one replaces the old buffers instead of piling up."""
        hit = self._xch.get(li)
        if hit is None or hit.n != n:
            if self._side is None:
                self._side = torch.cuda.Stream(device=device)
            hit = self._xch[li] = _P2PExchange(n, out, device, self.group, self._side)
        return hit

    def forward(self, x_shard: torch.Tensor, graph) -> torch.Tensor:
        h = self.local_ops.aggregate_sum(x_shard, graph, self_scale=1.0 + self.eps)
        n = h.size(0)
        bounds = _chunk_bounds(n, self.chunks if self.chunks is not None else (4 if n >= 262144 else 1))
        # the weight packs of the whole chain in ONE launch, shared by all row chunks (otherwise every chunk packs again)
        packs = None
        pack_chain = getattr(self.local_ops, "kan_pack_chain", None)
        if pack_chain is not None and h.is_cuda:
            first = self.layers[0]
            mode = first.precision if first.precision is not None else self.local_ops.default_precision()
            if all(l.spline_scaler is not None and l.grid_size == first.grid_size and l.spline_order == first.spline_order
                   and l.precision == first.precision for l in self.layers):
                packs = pack_chain([(l.base_weight, l.spline_weight, l.spline_scaler) for l in self.layers],
                                   first.grid_size, first.spline_order, mode)
        if self.comm == "rccl_c":
            from . import rccl
            if self.local_ops is not _hip_ops:
                raise ValueError("comm='rccl_c' needs the HIP ops (the exchange is a library call)")
            if self._ccomm is None:
                self._ccomm = rccl.Communicator.from_group(self.group, h.device)
            for li, layer in enumerate(self.layers):
                h = rccl.sharded_kan_linear(h, layer.base_weight, layer.spline_weight, layer.spline_scaler, layer.knots,
                                            layer.grid_size, layer.spline_order, layer.precision, self._ccomm,
                                            row_chunks=len(bounds), packed=None if packs is None else packs[li])
            return h
        if self.comm == "p2p":
            for li, layer in enumerate(self.layers):
                out = layer.out_features
                step = _P2PStep(self._exchange(li, n, out, h.device), n, out, bounds, h.device)
                rows = step.part.local[: n * out].view(n, out)
                parts = []
                for i, (r0, r1) in enumerate(bounds):
                    # the KAN forward writes its partial sums STRAIGHT into the peer-mapped buffer
                    partial = layer(h[r0:r1], self.local_ops, None if packs is None else packs[li], out=rows[r0:r1])
                    parts.append(_P2PPullChunk.apply(partial, step, i))
                h = _P2PJoin.apply(step, *parts)
            return h
        for li, layer in enumerate(self.layers):
            comm = _Comm(self.group)
            parts = [_ReduceScatterChunk.apply(layer(h[r0:r1], self.local_ops, None if packs is None else packs[li]), comm, i)
                     for i, (r0, r1) in enumerate(bounds)]
            h = _Finish.apply(comm, bounds, *parts)
        return h


# ----------------------------------------------------------------------------------------------------------
# feature-sharded FastKAN ("LayerNorm statistics -> all-reduce of 2 floats/row")
class _ShardedFastKANFn(Function):
    """rows [r0, r1) of one feature-sharded FastKANLayer (reference ``fastkan.py:76-85`` on a column shard): forward = this rank's
    partial sums over its input columns, LayerNorm on the row statistics of ALL ranks' columns (``stats``: merged by the layer
    before the row chunks start).  Backward, from the gathered gradient rows: the input-gradient half leaves d loss / dz and the
    two row sums of the LayerNorm backward over the local columns; their all-reduce (2 floats per row) runs on the layer's side
    stream BESIDE the weight-gradient half; the finish completes gx and the LayerNorm parameter gradients."""

    @staticmethod
    def forward(ctx, x, stats, ln_w, ln_b, spline_w, base_w, base_b, layer):
        ops = layer.local_ops
        y = ops.fastkan_shard_fwd(x.detach(), stats, ln_w, ln_b, spline_w, base_w, base_b, layer.centers, layer.denominator, layer.precision)
        ctx.save_for_backward(x, stats, ln_w, ln_b, spline_w, base_w)
        ctx.layer, ctx.has_bias = layer, base_b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, ln_w, ln_b, spline_w, base_w = ctx.saved_tensors
        layer = ctx.layer
        ops, group = layer.local_ops, layer.group
        gy = gy.contiguous()
        halves = getattr(ops, "fastkan_shard_bwd_halves", None)
        if halves is not None and ln_w is not None and x.is_cuda:
            # input-gradient half -> all-reduce of the row sums on the side stream || weight-gradient half -> finish
            st, sums = halves(x, gy, stats, ln_w, ln_b, spline_w, base_w, layer.centers, layer.denominator, layer.precision, part="input")
            work = layer.comm.launch(lambda: dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group, async_op=True), sums)
            gsw, gbw, gbb = halves(x, gy, stats, ln_w, ln_b, spline_w, base_w, layer.centers, layer.denominator, layer.precision,
                                   part="weight", state=st, want_bias=ctx.has_bias)
            work.wait()
        else:
            st, sums, gsw, gbw, gbb = ops.fastkan_shard_bwd(x, gy, stats, ln_w, ln_b, spline_w, base_w, layer.centers, layer.denominator,
                                                            layer.precision, want_bias=ctx.has_bias)
            if sums is not None:
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        gx, glw, glb = ops.fastkan_shard_bwd_finish(st, sums, layer.in_total)
        return gx, None, glw, glb, gsw, gbw, gbb, None


class ShardedFastKANLayer(nn.Module):
    """Input-column slice of a ``FastKANLayer`` (reference ``fastkan.py:49-85``): ``layernorm.weight / .bias`` and
    ``base_linear.weight`` on this rank's columns, ``spline_linear.weight`` on their ``num_grids`` RBF columns each;
    ``base_linear.bias`` lives on ``bias_rank`` only (it enters the summed partial sums once).  ``forward`` returns the partial
    sums ``[N, out]`` -- one tensor per row chunk of ``bounds`` -- for the caller to reduce-scatter (a convolution's chain) or
    all-reduce (a read-out).  ``columns`` as in ``ShardedKANLinear``."""

    def __init__(self, full, rank: int, world: int, group=None, local_ops=None, columns: Optional[torch.Tensor] = None,
                 bias_rank: int = 0):
        super().__init__()
        if columns is None:
            if full.input_dim % world:
                raise ValueError("input_dim must be divisible by the world size")
            w = full.input_dim // world
            self.lo, self.hi = rank * w, (rank + 1) * w
            columns = torch.arange(self.lo, self.hi)
        self.group, self.world, self.rank = group, world, rank
        self.local_ops = _hip_ops if local_ops is None else local_ops
        self.in_total, self.out_features = full.input_dim, full.output_dim
        ng = full.rbf.num_grids
        dev = full.spline_linear.weight.device
        columns = columns.to(dev)
        self.register_buffer("columns", columns.clone(), persistent=False)
        if full.layernorm is not None:
            self.ln_weight = nn.Parameter(full.layernorm.weight.detach()[columns].clone())
            self.ln_bias = nn.Parameter(full.layernorm.bias.detach()[columns].clone())
            self.ln_eps = float(full.layernorm.eps)
        else:
            self.ln_weight = self.ln_bias = None
            self.ln_eps = 1e-5
        sw = full.spline_linear.weight.detach().view(full.output_dim, full.input_dim, ng)
        self.spline_weight = nn.Parameter(sw[:, columns, :].reshape(full.output_dim, columns.numel() * ng).clone())
        if full.use_base_update:
            self.base_weight = nn.Parameter(full.base_linear.weight.detach()[:, columns].clone())
            self.base_bias = nn.Parameter(full.base_linear.bias.detach().clone()) if rank == bias_rank else None
        else:
            self.base_weight = self.base_bias = None
        self.register_buffer("centers", full.rbf.grid.detach().clone())
        self.denominator = float(full.rbf.denominator)
        self.precision = full.precision
        self.comm = _Comm(group)               # the side stream of the backward's row-sum all-reduce

    def row_stats(self, x_shard: torch.Tensor) -> Optional[torch.Tensor]:
        """LayerNorm statistics of every row over ALL ranks' columns: local (mean, M2) -> all-gather (2 floats per row and
        rank) -> merged in rank order on every rank (the same bits everywhere).  Not differentiated: the LayerNorm backward's
        two row sums carry the dependence of the statistics on x."""
        if self.ln_weight is None:
            return None
        ops = self.local_ops
        with torch.no_grad():
            mom = ops.fastkan_row_moments(x_shard.detach())
            gathered = torch.empty((self.world,) + tuple(mom.shape), dtype=mom.dtype, device=mom.device)
            dist.all_gather_into_tensor(gathered.view(-1, 2), mom, group=self.group)
            return ops.fastkan_merge_moments(gathered, x_shard.size(1), self.ln_eps)

    def forward(self, x_shard: torch.Tensor, bounds=None):
        stats = self.row_stats(x_shard)
        if bounds is None:
            bounds = [(0, x_shard.size(0))]
        return [_ShardedFastKANFn.apply(x_shard[r0:r1], None if stats is None else stats[r0:r1], self.ln_weight, self.ln_bias,
                                        self.spline_weight, self.base_weight, self.base_bias, self) for r0, r1 in bounds]


class ShardedGIFASTKANLayer(nn.Module):
    """``GIFASTKANLayer`` (reference ``models.py:85-92``: sum-aggregate + FastKAN chain) on 1/P of the feature columns per rank:
    column-sharded aggregation, then per FastKANLayer the LayerNorm exchange (2 floats per row each way), the local RBF
    expansion + contraction over this rank's input columns, and the SAME reduce-scatter (forward) / all-gather (backward) of
    the partial sums as ``ShardedGIKANLayer`` -- row-chunked and overlapped the same way."""

    def __init__(self, conv, group=None, local_ops=None, chunks: Optional[int] = None):
        super().__init__()
        self.group, self.chunks = group, chunks
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.local_ops = _hip_ops if local_ops is None else local_ops
        self.eps = float(conv.eps)
        for l in conv.nn.layers:
            if l.input_dim % self.world or l.output_dim % self.world:
                raise ValueError("layer widths must be divisible by the world size")
        self.layers = nn.ModuleList(ShardedFastKANLayer(l, self.rank, self.world, group, self.local_ops) for l in conv.nn.layers)

    def shard_columns(self, t: torch.Tensor) -> torch.Tensor:
        w = t.size(1) // self.world
        return t[:, self.rank * w:(self.rank + 1) * w].contiguous()

    def forward(self, x_shard: torch.Tensor, graph) -> torch.Tensor:
        h = self.local_ops.aggregate_sum(x_shard, graph, self_scale=1.0 + self.eps)
        n = h.size(0)
        bounds = _chunk_bounds(n, self.chunks if self.chunks is not None else (4 if n >= 262144 else 1))
        for layer in self.layers:
            comm = _Comm(self.group)
            parts = [_ReduceScatterChunk.apply(p, comm, i) for i, p in enumerate(layer(h, bounds))]
            h = _Finish.apply(comm, bounds, *parts)
        return h


# ----------------------------------------------------------------------------------------------------------
# a whole node model on column shards
class _AllReduceSum(Function):
    """partial sums -> their sum on every rank.  Backward: identity -- every rank evaluates the same loss on the same summed
    tensor, so the gradient it holds already IS the gradient of its partial sums (no communication)."""

    @staticmethod
    def forward(ctx, partial, group):
        y = partial.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


class ShardedNodeModel(nn.Module):
    """``GKAN_Nodes`` / ``GFASTKAN_Nodes`` with GIN convolutions (reference ``models.py:150-257``) on column shards:
    ``mp_layers x {sharded conv -> BatchNorm1d on the LOCAL columns (per-feature statistics: no communication) -> dropout}``,
    the skip read-out as an INPUT-sharded KANLinear / FastKANLayer over this rank's columns of ``[x | h1 | ... | hL]``
    (``models.py:200-203``) closed by ONE all-reduce of the ``[N, classes]`` partial sums (classes need not divide by P), so every
    rank holds the full logits and evaluates the loss itself.  Parameters are sharded (gradients are local); only the FastKAN
    base biases live on rank 0.  ``forward(x_shard, graph)``: ``x_shard = shard_columns(x)``."""

    def __init__(self, model, group=None, local_ops=None, chunks: Optional[int] = None, comm: str = "rccl"):
        super().__init__()
        from .models import GIFASTKANLayer
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.local_ops = _hip_ops if local_ops is None else local_ops
        self.skip = bool(model.skip)
        convs = []
        for c in model.convs:
            if isinstance(c, GIFASTKANLayer):
                convs.append(ShardedGIFASTKANLayer(c, group, local_ops, chunks))
            elif isinstance(c, GIKANLayer):
                convs.append(ShardedGIKANLayer(c, group, local_ops, chunks, comm))
            else:
                raise ValueError("ShardedNodeModel shards the GIN convolutions (GIKANLayer / GIFASTKANLayer); "
                                 f"got {type(c).__name__}")
        self.convs = nn.ModuleList(convs)
        bn_cls = getattr(self.local_ops, "BatchNorm1d", None)
        if bn_cls is None:
            from .norm import BatchNorm1d as bn_cls
        widths = []
        self.bns = nn.ModuleList()
        for bn in model.bns:
            if bn.num_features % self.world:
                raise ValueError("hidden_channels must be divisible by the world size")
            w = bn.num_features // self.world
            sl = slice(self.rank * w, (self.rank + 1) * w)
            b = bn_cls(w, eps=bn.eps, momentum=bn.momentum, affine=bn.affine, track_running_stats=bn.track_running_stats)
            with torch.no_grad():
                if bn.affine:
                    b.weight.copy_(bn.weight[sl]); b.bias.copy_(bn.bias[sl])
                if bn.track_running_stats:
                    b.running_mean.copy_(bn.running_mean[sl]); b.running_var.copy_(bn.running_var[sl])
                    b.num_batches_tracked.copy_(bn.num_batches_tracked)
            self.bns.append(b.to(bn.weight.device if bn.affine else bn.running_mean.device))
            widths.append(bn.num_features)
        self.dropout = nn.Dropout(model.dropout.p)
        # this rank's input columns of the read-out: its block of every concatenated part
        first = model.convs[0].nn.layers[0]
        fin = first.in_features if hasattr(first, "in_features") else first.input_dim
        if fin % self.world:
            raise ValueError("num_features must be divisible by the world size")
        parts = ([fin] + widths) if self.skip else [widths[-1]]
        cols, off = [], 0
        for wd in parts:
            w = wd // self.world
            cols.append(torch.arange(off + self.rank * w, off + (self.rank + 1) * w))
            off += wd
        cols = torch.cat(cols)
        self.num_features = fin
        if isinstance(model.lay_out, KANLinear):
            self.lay_out = ShardedKANLinear(model.lay_out, self.rank, self.world, columns=cols, scatter_out=False)
        else:
            self.lay_out = ShardedFastKANLayer(model.lay_out, self.rank, self.world, group, self.local_ops, columns=cols)

    def shard_columns(self, t: torch.Tensor) -> torch.Tensor:
        w = t.size(1) // self.world
        return t[:, self.rank * w:(self.rank + 1) * w].contiguous()

    def forward(self, x_shard: torch.Tensor, graph) -> torch.Tensor:
        if self.local_ops is _hip_ops and not isinstance(graph, _hip_ops.GraphIndex):
            graph = _hip_ops.graph_index(graph, x_shard.size(0))
        cat = getattr(self.local_ops, "concat_columns", None) or (lambda parts: torch.cat(parts, dim=1))
        outs = [x_shard]
        x = x_shard
        for conv, bn in zip(self.convs, self.bns):
            x = self.dropout(bn(conv(x, graph)))
            outs.append(x)
        h = cat(outs) if self.skip else x
        if isinstance(self.lay_out, ShardedKANLinear):
            partial = self.lay_out(h, self.local_ops)
        else:
            partial = self.lay_out(h)[0]
        return _AllReduceSum.apply(partial, self.group)


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


class _P2PTranspose(Function):
    """the two all-to-alls of the transposed layer as direct peer-to-peer pulls (``comm="p2p"``): ``to_rows=True`` turns this
    rank's column shard ``[n, w]`` into its row shard ``[n_me, P*w]``, ``False`` the other way; the backward is the opposite
    direction through the partner buffer.  The operand is copied into this rank's peer-mapped buffer (1/P of an activation),
    a rank barrier follows, one kernel (or one per peer) pulls over xGMI into the final layout -- no send-side packing pass."""

    @staticmethod
    def forward(ctx, x, xch, to_rows, n):
        ctx.xch, ctx.to_rows, ctx.n = xch, to_rows, n
        return xch.run(x, to_rows, n, fwd=True)

    @staticmethod
    def backward(ctx, g):
        return ctx.xch.run(g.contiguous(), not ctx.to_rows, ctx.n, fwd=False), None, None, None


class _P2PTransposeExchange:
    """peer-mapped buffers of the transposed layer: [n, w] column shards and [n_max, P*w] row blocks, one pair per direction
    of the tape (forward / backward), allocated once per (row count, width)"""

    def __init__(self, n: int, w: int, device, group):
        from . import p2p
        world = dist.get_world_size(group)
        self.group, self.w = group, w
        self.splits = _row_splits(n, world)
        self.cols = [p2p.PeerBuffers(n * w, device, group) for _ in range(2)]                     # fwd, bwd
        self.rows = [p2p.PeerBuffers(max(self.splits) * w * world, device, group) for _ in range(2)]

    def run(self, x, to_rows, n, fwd):
        # (buffer reuse: every buffer here is written once per step and the layer runs TWO exchanges per direction -- in and
        # out, each with its rank barrier -- so a rank's next write of a buffer always follows a later barrier that its peers
        # entered after their pulls of the previous use; no double buffering needed, unlike _P2PExchange)
        from . import p2p
        k = 0 if fwd else 1
        if to_rows:
            buf = self.cols[k]
            buf.local[: x.numel()].view_as(x).copy_(x)
            p2p.rank_barrier(self.group)
            return p2p.cols_to_rows(buf, n, self.w, self.splits)
        buf = self.rows[k]
        buf.local[: x.numel()].view_as(x).copy_(x)
        p2p.rank_barrier(self.group)
        return p2p.rows_to_cols(buf, n, self.w, self.splits)


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


class _QueueFlatSync(Function):
    """identity; its backward runs first on the way back and queues ``module.sync_gradients`` to run when the whole
    backward pass has finished (``Variable._execution_engine.queue_callback``) -- one flat all-reduce per step"""

    @staticmethod
    def forward(ctx, y, module):
        ctx.module = module
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        ctx.module._arm_flat_sync()
        return g, None


class TransposedShardedGIKANLayer(nn.Module):
    """``GIKANLayer`` with the aggregation on column shards and the KAN chain on row shards (see the module
    docstring).  Same interface as ``ShardedGIKANLayer``: column shard in, column shard out."""

    def __init__(self, conv: GIKANLayer, group=None, local_ops=None, sync_in_backward="flat", comm: str = "rccl"):
        """``sync_in_backward="flat"`` (default): ONE flat all-reduce over all parameter gradients, queued as an
        end-of-backward callback of the autograd engine -- transparent to the caller.  ``True``: every parameter's
        gradient is all-reduced inside autograd (one small collective per parameter tensor).  ``False``: gradients stay
        rank-local until the caller runs ``sync_gradients()`` (what ``bench.py`` times explicitly)."""
        super().__init__()
        if comm not in ("rccl", "p2p"):
            raise ValueError("comm must be 'rccl' or 'p2p'")
        self.comm = comm                  # "p2p": the two all-to-alls as direct pulls over hipIpc-mapped peer buffers (_P2PTranspose)
        self._xch = {}
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
        self._flat_pending = None          # (engine run id, gradients as they stood when that backward pass reached the module)

    def _arm_flat_sync(self) -> None:
        """called by the first backward node of this module an engine run reaches (before any of the run's parameter
        gradients has been accumulated): remember the gradients accumulated by EARLIER backward passes and queue ONE
        end-of-backward callback.  The callback all-reduces only what this pass added -- with gradient accumulation
        (two backward() calls without zero_grad) or the module used twice in one forward, already-synced sums must not
        be summed over the ranks again (P*S1 + S2 instead of S1 + S2)."""
        run = _hip_ops.graph_task_id()
        if self._flat_pending is not None and self._flat_pending[0] == run:
            return                         # a later use of the module in the same pass: already armed
        # (a latch left by ANOTHER engine run is stale: that backward raised, and the engine dropped its queued callbacks --
        # keyed on a plain flag, every later pass returned above and the gradients silently stopped being
        # summed over the ranks)
        from torch.autograd import Variable
        self._flat_pending = (run, [None if p.grad is None else p.grad.detach().clone() for p in self.parameters()])
        Variable._execution_engine.queue_callback(lambda run=run: self._flat_sync_delta(run))

    def _flat_sync_delta(self, run: int) -> None:
        if self._flat_pending is None or self._flat_pending[0] != run:
            return
        prev, self._flat_pending = self._flat_pending[1], None
        params = [p for p in self.parameters()]
        live = [(p, b) for p, b in zip(params, prev) if p.grad is not None]
        if not live:
            return
        flat = torch.cat([(p.grad if b is None else p.grad - b).reshape(-1) for p, b in live])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p, b in live:
            d = flat[off:off + p.grad.numel()].view_as(p.grad)
            p.grad.copy_(d if b is None else b + d)
            off += p.grad.numel()

    def sync_gradients(self) -> None:
        """sum the parameter gradients over the ranks with a single flat all-reduce (for ``sync_in_backward=False``;
        reduces whatever sits in ``.grad`` -- call it once per optimiser step, after the last backward)"""
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
        xin = xout = None
        if self.comm == "p2p":                    # (the layer's input and output widths may differ: one exchange object each)
            for key, width in (("in", h.size(1)), ("out", self.layers[-1].out_features // self.world)):
                if (key, n, width) not in self._xch:
                    self._xch[(key, n, width)] = _P2PTransposeExchange(n, width, h.device, self.group)
            xin, xout = self._xch[("in", n, h.size(1))], self._xch[("out", n, self.layers[-1].out_features // self.world)]
        h = _P2PTranspose.apply(h, xin, True, n) if xin is not None else _ColsToRows.apply(h, self.group)
        for layer in self.layers:
            sc = layer.spline_scaler if layer.enable_standalone_scale_spline else None
            g = self.group
            wrap = (lambda p: _SumGradAcrossRanks.apply(p, g)) if self.sync_in_backward is True else (lambda p: p)
            h = self.local_ops.kan_linear(h, wrap(layer.base_weight), wrap(layer.spline_weight),
                                          None if sc is None else wrap(sc),
                                          layer.grid[0].contiguous(), layer.grid_size, layer.spline_order,
                                          layer.precision)
        out = _P2PTranspose.apply(h, xout, False, n) if xout is not None else _RowsToCols.apply(h, n, self.group)
        if self.sync_in_backward == "flat" and torch.is_grad_enabled():
            out = _QueueFlatSync.apply(out, self)
        return out
