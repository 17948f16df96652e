"""Autograd-aware Python front of the C ABI (include/kagnn_hip.h).

torch is plumbing here: it owns device memory, the current stream and the autograd tape.  Every
arithmetic step of the hot path is a call into libkagnn_hip.so through ctypes.  CPU tensors are
refused -- there is no CPU implementation in this package (the CPU restatement lives in
``oracle/`` and is test infrastructure only).
"""
from __future__ import annotations

import ctypes
import functools
import os
import weakref
from ctypes import byref, c_int64, c_size_t
from typing import Optional, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import PREC_FP32, PREC_FP32_GRID, PREC_HALF, PREC_SPLIT

HUB_THRESHOLD = int(os.environ.get("KAGNN_HUB_THRESHOLD", "96"))       # rows above this many edges are split into segments


def default_precision() -> int:
    """`KAGNN_PRECISION=fp32|split|half` (default split: fp16 hi/lo operands, three products, fp32 accumulate; `half`: the same
    kernels with ONE fp16 product per fp32 product -- the build-defined reduced-precision mode of BASELINE config 2)."""
    v = os.environ.get("KAGNN_PRECISION", "split").lower()
    if v in ("fp32", "exact", "0"):
        return PREC_FP32
    if v in ("split", "1"):
        return PREC_SPLIT
    if v in ("half", "fp16", "3"):
        return PREC_HALF
    raise ValueError(f"KAGNN_PRECISION={v!r}: expected 'fp32', 'split' or 'half'")


def split_like(mode) -> bool:
    """the modes that run on the split-precision kernels (and their packs, fused nodes and shape limits): PREC_SPLIT and its
    single-product variant PREC_HALF"""
    return mode == PREC_SPLIT or mode == PREC_HALF


class EntryPointTimer:
    """Optional per-entry-point device timing (HIP events on the launch stream), used by bench.py
    to report the dominant kernel's duration live.  Off by default: zero overhead."""

    def __init__(self, only: Optional[str] = None):
        self.records = []
        self.only = only                 # time just this entry point (None: all of them)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.records:
            d = out.setdefault(name, [0, 0.0])
            d[0] += 1
            d[1] += a.elapsed_time(b)
        return {k: {"launches": v[0], "total_ms": v[1], "avg_ms": v[1] / v[0]} for k, v in out.items()}


class LibraryStageTimer:
    """Per-stage device timing INSIDE the library (``kagnn_stage_timer_*``): HIP events on the launch stream around every
    per-operation entry point, also when it runs inside the one-call layer entry points -- what ``bench.py`` uses to time the
    dominant kernel live on the product's default path (``KAGNN_LAYER_ABI=1``).  ``only``: record just this stage."""

    def __init__(self, only: Optional[str] = None):
        self.only = only

    def __enter__(self):
        _lib.call("kagnn_stage_timer_enable", None if self.only is None else self.only.encode())
        return self

    def __exit__(self, *exc):
        _lib.call("kagnn_stage_timer_disable")
        return False

    @staticmethod
    def collect(capacity: int = 64):
        """{stage: {"launches", "total_ms", "avg_ms"}} of everything recorded since the last collect (waits for the events)"""
        names = ctypes.create_string_buffer(64 * capacity)
        launches = (c_int64 * capacity)()
        total = (ctypes.c_double * capacity)()
        n = ctypes.c_int32(0)
        _lib.call("kagnn_stage_timer_collect", names, launches, total, capacity, byref(n))
        out = {}
        for i in range(n.value):
            name = names.raw[64 * i:64 * (i + 1)].split(b"\0", 1)[0].decode()
            out[name] = {"launches": int(launches[i]), "total_ms": float(total[i]), "avg_ms": float(total[i]) / max(1, int(launches[i]))}
        return out


_timer: Optional[EntryPointTimer] = None


def set_timer(t: Optional[EntryPointTimer]) -> None:
    global _timer
    _timer = t


def _call(name: str, *args) -> None:
    if _timer is None or (_timer.only is not None and _timer.only != name):
        _lib.call(name, *args)
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.call(name, *args)
    b.record()
    _timer.records.append((name, a, b))


_SIZE_CACHE: dict = {}


def _sizes(name: str, *args, outputs: int = 1):
    """Size queries (``*_bytes`` entry points) are pure functions of their integer arguments: ask the library
    once per distinct shape.  On small graphs the ctypes round trips were a measurable part of a step."""
    key = (name, args)
    hit = _SIZE_CACHE.get(key)
    if hit is None:
        outs = [c_size_t(0) for _ in range(outputs)]
        cargs = [(ctypes.c_int32 * len(a))(*a) if isinstance(a, tuple) else a for a in args]     # (int tuples: int32 arrays)
        _call(name, *cargs, *[byref(o) for o in outs])
        hit = tuple(o.value for o in outs)
        _SIZE_CACHE[key] = hit
    return hit if outputs > 1 else hit[0]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream.  ``torch.cuda.current_stream()`` builds a Stream object through several
    Python layers (~14 us a call, ten calls a layer step: most of the host time of a small graph's step); the raw
    getter is one C call."""
    if _RAW_STREAM is not None:
        return ctypes.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "kagnn_amd ops run only on MI355X device tensors (libkagnn_hip.so); got a CPU tensor. "
                "There is no CPU fallback in this package.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"kagnn_amd ops need all operands on one device; got {dev} and {t.device}")


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_of(t):
    """Context that makes ``t``'s GPU the current device (kernels launch on the CURRENT device's stream: a model moved
    to ``cuda:1`` without ``torch.cuda.set_device(1)`` would otherwise launch on device 0).  Free when it already is."""
    if t is None or not t.is_cuda or t.device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(t.device)


def _on_operand_device(fn):
    """Decorator for ``Function.forward`` / ``.backward``: run on the device of the first CUDA tensor argument."""
    @functools.wraps(fn)
    def run(ctx, *args):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(ctx, *args)
                break
        return fn(ctx, *args)
    return run


def default_activation_dtype() -> torch.dtype:
    """`KAGNN_ACT=fp32|bf16` (default fp32).  bf16: the rows the neighbour aggregation GATHERS (conv inputs on the way
    forward, d loss / d h0 on the way back) are stored as bf16 -- fp32 accumulation, KAN layers unchanged; a
    build-defined mode for BASELINE.json's config 2, outside the 1e-4 fp32 contract (tests: 4e-3)."""
    v = os.environ.get("KAGNN_ACT", "fp32").lower()
    if v in ("fp32", "f32", "float32"):
        return torch.float32
    if v in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError(f"KAGNN_ACT={v!r}: expected 'fp32' or 'bf16'")


def _rows(t: torch.Tensor, allow_bf16: bool = False) -> torch.Tensor:
    """2-D fp32 (or, where the entry point takes them, bf16) rows with unit column stride (row stride is passed to the
    kernels as ld)."""
    if t.dim() != 2:
        raise AssertionError(f"expected a 2-D tensor, got shape {tuple(t.shape)}")
    if t.dtype != torch.float32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise TypeError(f"kagnn_amd kernels are fp32{' / bf16 gather operands' if allow_bf16 else ''}; got {t.dtype}")
    if t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1)):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0))


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ======================================================================== graph structure
class GraphIndex:
    """CSR (by destination) and its transpose (by source) of one ``edge_index``, int32, on device.

    Built once per edge list by ``kagnn_csr_build`` (stable radix sort => ``perm`` equals
    ``argsort(dst, stable=True)`` bit for bit) and cached; replaces the index bookkeeping
    torch_geometric redoes on every ``propagate`` call.
    """

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, hub_threshold: int = HUB_THRESHOLD, defer_validation: bool = False):
        _need_cuda(edge_index)
        if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
        self.num_nodes = int(num_nodes)
        self.num_edges = int(edge_index.size(1))
        self.hub_threshold = int(hub_threshold)
        self.device = edge_index.device
        src = edge_index[0].contiguous()
        dst = edge_index[1].contiguous()
        self._flags = None
        with _device_of(edge_index):
            if _SMALL_CSR and _lib.load().kagnn_csr_small_ok(self.num_edges, self.num_nodes):
                self._build_small(src, dst)
            else:
                self.rowptr, self.col, self.perm, self.hub_seg, self.num_hub_seg = self._build(dst, src)
                self.rowptr_t, self.col_t, self.perm_t, self.hub_seg_t, self.num_hub_seg_t = self._build(src, dst)
        self._dis = None
        if not defer_validation:
            self.validate()             # (small path: waits for its flags -- graphs that are indexed once pay one synchronisation, as
                                        # the rocPRIM path does; the per-batch graphs of the graph-level models defer it)

    def _build_small(self, src, dst):
        """E, N <= 65 536 (the graph-level models' mini-batches, rebuilt per batch): both structures from ONE launch and without
        blocking the stream (``kagnn_csr_build_small``); same arrays, bit for bit; no hub segments.  Out-of-range node ids are
        clamped on the device (nothing reads out of bounds) and the flags travel to pinned host memory behind the launch;
        ``validate()`` reads them: at once (one synchronisation, like the rocPRIM path) unless the caller asked to defer it
        (``graph_index(..., cache=False)``: the per-batch graphs) -- then without waiting when the next small graph is indexed, or
        by whoever calls ``validate()``."""
        n, e, dev = self.num_nodes, self.num_edges, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        _validate_pending(dev)
        self.rowptr, self.col, self.perm = torch.empty(n + 1, **i32), torch.empty(e, **i32), torch.empty(e, **i32)
        self.rowptr_t, self.col_t, self.perm_t = torch.empty(n + 1, **i32), torch.empty(e, **i32), torch.empty(e, **i32)
        self.hub_seg = self.hub_seg_t = torch.empty(0, **i32)
        self.num_hub_seg = self.num_hub_seg_t = 0
        ws = _ws(_sizes("kagnn_csr_small_workspace_bytes", e), dev)
        flags = torch.empty(2, **i32)
        _call("kagnn_csr_build_small", _ptr(src), _ptr(dst), e, n, _ptr(self.rowptr), _ptr(self.col), _ptr(self.perm),
              _ptr(self.rowptr_t), _ptr(self.col_t), _ptr(self.perm_t), _ptr(flags), _ptr(ws), ws.numel(), _stream())
        self._flags = _defer_flag_check(flags)

    def validate(self, wait: bool = True) -> None:
        """raise if the edge list named node ids outside [0, num_nodes) (small-graph path; the rocPRIM path has already raised)"""
        if self._flags is not None:
            _check_flags(self._flags, wait)

    def _build(self, key, val):
        n, e, dev = self.num_nodes, self.num_edges, self.device
        nbytes = c_size_t(0)
        _call("kagnn_csr_workspace_bytes", e, n, byref(nbytes))
        ws = _ws(nbytes.value, dev)
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        col = torch.empty(e, dtype=torch.int32, device=dev)
        perm = torch.empty(e, dtype=torch.int32, device=dev)
        seg_len = max(self.hub_threshold // 4, 32)                  # as in csr.hip
        cap = e // seg_len + e // self.hub_threshold + 1
        hub = torch.empty(3 * cap, dtype=torch.int32, device=dev)
        nseg = c_int64(0)
        _call("kagnn_csr_build", _ptr(key), _ptr(val), e, n, _ptr(rowptr), _ptr(col), _ptr(perm),
                  self.hub_threshold, _ptr(hub), cap, byref(nseg), _ptr(ws), ws.numel(), _stream())
        return rowptr, col, perm, hub, int(nseg.value)

    @property
    def gcn_dis(self) -> torch.Tensor:
        """deg^-1/2 with one self loop per node (gcn_norm, add_remaining_self_loops)."""
        if self._dis is None:
            dis = torch.empty(self.num_nodes, dtype=torch.float32, device=self.device)
            with _device_of(self.rowptr):
                _call("kagnn_gcn_deg_inv_sqrt", _ptr(self.rowptr), _ptr(self.col), self.num_nodes,
                      _ptr(dis), _stream())
            self._dis = dis
        return self._dis

    def side(self, transposed: bool):
        if transposed:
            return self.rowptr_t, self.col_t, self.perm_t, self.hub_seg_t, self.num_hub_seg_t
        return self.rowptr, self.col, self.perm, self.hub_seg, self.num_hub_seg


def _defer_flag_check(flags: torch.Tensor):
    """the out-of-range-id flags of a small-graph CSR build (2 ints on the device) travel to pinned host memory behind the launches
    queued so far; the entry joins ``_pending_checks`` and is looked at -- without waiting -- when a later small graph is indexed, or
    by ``flush_graph_checks()`` / ``GraphIndex.validate()``"""
    dev = flags.device
    host = torch.empty(2, dtype=torch.int32, pin_memory=True)
    host.copy_(flags, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    entry = [host, ev, flags, False]
    _pending_checks.setdefault(dev.index, []).append(entry)
    return entry


_SMALL_CSR = True      # False: always the rocPRIM build (a module attribute for the bitwise tests; identical arrays)
_pending_checks: "dict[int, list]" = {}


def _check_flags(entry, wait: bool) -> bool:
    """``entry`` = [pinned flags, event, device flags, done]: True once the flags have been looked at (each entry reports ONCE)"""
    if entry[3]:
        return True
    host, ev = entry[0], entry[1]
    if not wait and not ev.query():
        return False
    ev.synchronize()
    entry[3] = True
    if int(host[0]) or int(host[1]):
        raise RuntimeError("kagnn_csr_build_small: edge_index holds node ids outside [0, num_nodes)")
    return True


def _validate_pending(dev) -> None:
    """the deferred range checks of earlier small graphs on this device whose flags have arrived (never waits)"""
    lst = _pending_checks.get(dev.index)
    if lst:
        keep, bad = [], False
        for en in lst:
            try:
                if not _check_flags(en, False):
                    keep.append(en)
            except RuntimeError:
                bad = True
        _pending_checks[dev.index] = keep
        if bad:
            raise RuntimeError("kagnn_csr_build_small: an edge_index indexed earlier held node ids outside [0, num_nodes)")


def flush_graph_checks(device=None) -> None:
    """WAIT for the deferred node-id range checks of every small graph indexed so far (``graph_index(cache=False)``:
    ``kagnn_csr_build_small`` clamps out-of-range ids and raises a flag that is otherwise only looked at when a LATER graph is
    indexed) and raise if any failed.  Call it at a natural synchronisation point -- the end of an epoch or an evaluation loop,
    before reading outputs back on the host (the last batch of a loop, or a single inference call, was never
    validated).  ``harness.train_graph_batches`` does so at the end of every epoch."""
    devs = list(_pending_checks) if device is None else [torch.device(device).index]
    bad = False
    for d in devs:
        for en in _pending_checks.get(d) or []:
            try:
                _check_flags(en, True)
            except RuntimeError:
                bad = True
        _pending_checks[d] = []
    if bad:
        raise RuntimeError("kagnn_csr_build_small: an edge_index indexed earlier held node ids outside [0, num_nodes)")


_graph_cache: "dict[tuple, GraphIndex]" = {}
_GRAPH_CACHE_MAX = 8
_prefetched: "dict[tuple, tuple]" = {}          # key of graph_index -> (GraphIndex built on the side stream, its completion event)
_prefetch_streams: "dict[int, torch.cuda.Stream]" = {}


def prefetch_graph_index(edge_index: torch.Tensor, num_nodes: int) -> bool:
    """Index a LATER mini-batch's graph now, on a side stream, so that the one-workgroup-per-direction CSR build (55 us for a
    256-molecule batch: 7 % of a step's device time, and nothing in the step depends on it) runs beside the current step's kernels
    instead of in front of the next step's.  ``graph_index(edge_index, num_nodes, cache=False)`` on the same tensor then adopts the
    arrays after making the compute stream wait for the build.  What a data loader's worker would do; ``harness.train_graph_batches``
    calls it for batch k + 1 before it issues step k.  Small graphs only (``kagnn_csr_small_ok``); returns whether it was queued."""
    if not (edge_index.is_cuda and edge_index.dim() == 2 and edge_index.size(0) == 2 and edge_index.dtype == torch.int64 and _SMALL_CSR):
        return False
    if not _lib.load().kagnn_csr_small_ok(int(edge_index.size(1)), int(num_nodes)):
        return False
    dev = edge_index.device
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes), dev.index)
    if key in _prefetched:
        return True
    while len(_prefetched) >= 4:                 # (never consumed: a loop that ended, an evaluation pass that took another path)
        _prefetched.pop(next(iter(_prefetched)))
    side = _prefetch_streams.get(dev.index)
    if side is None:
        side = _prefetch_streams[dev.index] = torch.cuda.Stream(device=dev)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))        # edge_index is complete once everything queued so far has run
    with torch.cuda.stream(side):
        side.wait_event(ready)
        g = GraphIndex(edge_index, num_nodes, defer_validation=True)
        done = torch.cuda.Event()
        done.record(side)
    edge_index.record_stream(side)
    _prefetched[key] = (g, done)
    return True


def graph_index(edge_index: torch.Tensor, num_nodes: int, cache: bool = True) -> GraphIndex:
    """Cached GraphIndex keyed on the identity and version of ``edge_index`` (SURVEY 8(b) ownership): full-batch node
    models see the same ``edge_index`` every epoch.  ``cache=False`` (the graph-level models' mini-batches, which never
    repeat): build and drop -- a cache that never hits only pins the last 8 batches and both of their CSRs."""
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes),
           edge_index.device.index)
    if not cache:
        hit = _prefetched.pop(key, None) if _prefetched else None
        if hit is not None:                     # built ahead on the side stream (prefetch_graph_index): wait for it, adopt its arrays
            g, ev = hit
            cur = torch.cuda.current_stream(edge_index.device)
            cur.wait_event(ev)
            for t in (g.rowptr, g.col, g.perm, g.rowptr_t, g.col_t, g.perm_t):
                t.record_stream(cur)            # (allocated under the side stream, read from now on by this one)
            return g
        return GraphIndex(edge_index, num_nodes, defer_validation=True)
    g = _graph_cache.get(key)
    if g is None:
        if len(_graph_cache) >= _GRAPH_CACHE_MAX:
            _graph_cache.pop(next(iter(_graph_cache)))
        g = GraphIndex(edge_index, num_nodes)
        g._keepalive = edge_index     # the key holds a raw pointer: pin the tensor it names
        _graph_cache[key] = g
    return g


def clear_graph_cache() -> None:
    _graph_cache.clear()
    _prefetched.clear()
    _gcn_cache.clear()


class WeightedGcnGraph:
    """``gcn_norm`` with edge weights (torch_geometric 2.5.3 semantics, restated in oracle/kan_oracle.py).

    Dense ``edge_index`` (``gcn_norm`` -> ``add_remaining_self_loops``): non-loop edges keep their weight, every node
    gets exactly one self loop (an existing loop keeps its weight, otherwise 1).  Sparse adjacency
    (``sparse_loops=True``; ``gcn_norm`` -> ``add_self_loops`` + coalesce): every node gets a weight-1 loop ADDED to
    whatever its diagonal entry already holds.  Either way ``deg`` = weighted in-degree, ``dis = deg^-1/2`` (inf -> 0).
    The augmented edge list is indexed once (CSR + transpose); built with a handful of torch ops -- per graph, not
    per step."""

    def __init__(self, edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int,
                 sparse_loops: bool = False):
        _need_cuda(edge_index)
        dev = edge_index.device
        src, dst = edge_index[0], edge_index[1]
        w = (torch.ones(src.numel(), dtype=torch.float32, device=dev) if edge_weight is None
             else edge_weight.to(device=dev, dtype=torch.float32))
        keep = src != dst
        if sparse_loops:         # coalesced input: at most one (i, i) entry per node
            loop_w = torch.ones(num_nodes, dtype=torch.float32, device=dev).index_add_(0, src[~keep], w[~keep])
        else:
            loop_w = torch.ones(num_nodes, dtype=torch.float32, device=dev)
            loop_w[src[~keep]] = w[~keep]
        ar = torch.arange(num_nodes, dtype=src.dtype, device=dev)
        ei = torch.stack([torch.cat([src[keep], ar]), torch.cat([dst[keep], ar])]).contiguous()
        self.weight = torch.cat([w[keep], loop_w]).contiguous()
        deg = torch.zeros(num_nodes, dtype=torch.float32, device=dev).index_add_(0, ei[1], self.weight)
        dis = deg.pow(-0.5)
        self.dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis).contiguous()
        self.graph = GraphIndex(ei, num_nodes)
        self._keepalive = (ei,)


_gcn_cache: "dict[tuple, WeightedGcnGraph]" = {}


def weighted_gcn_graph(edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int) -> WeightedGcnGraph:
    """Cached per (edge_index, edge_weight) identity / version.  ``edge_index`` may also be a torch sparse COO
    matrix, read the way torch_geometric reads a sparse ``adj_t``: entry (i, j) = weight of the edge j -> i (the
    form the reference's gcn timing branch passes, ``time_model.py:70-80``); its self loops follow
    ``add_self_loops`` (+1 on the diagonal), see ``WeightedGcnGraph``."""
    if isinstance(edge_index, torch.Tensor) and edge_index.is_sparse:
        key = (id(edge_index), int(num_nodes))
        hit = _gcn_cache.get(key)
        if hit is not None and hit._owner is edge_index:
            return hit
        adj = edge_index.coalesce()
        idx = adj.indices()
        hit = WeightedGcnGraph(torch.stack([idx[1], idx[0]]).contiguous(), adj.values(), num_nodes, sparse_loops=True)
        hit._owner = edge_index
    else:
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes),
               None if edge_weight is None else (edge_weight.data_ptr(), edge_weight._version))
        hit = _gcn_cache.get(key)
        if hit is not None:
            return hit
        hit = WeightedGcnGraph(edge_index, edge_weight, num_nodes)
        hit._owner = (edge_index, edge_weight)
    if len(_gcn_cache) >= _GRAPH_CACHE_MAX:
        _gcn_cache.pop(next(iter(_gcn_cache)))
    _gcn_cache[key] = hit
    return hit


# ======================================================================== aggregation
def _aggregate_raw(x, g: GraphIndex, transposed, self_scale, edge_weight, in_scale, out_scale, bias,
                   skip_self, out_dtype=torch.float32, addend=None) -> torch.Tensor:
    x = _rows(x, allow_bf16=True)
    if x.size(0) != g.num_nodes:
        raise ValueError(f"x has {x.size(0)} rows but the graph has {g.num_nodes} nodes")
    rowptr, col, _, hub, nhub = g.side(transposed)
    if addend is not None and (x.dtype != torch.float32 or out_dtype != torch.float32):
        return _aggregate_csr(x, rowptr, col, hub, nhub, g.hub_threshold, self_scale, edge_weight, in_scale, out_scale, bias,
                              skip_self, out_dtype) + addend.to(out_dtype)
    return _aggregate_csr(x, rowptr, col, hub, nhub, g.hub_threshold, self_scale, edge_weight, in_scale, out_scale, bias,
                          skip_self, out_dtype, addend)


def aggregate_sum_affine(x, g: GraphIndex, self_scale: float, affine: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """``aggregate_sum(x * affine[0] + affine[1], g, self_scale)`` without that matrix (``kagnn_aggregate_sum_affine``: the affine is
    applied to the row sums -- ``a * (self * x_i + sum_j x_j) + (self + deg_i) * b``).  Forward only (no autograd): inside the
    fused convolution node the library makes this call itself; this wrapper serves tests and the C-ABI documentation."""
    _need_cuda(x, affine)
    x = _rows(x.detach())
    rowptr, col, _, hub, nhub = g.side(transposed)
    n, f = x.shape
    out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    ws = _ws(_sizes("kagnn_aggregate_workspace_bytes", nhub, f), x.device) if nhub else None
    with _device_of(x):
        _call("kagnn_aggregate_sum_affine", _ptr(x), _ld(x), _ptr(out), f, _ptr(rowptr), _ptr(col), n, f, float(self_scale),
              _ptr(affine[0]), _ptr(affine[1]), _ptr(hub) if nhub else None, nhub, g.hub_threshold, None, 0,
              _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    return out


def _bf16_rows_ok(t: torch.Tensor) -> bool:
    return t.size(1) % 8 == 0 and t.size(1) <= 512 and _ld(t) % 8 == 0 and t.data_ptr() % 16 == 0


def to_bf16_rows(x: torch.Tensor) -> torch.Tensor:
    """fp32 rows -> bf16 (round to nearest even) in one pass (kagnn_rows_to_bf16)"""
    x = _rows(x)
    if not _bf16_rows_ok(x.new_empty((1, x.size(1)), dtype=torch.bfloat16)) or _ld(x) % 4 or x.data_ptr() % 16:
        return x.to(torch.bfloat16)
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _call("kagnn_rows_to_bf16", _ptr(x), _ld(x), _ptr(y), x.size(1), x.size(0), x.size(1), _stream())
    return y


def _aggregate_csr(x, rowptr, col, hub, nhub, hub_threshold, self_scale, edge_weight, in_scale, out_scale, bias,
                   skip_self, out_dtype=torch.float32, addend=None) -> torch.Tensor:
    n, f = x.shape
    if rowptr.numel() != n + 1:
        raise ValueError(f"x has {n} rows but the graph has {rowptr.numel() - 1} nodes")
    if x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16:
        if x.dtype != torch.bfloat16:
            x = to_bf16_rows(x)
        out = torch.empty((n, f), dtype=out_dtype, device=x.device)
        if not (_bf16_rows_ok(x) and (out_dtype != torch.bfloat16 or _bf16_rows_ok(out))):
            # widths the bf16 kernels do not take: through fp32 (same values, one more pass)
            return _aggregate_csr(x.float(), rowptr, col, hub, nhub, hub_threshold, self_scale, edge_weight, in_scale,
                                  out_scale, bias, skip_self).to(out_dtype)
        ws = _ws(_sizes("kagnn_aggregate_workspace_bytes", nhub, f), x.device) if nhub else None
        _call("kagnn_aggregate_sum_bf16", _ptr(x), _ld(x), _ptr(out), f,
              _lib.DTYPE_BF16 if out_dtype == torch.bfloat16 else _lib.DTYPE_F32, _ptr(rowptr), _ptr(col),
              _ptr(edge_weight), n, f, float(self_scale), _ptr(in_scale), _ptr(out_scale), _ptr(bias), int(skip_self),
              _ptr(hub) if nhub else None, nhub, hub_threshold, _ptr(ws), ws.numel() if nhub else 0, _stream())
        return out
    out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    ws = _ws(_sizes("kagnn_aggregate_workspace_bytes", nhub, f), x.device) if nhub else None   # per-segment partial sums
    if addend is None:
        _call("kagnn_aggregate_sum", _ptr(x), _ld(x), _ptr(out), f, _ptr(rowptr), _ptr(col),
              _ptr(edge_weight), n, f, float(self_scale), _ptr(in_scale), _ptr(out_scale), _ptr(bias),
              int(skip_self), _ptr(hub) if nhub else None, nhub, hub_threshold, _ptr(ws), ws.numel() if nhub else 0, _stream())
    else:
        _call("kagnn_aggregate_sum_add", _ptr(x), _ld(x), _ptr(out), f, _ptr(rowptr), _ptr(col),
              _ptr(edge_weight), n, f, float(self_scale), _ptr(in_scale), _ptr(out_scale), _ptr(bias),
              int(skip_self), _ptr(hub) if nhub else None, nhub, hub_threshold, _ptr(addend), _ld(addend), _ptr(ws),
              ws.numel() if nhub else 0, _stream())
    return out


class _AggregateFn(Function):
    """out = out_scale * (self_scale*s*x_i + sum_e w_e * s_j * x_j) + bias over the by-dst CSR;
    backward is the same kernel on the transposed CSR with in/out scales swapped."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x, bias, g, self_scale, edge_weight, in_scale, out_scale, skip_self):
        _need_cuda(x)
        ctx.g, ctx.self_scale, ctx.skip_self = g, self_scale, skip_self
        ctx.in_scale, ctx.out_scale = in_scale, out_scale
        ctx.edge_weight_t = None
        if edge_weight is not None:
            w = edge_weight.to(torch.float32)
            ctx.edge_weight_t = w[g.perm_t.long()].contiguous()
            edge_weight = w[g.perm.long()].contiguous()
        ctx.x_dtype = x.dtype
        return _aggregate_raw(x, g, False, self_scale, edge_weight, in_scale, out_scale, bias, skip_self)

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gout):
        gx = gb = None
        if ctx.needs_input_grad[0]:
            gx = _aggregate_raw(gout, ctx.g, True, ctx.self_scale, ctx.edge_weight_t, ctx.out_scale,
                                ctx.in_scale, None, ctx.skip_self, out_dtype=ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            gb = gout.sum(0)
        return gx, gb, None, None, None, None, None, None


def aggregate_sum(x, g: GraphIndex, self_scale: float = 1.0, edge_weight=None, in_scale=None,
                  out_scale=None, bias=None, skip_self_loops: bool = False) -> torch.Tensor:
    if torch.compiler.is_compiling():
        from . import library
        return library.aggregate(x, g, self_scale, edge_weight, in_scale, out_scale, bias, skip_self_loops)
    return _AggregateFn.apply(x, bias, g, float(self_scale), edge_weight, in_scale, out_scale,
                              bool(skip_self_loops))


class _GineFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, x, edge_attr, g, self_scale):
        _need_cuda(x, edge_attr)
        x, ea = _rows(x), _rows(edge_attr)
        n, f = x.shape
        if ea.shape != (g.num_edges, f):
            raise ValueError("edge_attr must be [E, F] with F == x.size(1)")
        out = torch.empty((n, f), dtype=torch.float32, device=x.device)
        _call("kagnn_aggregate_gine", _ptr(x), _ld(x), _ptr(ea), _ld(ea), _ptr(out), f,
                  _ptr(g.rowptr), _ptr(g.col), _ptr(g.perm), n, f, float(self_scale), _stream())
        ctx.save_for_backward(x, ea)
        ctx.g, ctx.self_scale = g, self_scale
        return out

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gout):
        x, ea = ctx.saved_tensors
        g = ctx.g
        gout = _rows(gout)
        n, f = x.shape
        gx = torch.empty((n, f), dtype=torch.float32, device=x.device)
        gea = torch.empty((g.num_edges, f), dtype=torch.float32, device=x.device) \
            if ctx.needs_input_grad[1] else None
        _call("kagnn_aggregate_gine_bwd", _ptr(x), _ld(x), _ptr(ea), _ld(ea), _ptr(gout), _ld(gout),
                  _ptr(gx), f, _ptr(gea), f, _ptr(g.rowptr_t), _ptr(g.col_t), _ptr(g.perm_t), n, f,
                  float(ctx.self_scale), _stream())
        return gx, gea, None, None


def aggregate_gine(x, edge_attr, g: GraphIndex, self_scale: float = 1.0) -> torch.Tensor:
    return _GineFn.apply(x, edge_attr, g, float(self_scale))


# ======================================================================== pooling
def segment_ptr(batch: torch.Tensor, num_graphs: int) -> torch.Tensor:
    """Offsets of a SORTED batch vector (torch_geometric's DataLoader emits it sorted): ptr[b] = the first node of graph b.
    One binary search per offset (round 5: bincount + cumsum + zero-fill + slice copy were four launches per mini-batch)."""
    marks = torch.arange(num_graphs + 1, dtype=batch.dtype, device=batch.device)
    return torch.searchsorted(batch, marks, out_int32=True)


def _segment_pool_raw(x, seg, mean):
    b, f = seg.numel() - 1, x.size(1)
    out = torch.empty((b, f), dtype=torch.float32, device=x.device)
    _call("kagnn_segment_pool", _ptr(x), _ld(x), _ptr(out), f, _ptr(seg), b, f, int(mean), _stream())
    return out


def _segment_broadcast_raw(gout, seg, n, mean):
    b, f = gout.shape
    gx = torch.empty((n, f), dtype=torch.float32, device=gout.device)
    _call("kagnn_segment_broadcast", _ptr(gout), _ld(gout), _ptr(gx), f, _ptr(seg), b, f, int(mean), _stream())
    return gx


class _SegmentPoolFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, x, seg, mean):
        _need_cuda(x, seg)
        x = _rows(x)
        ctx.seg, ctx.mean, ctx.n = seg, mean, x.size(0)
        return _segment_pool_raw(x, seg, mean)

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gout):
        return _segment_broadcast_raw(_rows(gout), ctx.seg, ctx.n, ctx.mean), None, None


def segment_pool(x, seg_ptr, mean: bool = False) -> torch.Tensor:
    if torch.compiler.is_compiling():
        from . import library
        return library.segment_pool(x, seg_ptr, bool(mean))
    return _SegmentPoolFn.apply(x, seg_ptr, bool(mean))


# ======================================================================== efficient-KAN layer
def kan_pack_chain(layers, grid_size: int, spline_order: int, mode: int):
    """Pack the weights of all layers of a KAN chain in ONE launch (``kagnn_kan_pack_batch``): ``layers`` is a list of
    ``(base_weight, spline_weight, spline_scaler_or_None)``.  Returns ``[(pack_fwd, pack_dx, key), ...]`` to hand to
    ``kan_linear(..., packed=)``, or ``None`` when the shapes are not covered (each layer then packs itself)."""
    n = len(layers)
    if not (2 <= n <= 8) or not split_like(mode) or spline_order != 3 or grid_size + spline_order > 8:
        return None
    if any(sw.size(0) > 64 or not sw.is_cuda for _, sw, _ in layers):
        return None
    dev = layers[0][1].device
    keep, packs = [], []
    arr = lambda vals, ty: (ty * n)(*vals)
    bws, sws, scs, ins, outs, pfs, pds = [], [], [], [], [], [], []
    for bw, sw, sc in layers:
        bw_c, sw_c = bw.detach().contiguous(), sw.detach().contiguous()
        sc_c = None if sc is None else sc.detach().contiguous()
        keep += [bw_c, sw_c, sc_c]
        fout, fin = sw_c.size(0), sw_c.size(1)
        fb, db = _sizes("kagnn_kan_pack_bytes", fin, fout, grid_size, spline_order, mode, outputs=2)
        pf, pd = _ws(fb, dev), _ws(db, dev)
        packs.append((pf, pd, _weights_key(bw, sw, sc)))
        bws.append(bw_c.data_ptr()); sws.append(sw_c.data_ptr()); scs.append(0 if sc_c is None else sc_c.data_ptr())
        ins.append(fin); outs.append(fout); pfs.append(pf.data_ptr()); pds.append(pd.data_ptr())
    vp = ctypes.c_void_p
    with _device_of(layers[0][1]):
        _call("kagnn_kan_pack_batch", n, arr(bws, vp), arr(sws, vp), arr(scs, vp), arr(ins, ctypes.c_int32),
              arr(outs, ctypes.c_int32), int(grid_size), int(spline_order), int(mode), arr(pfs, vp), arr(pds, vp), _stream())
    return packs


def _weights_key(bw, sw, sc):
    return tuple((t.data_ptr(), t._version) for t in (bw, sw, sc) if t is not None)


# raw (no autograd) pieces of the KANLinear forward / backward: shared by the autograd.Function below (eager) and by the
# torch.library ops of kagnn_amd/library.py (torch.compile sees those as opaque ops)
def _kan_fwd_raw(x, bw, sw, sc, knots, grid_size, spline_order, mode, packed=None, pack_key=None, moments=False, out=None):
    """-> (y, pack_dx): forward output and the input-gradient pack of the current weights; with ``moments`` also the
    column moments of y, ``(mean [out], M2 [out])`` (``kagnn_kan_linear_fwd_moments``: the BatchNorm1d statistics)"""
    n, fin = x.shape
    fout = sw.size(0)
    if packed is not None and packed[2] == pack_key:
        pack_f, pack_d = packed[0], packed[1]         # packed with its chain (kan_pack_chain), weights unchanged since
    else:
        fb, db = _sizes("kagnn_kan_pack_bytes", fin, fout, grid_size, spline_order, mode, outputs=2)
        pack_f, pack_d = _ws(fb, x.device), _ws(db, x.device)
        _call("kagnn_kan_pack", _ptr(bw), _ptr(sw), _ptr(sc), fin, fout, grid_size, spline_order, mode,
              _ptr(pack_f), _ptr(pack_d), _stream())
    if out is not None and (out.shape != (n, fout) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device):
        raise ValueError("out must be a contiguous fp32 [rows, out_features] tensor on the input's device")
    y = out if out is not None else torch.empty((n, fout), dtype=torch.float32, device=x.device)
    if moments:
        mom = torch.empty((2, fout), dtype=torch.float32, device=x.device)
        wb = _sizes("kagnn_kan_fwd_moments_workspace_bytes", n, fin, fout, grid_size, spline_order, mode)
        ws = _ws(wb, x.device) if wb else None
        _call("kagnn_kan_linear_fwd_moments", _ptr(x), _ld(x), n, _ptr(knots), fin, fout, grid_size, spline_order, mode,
              _ptr(pack_f), _ptr(y), fout, _ptr(mom[0]), _ptr(mom[1]), _ptr(ws), wb, _stream())
        return y, pack_d, mom
    wb = _sizes("kagnn_kan_fwd_workspace_bytes", n, fin, fout, grid_size, spline_order, mode)
    ws = _ws(wb, x.device) if wb else None
    _call("kagnn_kan_linear_fwd", _ptr(x), _ld(x), n, _ptr(knots), fin, fout, grid_size,
          spline_order, mode, _ptr(pack_f), _ptr(y), fout, _ptr(ws), wb, _stream())
    return y, pack_d


def _kan_bwd_input_raw(x, gy, knots, pack_d, fin, fout, G, K, mode, bf16_out=False, x_affine=None):
    n = x.size(0)
    gx = torch.empty((n, fin), dtype=torch.bfloat16 if bf16_out else torch.float32, device=x.device)
    if x_affine is not None:           # the layer input is x_affine[0] * x + x_affine[1] (AffineRows); gx is w.r.t. that input
        _call("kagnn_kan_linear_bwd_input_affine", _ptr(x), _ld(x), _ptr(x_affine), _ptr(gy), _ld(gy), n, _ptr(knots), fin,
              fout, G, K, mode, _ptr(pack_d), _ptr(gx), fin, _lib.DTYPE_F32, _stream())
        return gx
    _call("kagnn_kan_linear_bwd_input", _ptr(x), _ld(x), _ptr(gy), _ld(gy), n, _ptr(knots), fin,
          fout, G, K, mode, _ptr(pack_d), _ptr(gx), fin, _lib.DTYPE_BF16 if bf16_out else _lib.DTYPE_F32, _stream())
    return gx


def _kan_bwd_input_sums_raw(x, gy, knots, pack_d, fin, fout, G, K, mode, x_affine, ns):
    """``_kan_bwd_input_raw(x_affine=)`` + the two column sums of gx the folded norm's backward starts from (``ns``: its NormSums)"""
    n = x.size(0)
    gx = torch.empty((n, fin), dtype=torch.float32, device=x.device)
    sums = torch.empty((2, fin), dtype=torch.float32, device=x.device)
    wb = _sizes("kagnn_kan_bwd_input_sums_workspace_bytes", n, fin)
    ws = _ws(wb, x.device)
    _call("kagnn_kan_linear_bwd_input_affine_sums", _ptr(x), _ld(x), _ptr(x_affine), _ptr(ns.mean), _ptr(ns.rstd), _ptr(gy), _ld(gy), n,
          _ptr(knots), fin, fout, G, K, mode, _ptr(pack_d), _ptr(gx), fin, _ptr(sums), _ptr(ws), ws.numel(), _stream())
    return gx, sums


def _kan_bwd_weight_raw(x, gy, knots, sw, sc, fin, fout, G, K, mode, has_base, x_affine=None):
    n = x.size(0)
    ws = _ws(_sizes("kagnn_kan_bwd_weight_workspace_bytes", n, fin, fout, G, K, mode), x.device)
    gbw = torch.empty((fout, fin), dtype=torch.float32, device=x.device) if has_base else None
    gsw = torch.empty((fout, fin, G + K), dtype=torch.float32, device=x.device)
    gsc = None if sc is None else torch.empty((fout, fin), dtype=torch.float32, device=x.device)
    if x_affine is not None:
        _call("kagnn_kan_linear_bwd_weight_affine", _ptr(x), _ld(x), _ptr(x_affine), _ptr(gy), _ld(gy), n, _ptr(knots), fin,
              fout, G, K, mode, _ptr(sw), _ptr(sc), _ptr(gbw), _ptr(gsw), _ptr(gsc), _ptr(ws), ws.numel(), _stream())
        return gbw, gsw, gsc
    _call("kagnn_kan_linear_bwd_weight", _ptr(x), _ld(x), _ptr(gy), _ld(gy), n, _ptr(knots), fin,
          fout, G, K, mode, _ptr(sw), _ptr(sc), _ptr(gbw), _ptr(gsw), _ptr(gsc), _ptr(ws),
          ws.numel(), _stream())
    return gbw, gsw, gsc


class _OutBuffer:
    """a caller-owned destination for a forward output, hidden from autograd (see _KANLinearFn.forward)"""

    def __init__(self, t: torch.Tensor):
        self.t = t

    def view(self, n: int, f: int) -> torch.Tensor:
        if self.t.numel() < n * f or self.t.dtype != torch.float32 or not self.t.is_contiguous():
            raise ValueError("out must be a contiguous fp32 buffer of at least rows * out_features elements")
        return self.t.detach().view(-1)[: n * f].view(n, f)


class _KANLinearFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, x, base_weight, spline_weight, spline_scaler, knots, grid_size, spline_order, mode, packed=None, out=None):
        _need_cuda(x, base_weight, spline_weight, spline_scaler, knots)
        x = _rows(x)
        fin = x.size(1)
        fout = spline_weight.size(0)
        bw = None if base_weight is None else base_weight.contiguous()     # None: no SiLU branch (coefficient groups)
        sw = spline_weight.contiguous()
        sc = None if spline_scaler is None else spline_scaler.contiguous()
        # out: an _OutBuffer (NOT a tensor argument: autograd must not see the caller's buffer -- e.g. a peer-mapped exchange
        # buffer -- as an input modified in place); the output is a fresh view of it made here, under the forward's no_grad
        y, pack_d = _kan_fwd_raw(x, bw, sw, sc, knots, grid_size, spline_order, mode, packed,
                                 _weights_key(base_weight, spline_weight, spline_scaler) if packed is not None else None,
                                 out=None if out is None else out.view(x.size(0), fout))
        ctx.save_for_backward(x, sw, sc, knots, pack_d)
        ctx.dims = (fin, fout, grid_size, spline_order, mode)
        ctx.has_base = bw is not None
        return y

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gy):
        x, sw, sc, knots, pack_d = ctx.saved_tensors
        fin, fout, G, K, mode = ctx.dims
        gy = _rows(gy)
        gx = gbw = gsw = gsc = None
        if ctx.needs_input_grad[0]:
            gx = _kan_bwd_input_raw(x, gy, knots, pack_d, fin, fout, G, K, mode)
        if any(ctx.needs_input_grad[1:4]):
            gbw, gsw, gsc = _kan_bwd_weight_raw(x, gy, knots, sw, sc, fin, fout, G, K, mode, ctx.has_base)
        return gx, gbw, gsw, gsc, None, None, None, None, None, None


_LAYER_ABI = os.environ.get("KAGNN_LAYER_ABI", "1") != "0"     # 1: kagnn_gin_kan_layer_fwd / _bwd (one library call each way)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _gin_kan_layer_fwd_raw(xg, g, self_scale, knots, grid_size, spline_order, mode, layers, widths, moments, in_affine=None):
    """``kagnn_gin_kan_layer_fwd``: -> (acts [h0, ..., y], input-gradient packs per layer, column moments of y or None)"""
    n, dev, nl = xg.size(0), xg.device, len(layers)
    acts = [torch.empty((n, w), dtype=torch.float32, device=dev) for w in widths]
    pfs, pds = [], []
    for i in range(nl):
        fb, db = _sizes("kagnn_kan_pack_bytes", widths[i], widths[i + 1], grid_size, spline_order, mode, outputs=2)
        pfs.append(_ws(fb, dev)); pds.append(_ws(db, dev))
    warr = (ctypes.c_int32 * (nl + 1))(*widths)
    wf, _ = _sizes("kagnn_gin_kan_layer_workspace_bytes", n, nl, tuple(widths), grid_size, spline_order, mode,
                   g.num_hub_seg, g.num_hub_seg_t, outputs=2)
    ws = _ws(wf, dev)
    mom = torch.empty((2, widths[nl]), dtype=torch.float32, device=dev) if moments else None
    if in_affine is not None:          # the gathered rows are in_affine[0] * xg + in_affine[1]: a folded BatchNorm1d (AffineRows)
        _call("kagnn_gin_kan_layer_fwd_affine", _ptr(xg), _ld(xg), n,
              _ptr(g.rowptr), _ptr(g.col), _ptr(g.hub_seg) if g.num_hub_seg else None, g.num_hub_seg, g.hub_threshold,
              float(self_scale), _ptr(in_affine[0]), _ptr(in_affine[1]), nl, warr, _ptr_array([l[0] for l in layers]),
              _ptr_array([l[1] for l in layers]), _ptr_array([l[2] for l in layers]), _ptr(knots), grid_size, spline_order, mode,
              _ptr_array(acts), _ptr_array(pfs), _ptr_array(pds), _ptr(mom[0]) if moments else None,
              _ptr(mom[1]) if moments else None, _ptr(ws), ws.numel(), _stream())
        return acts, pds, mom
    _call("kagnn_gin_kan_layer_fwd", _ptr(xg), _lib.DTYPE_BF16 if xg.dtype == torch.bfloat16 else _lib.DTYPE_F32, _ld(xg), n,
          _ptr(g.rowptr), _ptr(g.col), _ptr(g.hub_seg) if g.num_hub_seg else None, g.num_hub_seg, g.hub_threshold,
          float(self_scale), nl, warr, _ptr_array([l[0] for l in layers]), _ptr_array([l[1] for l in layers]),
          _ptr_array([l[2] for l in layers]), _ptr(knots), grid_size, spline_order, mode, _ptr_array(acts),
          _ptr_array(pfs), _ptr_array(pds), _ptr(mom[0]) if moments else None, _ptr(mom[1]) if moments else None,
          _ptr(ws), ws.numel(), _stream())
    return acts, pds, mom


class AffineRows:
    """A matrix that exists only as ``y * affine[0] + affine[1]`` (per-column scale and shift) -- the output of a training-mode
    ``BatchNorm1d`` whose normalising pass is folded into the kernels that read it (reference
    ``node_classification_clean/models.py:198-203``).  ``y`` is the convolution's raw output, on the tape; ``affine`` is a [2, F]
    fp32 tensor outside it.  CONVENTION: whoever consumes this object treats ``y``'s gradient slot as the gradient with respect
    to the NORMALISED rows -- the producing node (``_GinKanBnLayerFn`` in lazy mode) runs the norm's backward from there.  Only
    ``kagnn_amd.models._NodeModel.forward`` builds these, and only between nodes that follow the convention."""
    __slots__ = ("y", "affine", "stats")

    def __init__(self, y: torch.Tensor, affine: torch.Tensor, stats: Optional["NormSums"] = None):
        self.y, self.affine, self.stats = y, affine, stats     # stats: the producing norm's NormSums (or None)

    def size(self, d):
        return self.y.size(d)

    def materialise(self) -> torch.Tensor:
        """the normalised rows as an ordinary tensor (for consumers that cannot fold the affine); the gradient passes through
        unchanged, as the convention above requires"""
        return _MaterialiseAffineFn.apply(self.y, self.affine)


class NormSums:
    """Side channel between two fused convolution + norm nodes (``_GinKanBnLayerFn``), outside the tape like ``SkipGradient``:
    the backward of layer l's norm starts from two column sums of its incoming gradient g -- ``sum g`` and ``sum g * xhat`` -- which
    used to take a pass over g and the norm's input.  In the node models g is exactly what layer l+1's transposed aggregation
    writes (reference ``node_classification_clean/models.py:198-200``), so that kernel forms the sums in its epilogue
    (``kagnn_gin_kan_layer_bwd_bn_sums``).  The PRODUCING node (layer l) creates this object in its forward and fills ``mean`` /
    ``rstd``; the CONSUMING node (layer l+1) gets it with the ``AffineRows``, parks ``sums`` for the gradient tensor it returns;
    layer l's backward takes them only if the gradient it receives IS that tensor (same storage, same engine run) -- any other
    consumer of the activation makes autograd sum into a new tensor and the statistics pass runs as before."""
    __slots__ = ("mean", "rstd", "sums", "grad", "task", "version")

    def __init__(self):
        self.mean = self.rstd = self.sums = self.grad = None
        self.task = self.version = -1

    def park(self, sums: torch.Tensor, grad: torch.Tensor) -> None:
        # (holding ``grad`` keeps its storage from being reused; its version counter is recorded so that an engine that
        # accumulated another consumer's gradient INTO it in place -- same storage, new contents -- is noticed: ADVICE r04)
        self.sums, self.grad, self.task, self.version = sums, grad, graph_task_id(), grad._version

    def take(self, grad: torch.Tensor):
        sums, parked, t, ver = self.sums, self.grad, self.task, self.version
        self.sums = self.grad = None
        self.task = self.version = -1
        if sums is None or t != graph_task_id() or parked is None:
            return None
        same = (parked.data_ptr() == grad.data_ptr() and parked.shape == grad.shape and parked.stride() == grad.stride()
                and parked.dtype == grad.dtype and parked.device == grad.device
                and parked._version == ver and grad._version == ver)
        return sums if same else None


_FOLD_NORM_STATS = os.environ.get("KAGNN_FOLD_NORM_STATS", "1") != "0"


class _MaterialiseAffineFn(Function):
    @staticmethod
    def forward(ctx, y, affine):
        return torch.addcmul(affine[1], y, affine[0])

    @staticmethod
    def backward(ctx, g):
        return g, None


class _GinKanLayerFn(Function):
    """One KAN-GIN convolution -- ``KAN((1 + eps) x_i + sum_{j->i} x_j)`` -- as a single tape node over
    ``kagnn_gin_kan_layer_fwd / _bwd`` (the ``gin_kan_fused_fwd / _bwd`` of SURVEY.md 8(b)): forward = aggregation + ONE
    weight-pack launch + the chain's KANLinear forwards, backward = per layer dW / dX, then the transposed aggregation --
    one library call each way.  Saves the layer inputs only.  With ``act_bf16`` the two matrices the aggregation GATHERS
    travel as bf16: ``x`` on the way forward and ``d loss / d h0`` on the way back (written as bf16 by the
    input-gradient kernel itself, no extra pass).  ``KAGNN_LAYER_ABI=0`` composes the same kernels from the per-op entry
    points instead (bit-identical; kept for A/B)."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x, g, self_scale, knots, grid_size, spline_order, mode, act_bf16, moments, skip_gradient, *params):
        _need_cuda(x, *params)
        nl = len(params) // 3
        layers = [(params[3 * i].contiguous(), params[3 * i + 1].contiguous(), params[3 * i + 2].contiguous()) for i in range(nl)]
        xg = _rows(x, allow_bf16=True)
        if xg.size(0) != g.num_nodes:            # (the library call below indexes rowptr / x by this count)
            raise ValueError(f"x has {xg.size(0)} rows but the graph has {g.num_nodes} nodes")
        if act_bf16 and xg.dtype != torch.bfloat16 and xg.size(1) % 8 == 0 and xg.size(1) <= 512:
            xg = to_bf16_rows(xg)                # (rows the bf16 aggregation cannot take stay fp32 -- unrounded)
        if xg.dtype == torch.bfloat16 and not _bf16_rows_ok(xg):
            xg = xg.float()
        n, dev = xg.size(0), xg.device
        widths = [layers[0][1].size(1)] + [sw.size(0) for _, sw, _ in layers]
        ctx.meta = (g, self_scale, grid_size, spline_order, mode, act_bf16, nl, x.dtype, widths)
        # a second gradient of x that another tape node (the skip-concat read-out) hands over outside the tape: this node's
        # backward adds it inside the transposed aggregation's epilogue (SkipGradient)
        ctx.skip_gradient = None
        if (skip_gradient is not None and x.requires_grad and x.dtype == torch.float32 and xg.dtype == torch.float32
                and not act_bf16):
            skip_gradient.consumer = True
            ctx.skip_gradient = skip_gradient
        if not _LAYER_ABI:
            h = _aggregate_raw(xg, g, False, self_scale, None, None, None, None, False)
            packs = kan_pack_chain(layers, grid_size, spline_order, mode) if nl > 1 else None
            saved = []
            mom = None
            for i, (bw, sw, sc) in enumerate(layers):
                out = _kan_fwd_raw(h, bw, sw, sc, knots[i], grid_size, spline_order, mode,
                                   None if packs is None else packs[i], None if packs is None else packs[i][2],
                                   moments=moments and i == nl - 1)
                y, pack_d = out[0], out[1]
                mom = out[2] if len(out) > 2 else mom
                saved += [h, sw, sc, pack_d]
                h = y
            ctx.save_for_backward(*saved, knots[0])
            if moments:
                ctx.mark_non_differentiable(mom)
                return h, mom
            return h
        acts, pds, mom = _gin_kan_layer_fwd_raw(xg, g, self_scale, knots[0], grid_size, spline_order, mode, layers, widths, moments)
        saved = []
        for i in range(nl):
            saved += [acts[i], layers[i][1], layers[i][2], pds[i]]
        ctx.save_for_backward(*saved, knots[0])
        if moments:
            ctx.mark_non_differentiable(mom)
            return acts[nl], mom
        return acts[nl]

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gy, _g_moments=None):
        g, self_scale, G, K, mode, act_bf16, nl, x_dtype, widths = ctx.meta
        t = ctx.saved_tensors
        knots = t[4 * nl]
        gy = _rows(gy)
        need_x = ctx.needs_input_grad[0]
        gx_dtype = torch.bfloat16 if x_dtype == torch.bfloat16 else torch.float32
        addend = None
        if ctx.skip_gradient is not None:
            addend = ctx.skip_gradient.take()
            if addend is not None:
                addend = _rows(addend)
                if addend.shape != (gy.size(0), widths[0]) or not need_x:
                    raise RuntimeError("SkipGradient: the gradient handed over does not belong to this convolution's input")
        if not _LAYER_ABI:
            grads = [None] * (3 * nl)
            for i in reversed(range(nl)):
                h_in, sw, sc, pack_d = t[4 * i:4 * i + 4]
                fin, fout = widths[i], widths[i + 1]
                if any(ctx.needs_input_grad[10 + 3 * i:10 + 3 * i + 3]):
                    grads[3 * i], grads[3 * i + 1], grads[3 * i + 2] = _kan_bwd_weight_raw(h_in, gy, knots, sw, sc, fin, fout,
                                                                                           G, K, mode, True)
                if i > 0 or need_x:
                    bf16_out = (i == 0 and act_bf16 and split_like(mode) and K == 3 and G + K <= 8 and fout <= 128
                                and fin % 8 == 0 and fin <= 512 and _fits32(h_in, fout))      # the dX variant that stores bf16 rows (<= 512: the bf16 aggregation's limit)
                    gy = _kan_bwd_input_raw(h_in, gy, knots, pack_d, fin, fout, G, K, mode, bf16_out)
            gx = _aggregate_raw(gy, g, True, self_scale, None, None, None, None, False, out_dtype=gx_dtype, addend=addend) if need_x else None
            return (gx, None, None, None, None, None, None, None, None, None, *grads)
        n, dev = gy.size(0), gy.device
        if n != g.num_nodes:
            raise ValueError(f"the incoming gradient has {n} rows but the graph has {g.num_nodes} nodes")
        f32 = dict(dtype=torch.float32, device=dev)
        acts = [t[4 * i] for i in range(nl)]
        sws, scs, pds = [t[4 * i + 1] for i in range(nl)], [t[4 * i + 2] for i in range(nl)], [t[4 * i + 3] for i in range(nl)]
        gbw = [torch.empty((widths[i + 1], widths[i]), **f32) for i in range(nl)]
        gsw = [torch.empty((widths[i + 1], widths[i], G + K), **f32) for i in range(nl)]
        gsc = [torch.empty((widths[i + 1], widths[i]), **f32) for i in range(nl)]
        gx = torch.empty((n, widths[0]), dtype=gx_dtype, device=dev) if need_x else None
        if gx is not None and gx_dtype == torch.bfloat16 and not _bf16_rows_ok(gx):
            gx = torch.empty((n, widths[0]), **f32)                       # widths the bf16 kernels do not take
        warr = (ctypes.c_int32 * (nl + 1))(*widths)
        _, wb = _sizes("kagnn_gin_kan_layer_workspace_bytes", n, nl, tuple(widths), G, K, mode, g.num_hub_seg,
                       g.num_hub_seg_t, outputs=2)
        ws = _ws(wb, dev)
        _call("kagnn_gin_kan_layer_bwd_add", _ptr(gy), _ld(gy), n, _ptr(g.rowptr_t), _ptr(g.col_t),
              _ptr(g.hub_seg_t) if g.num_hub_seg_t else None, g.num_hub_seg_t, g.hub_threshold, float(self_scale), nl, warr,
              _ptr_array(sws), _ptr_array(scs), _ptr(knots), G, K, mode, _ptr_array(acts), _ptr_array(pds), _ptr(gx),
              _lib.DTYPE_BF16 if (gx is not None and gx.dtype == torch.bfloat16) else _lib.DTYPE_F32, widths[0],
              int(bool(act_bf16) and widths[0] % 8 == 0 and widths[0] <= 512), _ptr(addend), _ld(addend) if addend is not None else 0,
              _ptr_array(gbw), _ptr_array(gsw), _ptr_array(gsc), _ptr(ws), ws.numel(), _stream())
        if gx is not None and gx.dtype != gx_dtype:
            gx = gx.to(gx_dtype)
        grads = []
        for i in range(nl):
            grads += [gbw[i], gsw[i], gsc[i]]
        return (gx, None, None, None, None, None, None, None, None, None, *grads)


class _GinKanBnLayerFn(Function):
    """``BatchNorm1d(KAN((1 + eps) x_i + sum_j x_j))`` in training mode -- a KAN-GIN convolution and the norm that follows it in
    every node model (reference ``node_classification_clean/models.py:198-200``) -- as ONE tape node.  Forward:
    ``kagnn_gin_kan_layer_fwd`` (column moments from the last kernel's epilogue) + the normalising pass.  Backward:
    ``kagnn_gin_kan_layer_bwd_bn`` -- the norm's statistics pass, then its element-wise backward INSIDE the last layer's
    input-gradient kernel (the rows are transformed as they are loaded and left for the weight gradient), so the separate
    normalisation-backward pass over [N, out] is gone.  Same values as the two nodes (``_GinKanLayerFn`` -> ``_BatchNormFn``)."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x, g, self_scale, knots, grid_size, spline_order, mode, act_bf16, skip_gradient, in_affine, lazy, in_stats, out_stats,
                bn_weight, bn_bias, running_mean, running_var, momentum, eps, *params):
        """``in_affine``: the input is an ``AffineRows`` (``x`` = its y): the aggregation folds the previous layer's norm.
        ``lazy``: do not write the normalised rows -- return ``(y, affine)`` for an ``AffineRows`` (see its convention).
        ``in_stats`` / ``out_stats``: the ``NormSums`` of the input ``AffineRows`` / a fresh one for the output (lazy mode)."""
        _need_cuda(x, bn_weight, bn_bias, running_mean, running_var, in_affine, *params)
        nl = len(params) // 3
        layers = [(params[3 * i].contiguous(), params[3 * i + 1].contiguous(), params[3 * i + 2].contiguous()) for i in range(nl)]
        xg = _rows(x, allow_bf16=True)
        if xg.size(0) != g.num_nodes:
            raise ValueError(f"x has {xg.size(0)} rows but the graph has {g.num_nodes} nodes")
        if act_bf16 and xg.dtype != torch.bfloat16 and xg.size(1) % 8 == 0 and xg.size(1) <= 512:
            xg = to_bf16_rows(xg)
        if xg.dtype == torch.bfloat16 and not _bf16_rows_ok(xg):
            xg = xg.float()
        widths = [layers[0][1].size(1)] + [sw.size(0) for _, sw, _ in layers]
        ctx.meta = (g, self_scale, grid_size, spline_order, mode, act_bf16, nl, x.dtype, widths)
        ctx.skip_gradient = None
        if (skip_gradient is not None and x.requires_grad and x.dtype == torch.float32 and xg.dtype == torch.float32
                and not act_bf16):
            skip_gradient.consumer = True
            ctx.skip_gradient = skip_gradient
        acts, pds, mom = _gin_kan_layer_fwd_raw(xg, g, self_scale, knots[0], grid_size, spline_order, mode, layers, widths, True,
                                                in_affine=in_affine)
        y = acts[nl]
        affine = None
        if lazy:
            f = widths[nl]
            mean = torch.empty(f, dtype=torch.float32, device=y.device)
            rstd = torch.empty(f, dtype=torch.float32, device=y.device)
            affine = torch.empty((2, f), dtype=torch.float32, device=y.device)
            _call("kagnn_batchnorm_stats_affine", _ptr(mom[0]), _ptr(mom[1]), y.size(0), f, _ptr(bn_weight), _ptr(bn_bias),
                  _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), _ptr(mean), _ptr(rstd), _ptr(affine), _stream())
        else:
            h, mean, rstd = _batchnorm_fwd_raw(y, bn_weight, bn_bias, running_mean, running_var, True, momentum, eps, mom)
        saved = []
        for i in range(nl):
            saved += [acts[i], layers[i][1], layers[i][2], pds[i]]
        # the norms' backward statistics travel with the gradients (NormSums): this node's own, and the previous norm's that
        # this node's transposed aggregation can produce -- fp32 rows of 17..256 columns (a multiple of 4), gradient wanted
        ctx.out_stats = ctx.in_stats = None
        if _FOLD_NORM_STATS and lazy and out_stats is not None:
            out_stats.mean, out_stats.rstd = mean, rstd
            ctx.out_stats = out_stats
        extra = []
        if (_FOLD_NORM_STATS and in_affine is not None and in_stats is not None and in_stats.mean is not None and x.requires_grad
                and x.dtype == torch.float32 and xg is x and not act_bf16 and 16 < widths[0] <= 256 and widths[0] % 4 == 0
                and x.data_ptr() % 16 == 0 and _ld(x) % 4 == 0):
            ctx.in_stats = in_stats
            extra = [x]                                   # x = the previous norm's input
        ctx.save_for_backward(*saved, knots[0], y, mean, rstd, bn_weight, *extra)
        ctx.has_bias = bn_bias is not None
        if lazy:
            ctx.mark_non_differentiable(affine)
            ctx.set_materialize_grads(False)          # (no zero-filled [2, F] "gradient" of the affine per backward: two tiny launches)
            return y, affine
        return h

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gh, _g_affine=None):
        g, self_scale, G, K, mode, act_bf16, nl, x_dtype, widths = ctx.meta
        t = ctx.saved_tensors
        knots, y, mean, rstd, bn_w = t[4 * nl:4 * nl + 5]
        if gh is None:                                # (lazy mode does not materialise absent gradients: an unused output)
            gh = torch.zeros_like(y)
        gh_in = gh
        gh = _rows(gh)
        need_x = ctx.needs_input_grad[0]
        gx_dtype = torch.bfloat16 if x_dtype == torch.bfloat16 else torch.float32
        addend = None
        if ctx.skip_gradient is not None:
            addend = ctx.skip_gradient.take()
            if addend is not None:
                addend = _rows(addend)
                if addend.shape != (gh.size(0), widths[0]) or not need_x:
                    raise RuntimeError("SkipGradient: the gradient handed over does not belong to this convolution's input")
        n, dev = gh.size(0), gh.device
        if n != g.num_nodes:
            raise ValueError(f"the incoming gradient has {n} rows but the graph has {g.num_nodes} nodes")
        f32 = dict(dtype=torch.float32, device=dev)
        acts = [t[4 * i] for i in range(nl)]
        sws, scs, pds = [t[4 * i + 1] for i in range(nl)], [t[4 * i + 2] for i in range(nl)], [t[4 * i + 3] for i in range(nl)]
        gbw = [torch.empty((widths[i + 1], widths[i]), **f32) for i in range(nl)]
        gsw = [torch.empty((widths[i + 1], widths[i], G + K), **f32) for i in range(nl)]
        gsc = [torch.empty((widths[i + 1], widths[i]), **f32) for i in range(nl)]
        gbn_w = torch.empty(widths[nl], **f32) if bn_w is not None else None
        gbn_b = torch.empty(widths[nl], **f32) if ctx.has_bias else None
        gx = torch.empty((n, widths[0]), dtype=gx_dtype, device=dev) if need_x else None
        if gx is not None and gx_dtype == torch.bfloat16 and not _bf16_rows_ok(gx):
            gx = torch.empty((n, widths[0]), **f32)
        warr = (ctypes.c_int32 * (nl + 1))(*widths)
        _, wb = _sizes("kagnn_gin_kan_layer_workspace_bytes", n, nl, tuple(widths), G, K, mode, g.num_hub_seg,
                       g.num_hub_seg_t, outputs=2)
        # statistics that came with the gradient (the next layer's aggregation made them for exactly this tensor), and the ones
        # this call makes for the previous norm
        my_sums = ctx.out_stats.take(gh_in) if ctx.out_stats is not None else None
        prev = ctx.in_stats if (ctx.in_stats is not None and gx is not None and gx.dtype == torch.float32) else None
        tail = (n, _ptr(g.rowptr_t), _ptr(g.col_t), _ptr(g.hub_seg_t) if g.num_hub_seg_t else None, g.num_hub_seg_t, g.hub_threshold,
                float(self_scale), nl, warr, _ptr_array(sws), _ptr_array(scs), _ptr(knots), G, K, mode, _ptr_array(acts), _ptr_array(pds),
                _ptr(gx), _lib.DTYPE_BF16 if (gx is not None and gx.dtype == torch.bfloat16) else _lib.DTYPE_F32, widths[0],
                int(bool(act_bf16) and widths[0] % 8 == 0 and widths[0] <= 512), _ptr(addend), _ld(addend) if addend is not None else 0,
                _ptr_array(gbw), _ptr_array(gsw), _ptr_array(gsc))
        wbn = _sizes("kagnn_gin_kan_layer_bwd_bn_workspace_bytes", n, widths[nl])
        if my_sums is None and prev is None:
            ws = _ws(wb + wbn, dev)
            _call("kagnn_gin_kan_layer_bwd_bn", _ptr(gh), _ld(gh), _ptr(y), _ld(y), _ptr(bn_w), _ptr(mean), _ptr(rstd), _ptr(gbn_w),
                  _ptr(gbn_b), *tail, _ptr(ws), ws.numel(), _stream())
        else:
            prev_sums = prev_y = None
            wst = 0
            if prev is not None:
                prev_y = t[4 * nl + 5]
                prev_sums = torch.empty((2, widths[0]), **f32)
                wst = _sizes("kagnn_gin_kan_layer_bwd_bn_sums_workspace_bytes", n, widths[0], g.num_hub_seg_t)
            ws = _ws(wb + wbn + wst, dev)
            _call("kagnn_gin_kan_layer_bwd_bn_sums", _ptr(gh), _ld(gh), _ptr(y), _ld(y), _ptr(bn_w), _ptr(mean), _ptr(rstd), _ptr(gbn_w),
                  _ptr(gbn_b), _ptr(my_sums), _ptr(prev_y), _ld(prev_y) if prev_y is not None else 0,
                  _ptr(prev.mean) if prev is not None else None, _ptr(prev.rstd) if prev is not None else None, _ptr(prev_sums),
                  *tail, _ptr(ws), ws.numel(), _stream())
            if prev is not None:
                prev.park(prev_sums, gx)
        if gx is not None and gx.dtype != gx_dtype:
            gx = gx.to(gx_dtype)
        grads = []
        for i in range(nl):
            grads += [gbw[i], gsw[i], gsc[i]]
        return (gx, None, None, None, None, None, None, None, None, None, None, None, None, gbn_w, gbn_b, None, None, None, None, *grads)


_GRAPH_TASK_ID = getattr(torch._C, "_current_graph_task_id", None)


def graph_task_id() -> int:
    """id of the autograd engine run this is called from (-1 outside a backward pass; always -1 on a torch build without the
    hook: the tags below then compare equal and the objects behave as before they were tagged)"""
    return _GRAPH_TASK_ID() if _GRAPH_TASK_ID is not None else -1


class SkipGradient:
    """A gradient that travels from one tape node to another OUTSIDE the tape.  In the skip-concat node models an activation
    ``h_l`` feeds the next convolution AND the read-out (reference ``node_classification_clean/models.py:196-202``); autograd
    would materialise both gradients and add them in a pass of its own (0.12 ms per layer at 1M x 64).  Instead: the
    convolution's fused node registers as the ``consumer`` of this object in its forward; the read-out's node
    (``_KANLinearPartsFn``) -- whose backward always runs first, the convolution's output feeds it -- leaves its gradient of
    ``h_l`` in ``grad`` and reports none to the tape; the convolution's backward adds it inside the transposed aggregation's
    epilogue (``kagnn_gin_kan_layer_bwd_add``).  Bit-identical to the separate sum."""
    __slots__ = ("grad", "consumer", "task")

    def __init__(self):
        self.grad = None
        self.consumer = False
        self.task = -1                  # the engine run (graph task id) that parked ``grad``

    def park(self, grad) -> None:
        self.grad, self.task = grad, graph_task_id()

    def take(self):
        """the parked gradient if THIS backward pass parked it, else None; always leaves the object empty.  A pass that
        prunes the consumer (``backward(inputs=[...])``) leaves a parked gradient behind: a later pass must not add it to a
        loss it does not belong to -- it is dropped here instead."""
        g, t = self.grad, self.task
        self.grad, self.task = None, -1
        return g if t == graph_task_id() else None


_KNOTS_EQUAL: dict = {}


def _same_knots(layers, knots) -> bool:
    """all layers of the chain on ONE knot vector (``KAN`` hands every layer the same grid_range)?  One host comparison per
    distinct set of knot tensors, cached."""
    key = tuple((k.data_ptr(), k._version) for k in knots)
    hit = _KNOTS_EQUAL.get(key)
    if hit is None:
        hit = all(bool(torch.equal(k, knots[0])) for k in knots[1:])
        if len(_KNOTS_EQUAL) > 256:
            _KNOTS_EQUAL.clear()
        _KNOTS_EQUAL[key] = hit
    return hit


def gin_kan_layer(x, g: GraphIndex, self_scale: float, chain, act_dtype: Optional[torch.dtype] = None,
                  moments: bool = False, skip_gradient: Optional["SkipGradient"] = None, batch_norm=None,
                  in_affine: Optional[torch.Tensor] = None, lazy_norm: bool = False, in_stats: Optional["NormSums"] = None):
    """``chain(aggregate_sum(x, g, self_scale))`` for a ``kagnn_amd.KAN`` chain as ONE autograd node, or ``None`` when the
    chain is outside what the fused node covers (adaptive grids, > 16 coefficients, mixed precisions): the caller then
    composes the ops.  ``moments=True`` -> ``(y, moments)`` with the [2, out] column moments (mean, M2) of y from the
    last forward kernel's epilogue, for ``batch_norm(..., moments=...)``.
    ``batch_norm=(weight, bias, running_mean, running_var, momentum, eps)``: the training-mode BatchNorm1d that follows the
    convolution joins the node (``_GinKanBnLayerFn``: its element-wise backward runs inside the last input-gradient kernel)
    and the normalised rows are returned."""
    layers = list(chain.layers)
    first = layers[0]
    mode = first.precision if first.precision is not None else default_precision()
    act = default_activation_dtype() if act_dtype is None else act_dtype
    if any(l.precision != first.precision or l.grid_size != first.grid_size or l.spline_order != first.spline_order
           or l.grid_size + l.spline_order > 16 or not l.enable_standalone_scale_spline for l in layers):
        return None
    knots = [l._knots() for l in layers]
    if any(k.dim() != 1 or k.numel() != knots[0].numel() for k in knots):
        return None
    if not _same_knots(layers, knots):
        return None
    width = max(max(l.in_features, l.out_features) for l in layers)
    if split_like(mode) and width > 7680:
        return None
    params = []
    for l in layers:
        params += [l.base_weight, l.spline_weight, l.spline_scaler]
    if batch_norm is not None:
        # (the library call keeps the norm's transformed rows in the chain's ping-pong gradient matrices, which are as wide as
        # the widest layer INPUT: a chain whose output is wider than every input stays on the two nodes)
        if not _LAYER_ABI or x.size(0) < 2 or layers[-1].out_features > max(l.in_features for l in layers):
            return None
        bw_, bb_, rm_, rv_, mom_, eps_ = batch_norm() if callable(batch_norm) else batch_norm      # (callable: evaluated only now that the node is certain -- the caller's per-call bookkeeping)
        out_stats = NormSums() if lazy_norm else None
        out = _GinKanBnLayerFn.apply(x, g, float(self_scale), knots, first.grid_size, first.spline_order, int(mode),
                                     act == torch.bfloat16 or x.dtype == torch.bfloat16, skip_gradient, in_affine, bool(lazy_norm),
                                     in_stats, out_stats, bw_, bb_, rm_, rv_, float(mom_), float(eps_), *params)
        return AffineRows(out[0], out[1], out_stats) if lazy_norm else out
    return _GinKanLayerFn.apply(x, g, float(self_scale), knots, first.grid_size, first.spline_order, int(mode),
                                act == torch.bfloat16 or x.dtype == torch.bfloat16, bool(moments), skip_gradient, *params)


def kan_linear(x, base_weight, spline_weight, spline_scaler, knots, grid_size: int, spline_order: int,
               mode: Optional[int] = None, packed=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = silu(x) @ base_weight.T + bases(x) @ (spline_weight*scaler).T  (ekan.py:154-162).
    ``knots`` is ONE row of the layer's grid buffer (uniform), fp32 [G+2k+1] on the device -- or the whole
    buffer [in, G+2k+1] when its rows differ / are non-uniform (after ``update_grid``): that runs the exact-fp32
    kernels on per-feature knots (``KAGNN_PREC_FP32_GRID``)."""
    if knots.dim() == 2:
        if knots.size(0) != x.size(-1):
            raise AssertionError("grid must be [in_features, G+2k+1]")
        mode = PREC_FP32_GRID
    if mode is None:
        mode = default_precision()
    if split_like(mode) and not _fits32(x, base_weight.size(0)):
        mode = PREC_FP32
    if not split_like(mode):
        packed = None                                    # chain packs are in the split kernels' layout
    n_coef = int(grid_size) + int(spline_order)
    if out is not None and ((split_like(mode) and n_coef > 16 and knots.dim() == 1) or torch.compiler.is_compiling()):
        raise ValueError("out= is not supported for layers with more than 16 coefficients or under torch.compile")
    if split_like(mode) and n_coef > 16 and knots.dim() == 1:
        # More than 16 coefficients per feature (the reference's search space goes to grid_size 32): a uniform
        # B-spline basis function only depends on its own k+2 knots, so the layer is the SUM of layers over
        # consecutive coefficient groups, each on its slice of the knot vector -- every group has <= 16 coefficients
        # and runs on the split-precision kernels (the SiLU branch rides with the first group).
        groups = -(-n_coef // 16)
        size, extra = divmod(n_coef, groups)
        y, c0 = None, 0
        for g in range(groups):
            cg = size + (1 if g < extra else 0)
            part = _KANLinearFn.apply(x, base_weight if g == 0 else None, spline_weight[:, :, c0:c0 + cg], spline_scaler,
                                      knots[c0:c0 + cg + int(spline_order) + 1], cg - int(spline_order),
                                      int(spline_order), int(mode))
            y = part if y is None else y + part
            c0 += cg
        return y
    if torch.compiler.is_compiling():
        from . import library
        return library.kan_linear(x, base_weight, spline_weight, spline_scaler, knots, int(grid_size), int(spline_order),
                                  int(mode))[0]
    return _KANLinearFn.apply(x, base_weight, spline_weight, spline_scaler, knots, int(grid_size),
                              int(spline_order), int(mode), packed, None if out is None else _OutBuffer(out))


_PARTS_OK: dict = {}
_PARTS_ONE_LAUNCH = True      # False: per-block layers summed (a module attribute for the bitwise A/B tests, no longer an environment switch)


def parts_one_launch_widths_ok(widths: tuple, fout: int, grid_size: int, spline_order: int, mode: int) -> bool:
    """block widths ``kagnn_kan_linear_fwd_parts`` covers (asked once per shape)"""
    key = (widths, fout, grid_size, spline_order, mode)
    hit = _PARTS_OK.get(key)
    if hit is None:
        hit = bool(getattr(_lib.load(), "kagnn_kan_fwd_parts_ok")((ctypes.c_int32 * len(widths))(*widths), len(widths), sum(widths),
                                                                    fout, grid_size, spline_order, mode))
        _PARTS_OK[key] = hit
    return hit


def _parts_one_launch(parts, fout: int, grid_size: int, spline_order: int, mode: int) -> bool:
    """the column blocks qualify for ``kagnn_kan_linear_fwd_parts`` (one forward launch over all of them)"""
    p0 = parts[0]
    if not all(t.is_cuda and t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0
               and t.size(1) <= t.stride(0) <= 7680 and t.size(0) == p0.size(0) and t.device == p0.device and t.data_ptr() % 16 == 0
               for t in parts):
        return False
    return parts_one_launch_widths_ok(tuple(int(t.size(1)) for t in parts), fout, grid_size, spline_order, mode)


class _KANLinearPartsFn(Function):
    """``KANLinear`` on ``[x_0 | x_1 | ...]`` as ONE tape node: the forward is a single launch that reads every block where
    it lies (``kagnn_kan_linear_fwd_parts``: no concatenation, no partial outputs to add up); the backward runs the
    input-gradient kernel per block that needs it and the weight-gradient kernel per block, and assembles the parameter
    gradients with one concatenation each."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, base_weight, spline_weight, spline_scaler, knots, grid_size, spline_order, mode, skip_gradients, affines, norm_stats,
                *parts):
        """``affines``: ``None`` or per block ``None`` / a [2, width] tensor -- the block is read as scale * x + shift
        (``AffineRows``: a BatchNorm1d output that was never written); ``norm_stats``: ``None`` or per block ``None`` / the
        ``NormSums`` of that norm (its backward statistics can then come out of this node's input-gradient kernel)"""
        _need_cuda(base_weight, spline_weight, spline_scaler, knots, *parts)
        ctx.skip_gradients = skip_gradients
        ctx.affines = affines
        ctx.norm_stats = norm_stats
        n, fout = parts[0].size(0), spline_weight.size(0)
        widths = [int(t.size(1)) for t in parts]
        fin = sum(widths)
        bw, sw = base_weight.contiguous(), spline_weight.contiguous()
        sc = None if spline_scaler is None else spline_scaler.contiguous()
        fb, db = _sizes("kagnn_kan_pack_bytes", fin, fout, grid_size, spline_order, mode, outputs=2)
        pack_f, pack_d = _ws(fb, bw.device), _ws(db, bw.device)
        _call("kagnn_kan_pack", _ptr(bw), _ptr(sw), _ptr(sc), fin, fout, grid_size, spline_order, mode, _ptr(pack_f), _ptr(pack_d),
              _stream())
        y = torch.empty((n, fout), dtype=torch.float32, device=bw.device)
        wb = _sizes("kagnn_kan_fwd_workspace_bytes", n, fin, fout, grid_size, spline_order, mode)
        ws = _ws(wb, bw.device) if wb else None
        _call("kagnn_kan_linear_fwd_parts_affine", _ptr_array(parts), (ctypes.c_int32 * len(widths))(*widths),
              (ctypes.c_int64 * len(widths))(*[_ld(t) for t in parts]), None if affines is None else _ptr_array(list(affines)),
              len(widths), n, _ptr(knots), fin, fout, grid_size, spline_order, mode, _ptr(pack_f), _ptr(y), fout, _ptr(ws), wb, _stream())
        ctx.save_for_backward(bw, sw, sc, knots, *parts)
        ctx.dims = (widths, fout, grid_size, spline_order, mode)
        return y

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gy):
        bw, sw, sc, knots, *parts = ctx.saved_tensors
        widths, fout, G, K, mode = ctx.dims
        gy = _rows(gy)
        want_w = any(ctx.needs_input_grad[0:3])
        gxs, gbws, gsws, gscs, f0 = [], [], [], [], 0
        slices = []                          # per block: contiguous (base, spline, scaler) columns, or None when nothing needs them
        npart, w0 = len(parts), widths[0]
        if npart > 1 and all(w == w0 for w in widths) and (want_w or all(ctx.needs_input_grad[10:10 + npart])):
            # equal blocks (the node models on hidden-wide inputs): ONE strided copy per parameter tensor -- [out, P, w(, C)] ->
            # [P, out, w(, C)] -- instead of one per block and tensor (12 four-microsecond launches per read-out backward)
            out_f = bw.size(0)
            bwb = bw.view(out_f, npart, w0).permute(1, 0, 2).contiguous()
            swb = sw.view(out_f, npart, w0, sw.size(2)).permute(1, 0, 2, 3).contiguous()
            scb = None if sc is None else sc.view(out_f, npart, w0).permute(1, 0, 2).contiguous()
            slices = [(bwb[i], swb[i], None if scb is None else scb[i]) for i in range(npart)]
        else:
            for i in range(npart):
                f1 = f0 + widths[i]
                if ctx.needs_input_grad[10 + i] or want_w:
                    slices.append((bw[:, f0:f1].contiguous(), sw[:, f0:f1].contiguous(), None if sc is None else sc[:, f0:f1].contiguous()))
                else:
                    slices.append(None)
                f0 = f1
        # the input-gradient packs of all blocks that need one: ONE launch when the batch entry point covers them
        need = [i for i in range(len(parts)) if ctx.needs_input_grad[10 + i]]
        packs = kan_pack_chain([slices[i] for i in need], G, K, mode) if sc is not None and len(need) >= 2 else None
        pack_of = {} if packs is None else {i: packs[k][1] for k, i in enumerate(need)}
        for i, part in enumerate(parts):
            want_x = ctx.needs_input_grad[10 + i]
            aff = None if ctx.affines is None else ctx.affines[i]
            bwp, swp, scp = slices[i] if slices[i] is not None else (None, None, None)
            gx = None
            if want_x:
                pack_d = pack_of.get(i)
                if pack_d is None:
                    fb, db = _sizes("kagnn_kan_pack_bytes", widths[i], fout, G, K, mode, outputs=2)
                    pack_f, pack_d = _ws(fb, part.device), _ws(db, part.device)
                    _call("kagnn_kan_pack", _ptr(bwp), _ptr(swp), _ptr(scp), widths[i], fout, G, K, mode, _ptr(pack_f), _ptr(pack_d),
                          _stream())
                sk = ctx.skip_gradients[i] if ctx.skip_gradients is not None else None
                handed = sk is not None and sk.consumer    # the convolution that consumed this block adds it in its own backward
                ns = None if (ctx.norm_stats is None or handed or aff is None) else ctx.norm_stats[i]
                if (ns is not None and ns.mean is not None and _FOLD_NORM_STATS and part.data_ptr() % 16 == 0
                        and getattr(_lib.load(), "kagnn_kan_bwd_input_sums_ok")(part.size(0), widths[i], fout, G, K, mode)):
                    # this gradient goes STRAIGHT to the norm whose folded output the block is (the last convolution's): the
                    # kernel also leaves the two column sums that norm's backward starts from (no statistics pass there)
                    gx, sums = _kan_bwd_input_sums_raw(part, gy, knots, pack_d, widths[i], fout, G, K, mode, aff, ns)
                    ns.park(sums, gx)
                else:
                    gx = _kan_bwd_input_raw(part, gy, knots, pack_d, widths[i], fout, G, K, mode, x_affine=aff)
                if handed:
                    sk.park(gx)
                    gx = None
            gxs.append(gx)
            if want_w:
                gbw, gsw, gsc = _kan_bwd_weight_raw(part, gy, knots, swp, scp, widths[i], fout, G, K, mode, True, x_affine=aff)
                gbws.append(gbw); gsws.append(gsw); gscs.append(gsc)
        gbw = torch.cat(gbws, dim=1) if want_w else None
        gsw = torch.cat(gsws, dim=1) if want_w else None
        gsc = torch.cat(gscs, dim=1) if want_w and sc is not None else None
        return (gbw, gsw, gsc, None, None, None, None, None, None, None, *gxs)


def parts_affine_ok(fout: int, grid_size: int, spline_order: int, parts, lazy) -> bool:
    """can the read-out kernels apply a folded BatchNorm1d to these blocks?  (``kagnn_kan_linear_*_affine``: cubic layers of
    <= 8 coefficients and <= 64 outputs; the weight-gradient kernel needs blocks wider than 32 columns)"""
    if spline_order != 3 or grid_size + spline_order > 8 or fout > 64:
        return False
    return all(not z or int(t.size(1)) > 32 for t, z in zip(parts, lazy))


def kan_linear_parts(parts, base_weight, spline_weight, spline_scaler, knots, grid_size: int, spline_order: int,
                     mode: Optional[int] = None, skip_gradients=None) -> torch.Tensor:
    """``kan_linear`` on the column-concatenation of ``parts`` without building it: both branches of the layer are
    sums over input features, so the output is the sum of the layer restricted to each part's columns of the
    weights.  For the skip-concat read-out of the node models this saves the concatenation, and -- in the backward --
    the strided gradient slices that had to be copied contiguous for every branch.  Blocks the forward kernel can read
    in place (``kagnn_kan_fwd_parts_ok``) run as ONE launch and one tape node; anything else is the sum of per-block
    layers.  ``skip_gradients``: per block ``None`` or a ``SkipGradient`` whose consumer adds this layer's gradient of
    the block itself (only honoured by the one-node form; everywhere else the tape sums as usual)."""
    parts = list(parts)
    if sum(int(t.size(1)) for t in parts) != base_weight.size(1):
        raise AssertionError("parts do not add up to in_features")
    m = default_precision() if mode is None else int(mode)
    lazy = [isinstance(t, AffineRows) for t in parts]
    raw = [t.y if z else t for t, z in zip(parts, lazy)]
    if (_PARTS_ONE_LAUNCH and split_like(m) and knots.dim() == 1 and len(parts) > 1 and not torch.compiler.is_compiling()
            and _parts_one_launch(raw, spline_weight.size(0), int(grid_size), int(spline_order), m)
            and (not any(lazy) or parts_affine_ok(spline_weight.size(0), int(grid_size), int(spline_order), raw, lazy))):
        sk = None if skip_gradients is None or not any(k is not None and k.consumer for k in skip_gradients) else tuple(skip_gradients)
        affines = tuple(t.affine if z else None for t, z in zip(parts, lazy)) if any(lazy) else None
        stats = tuple(t.stats if z else None for t, z in zip(parts, lazy)) if any(lazy) else None
        return _KANLinearPartsFn.apply(base_weight, spline_weight, spline_scaler, knots, int(grid_size), int(spline_order), m, sk,
                                       affines, stats, *raw)
    parts = [t.materialise() if z else t for t, z in zip(parts, lazy)]        # (blocks no kernel folds: write them out)
    y, f0 = None, 0
    for part in parts:
        f1 = f0 + part.size(1)
        sc = None if spline_scaler is None else spline_scaler[:, f0:f1]
        out = kan_linear(part, base_weight[:, f0:f1], spline_weight[:, f0:f1], sc, knots, grid_size, spline_order, mode)
        y = out if y is None else y + out
        f0 = f1
    return y


def kan_bsplines(x, grid, grid_size: int, spline_order: int) -> torch.Tensor:
    """Dense B-spline bases ``[N, in, G+k]`` of ``x[N, in]`` on the per-feature knot rows ``grid[in, G+2k+1]``
    (``KANLinear.b_splines``, ekan.py:79-112).  Not differentiable (the layer never stores this tensor)."""
    _need_cuda(x, grid)
    x = _rows(x.detach())
    g = grid.detach().to(torch.float32).contiguous()
    n, fin = x.shape
    if g.shape != (fin, grid_size + 2 * spline_order + 1):
        raise AssertionError("grid must be [in_features, G+2k+1]")
    out = torch.empty((n, fin, grid_size + spline_order), dtype=torch.float32, device=x.device)
    with _device_of(x):
        _call("kagnn_kan_bsplines", _ptr(x), _ld(x), n, _ptr(g), fin, int(grid_size), int(spline_order), _ptr(out),
              _stream())
    return out


def kan_grid_refit(x, grid_old, grid_new, spline_weight, spline_scaler, grid_size: int,
                   spline_order: int) -> torch.Tensor:
    """Coefficients on ``grid_new`` that reproduce, in the least-squares sense over the rows of ``x``, the
    per-feature curves the layer had on ``grid_old`` (the refit inside ``update_grid``, ekan.py:169-177,211)."""
    _need_cuda(x, grid_old, grid_new, spline_weight, spline_scaler)
    x = _rows(x.detach())
    n, fin = x.shape
    fout = spline_weight.size(0)
    go = grid_old.detach().to(torch.float32).contiguous()
    gn = grid_new.detach().to(torch.float32).contiguous()
    sw = spline_weight.detach().contiguous()
    sc = None if spline_scaler is None else spline_scaler.detach().contiguous()
    ws = _ws(_sizes("kagnn_kan_grid_refit_workspace_bytes", n, fin, int(grid_size), int(spline_order)), x.device)
    out = torch.empty_like(sw)
    with _device_of(x):
        _call("kagnn_kan_grid_refit", _ptr(x), _ld(x), n, _ptr(go), _ptr(gn), fin, fout, int(grid_size),
              int(spline_order), _ptr(sw), _ptr(sc), _ptr(out), _ptr(ws), ws.numel(), _stream())
    return out


# ======================================================================== FastKAN layer
def _fastkan_fwd_raw(x, ln_w, ln_b, spline_w, base_w, base_b, centers, denominator, ln_eps, mode):
    """-> (y, row_stats or None)"""
    n, fin = x.shape
    fout = spline_w.size(0)
    ng = centers.numel()
    if spline_w.size(1) != fin * ng:
        raise AssertionError("spline_linear.weight must be [out, in*num_grids]")
    sw = spline_w.contiguous()
    lw = None if ln_w is None else ln_w.contiguous()
    lb = None if ln_b is None else ln_b.contiguous()
    bw = None if base_w is None else base_w.contiguous()
    bb = None if base_b is None else base_b.contiguous()
    ws = _ws(_sizes("kagnn_fastkan_fwd_workspace_bytes", n, fin, fout, ng, mode), x.device)
    stats = torch.empty((n, 2), dtype=torch.float32, device=x.device) if lw is not None else None
    y = torch.empty((n, fout), dtype=torch.float32, device=x.device)
    _call("kagnn_fastkan_fwd", _ptr(x), _ld(x), n, fin, fout, ng, _ptr(centers), float(denominator),
          _ptr(lw), _ptr(lb), float(ln_eps), _ptr(sw), _ptr(bw), _ptr(bb), _ptr(y), fout,
          _ptr(stats), mode, _ptr(ws), ws.numel(), _stream())
    return y, stats


def _fastkan_bwd_raw(x, gy, ln_w, ln_b, spline_w, base_w, centers, stats, denominator, ln_eps, mode):
    """-> (gx, g_ln_weight, g_ln_bias, g_spline_weight, g_base_weight, g_base_bias); absent ones are None"""
    n, fin = x.shape
    fout = spline_w.size(0)
    ng = centers.numel()
    dev = x.device
    sw = spline_w.contiguous()
    lw = None if ln_w is None else ln_w.contiguous()
    lb = None if ln_b is None else ln_b.contiguous()
    bw = None if base_w is None else base_w.contiguous()
    ws = _ws(_sizes("kagnn_fastkan_bwd_workspace_bytes", n, fin, fout, ng, mode), dev)
    f32 = dict(dtype=torch.float32, device=dev)
    gx = torch.empty((n, fin), **f32)
    glw = torch.empty(fin, **f32) if lw is not None else None
    glb = torch.empty(fin, **f32) if lw is not None else None
    gsw = torch.empty((fout, fin * ng), **f32)
    gbw = torch.empty((fout, fin), **f32) if bw is not None else None
    gbb = torch.empty(fout, **f32) if bw is not None else None
    _call("kagnn_fastkan_bwd", _ptr(x), _ld(x), _ptr(gy), _ld(gy), n, fin, fout, ng, _ptr(centers), float(denominator),
          _ptr(lw), _ptr(lb), float(ln_eps), _ptr(sw), _ptr(bw), _ptr(stats), _ptr(gx), fin, _ptr(glw),
          _ptr(glb), _ptr(gsw), _ptr(gbw), _ptr(gbb), mode, _ptr(ws), ws.numel(), _stream())
    return gx, glw, glb, gsw, gbw, gbb


class _FastKANFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, x, ln_w, ln_b, spline_w, base_w, base_b, centers, denominator, ln_eps, mode):
        _need_cuda(x, spline_w, centers)
        x = _rows(x)
        y, stats = _fastkan_fwd_raw(x, ln_w, ln_b, spline_w, base_w, base_b, centers, denominator, ln_eps, mode)
        ctx.save_for_backward(x, ln_w, ln_b, spline_w, base_w, centers, stats)
        ctx.meta = (float(denominator), float(ln_eps), base_b is not None, mode)
        return y

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gy):
        x, lw, lb, sw, bw, centers, stats = ctx.saved_tensors
        den, eps, has_bb, mode = ctx.meta
        gx, glw, glb, gsw, gbw, gbb = _fastkan_bwd_raw(x, _rows(gy), lw, lb, sw, bw, centers, stats, den, eps, mode)
        return gx, glw, glb, gsw, gbw, (gbb if has_bb else None), None, None, None, None


# ------------------------------------------------------------------------ feature-sharded FastKAN layer
# The local-compute half of kagnn_amd.sharded.ShardedFastKANLayer: a rank holds `w` of the row's P*w input columns.  LayerNorm
# (fastkan.py:77-78) is the only reduction over the sharded axis -- 2 floats per row each way; the collectives themselves are the
# caller's (torch.distributed), these are the library calls around them (include/kagnn_hip.h: kagnn_fastkan_row_moments ...).
def fastkan_row_moments(x: torch.Tensor) -> torch.Tensor:
    """[N, 2] = (mean, sum of squared deviations from it) of every row over the LOCAL columns"""
    _need_cuda(x)
    x = _rows(x)
    mom = torch.empty((x.size(0), 2), dtype=torch.float32, device=x.device)
    with _device_of(x):
        _call("kagnn_fastkan_row_moments", _ptr(x), _ld(x), x.size(0), x.size(1), _ptr(mom), _stream())
    return mom


def fastkan_merge_moments(gathered: torch.Tensor, width: int, ln_eps: float) -> torch.Tensor:
    """``gathered`` [P, N, 2]: every rank's ``fastkan_row_moments`` (``width`` columns each), merged per row in rank order ->
    [N, 2] = (mean, 1 / sqrt(biased variance + eps)) over all P * width columns: torch.nn.LayerNorm's statistics"""
    _need_cuda(gathered)
    gathered = gathered.contiguous()
    p, n = gathered.size(0), gathered.size(1)
    stats = torch.empty((n, 2), dtype=torch.float32, device=gathered.device)
    with _device_of(gathered):
        _call("kagnn_fastkan_merge_moments", _ptr(gathered), p, n, int(width), float(ln_eps), _ptr(stats), _stream())
    return stats


def _fastkan_shard_mode(x, spline_w, ng, mode):
    if mode is None:
        mode = default_precision()
    if split_like(mode) and (ng > 16 or not _fits32(x, spline_w.size(0))):
        mode = PREC_FP32                    # (more than 16 centres / very wide rows: the exact-fp32 kernels)
    return int(mode)


def fastkan_shard_fwd(x, stats, ln_w, ln_b, spline_w, base_w, base_b, centers, denominator: float, mode=None, out=None):
    """this rank's partial sums ``[N, out]`` of FastKANLayer.forward over its input columns, LayerNorm on the MERGED row
    statistics ``stats`` (``None`` with ``ln_w is None``); ``base_b``: on one rank only"""
    _need_cuda(x, spline_w, centers)
    x = _rows(x)
    n, fin = x.shape
    fout, ng = spline_w.size(0), centers.numel()
    if spline_w.size(1) != fin * ng:
        raise AssertionError("spline_linear.weight slice must be [out, in_local*num_grids]")
    mode = _fastkan_shard_mode(x, spline_w, ng, mode)
    sw = spline_w.contiguous()
    lw, lb = (None, None) if ln_w is None else (ln_w.contiguous(), ln_b.contiguous())
    bw = None if base_w is None else base_w.contiguous()
    bb = None if base_b is None else base_b.contiguous()
    ws = _ws(_sizes("kagnn_fastkan_fwd_workspace_bytes", n, fin, fout, ng, mode), x.device)
    y = torch.empty((n, fout), dtype=torch.float32, device=x.device) if out is None else out
    with _device_of(x):
        _call("kagnn_fastkan_shard_fwd", _ptr(x), _ld(x), n, fin, fout, ng, _ptr(centers), float(denominator), _ptr(lw), _ptr(lb),
              _ptr(stats), _ptr(sw), _ptr(bw), _ptr(bb), _ptr(y), _ld(y), mode, _ptr(ws), ws.numel(), _stream())
    return y


class FastKANShardBackward:
    """the state between ``fastkan_shard_bwd`` and ``fastkan_shard_bwd_finish``: the library workspace holding d loss / dz, the
    partially written input gradient, and the shape the second call has to repeat"""
    __slots__ = ("x", "stats", "lw", "lb", "gx", "ws", "dims", "mode", "wgrads")


def fastkan_shard_bwd(x, gy, stats, ln_w, ln_b, spline_w, base_w, centers, denominator: float, mode=None, want_bias=False):
    """everything of the layer's backward but the LayerNorm backward, from the gathered gradient ``gy`` [N, out]:
    ``(state, row_sums [N, 2] or None, g_spline_weight, g_base_weight, g_base_bias)``.  ``row_sums`` = (sum_f gz*gamma,
    sum_f gz*gamma*zhat) over the local columns: to be summed over the ranks, then ``fastkan_shard_bwd_finish``."""
    st, sums = fastkan_shard_bwd_halves(x, gy, stats, ln_w, ln_b, spline_w, base_w, centers, denominator, mode, part="both",
                                        want_bias=want_bias)
    return (st, sums) + st.wgrads


def fastkan_shard_bwd_halves(x, gy, stats, ln_w, ln_b, spline_w, base_w, centers, denominator: float, mode=None, part="input",
                             state=None, want_bias=False):
    """``kagnn_fastkan_shard_bwd`` in its two halves: ``part="input"`` -> ``(state, row_sums)`` (gx's base-branch part and
    d loss / dz are in ``state``); ``part="weight", state=...`` -> ``(g_spline_weight, g_base_weight, g_base_bias)`` -- the
    caller's all-reduce of ``row_sums`` runs between / beside them.  ``part="both"``: one call, ``state.wgrads`` holds the three."""
    _need_cuda(x, gy, spline_w, centers)
    x, gy = _rows(x), _rows(gy)
    n, fin = x.shape
    fout, ng = spline_w.size(0), centers.numel()
    mode = _fastkan_shard_mode(x, spline_w, ng, mode)
    dev = x.device
    sw = spline_w.contiguous()
    lw, lb = (None, None) if ln_w is None else (ln_w.contiguous(), ln_b.contiguous())
    bw = None if base_w is None else base_w.contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    st = state
    if st is None:
        st = FastKANShardBackward()
        st.x, st.stats, st.lw, st.lb, st.mode, st.dims = x, stats, lw, lb, mode, (n, fin, fout, ng)
        st.ws = _ws(_sizes("kagnn_fastkan_bwd_workspace_bytes", n, fin, fout, ng, mode), dev)
        st.gx = torch.empty((n, fin), **f32)
        st.wgrads = None
    parts = {"input": 1, "weight": 2, "both": 3}[part]
    sums = gsw = gbw = gbb = None
    if parts & 1:
        sums = torch.empty((n, 2), **f32) if lw is not None else None
    if parts & 2:
        gsw = torch.empty((fout, fin * ng), **f32)
        gbw = torch.empty((fout, fin), **f32) if bw is not None else None
        gbb = torch.empty(fout, **f32) if (want_bias and bw is not None) else None
    with _device_of(x):
        _call("kagnn_fastkan_shard_bwd", _ptr(x), _ld(x), _ptr(gy), _ld(gy), n, fin, fout, ng, _ptr(centers), float(denominator),
              _ptr(lw), _ptr(lb), _ptr(sw), _ptr(bw), _ptr(stats), _ptr(st.gx), fin, _ptr(sums), _ptr(gsw), _ptr(gbw), _ptr(gbb),
              parts, mode, _ptr(st.ws), st.ws.numel(), _stream())
    if parts == 2:
        return gsw, gbw, gbb
    st.wgrads = (gsw, gbw, gbb)
    return st, sums


def fastkan_shard_bwd_finish(st: FastKANShardBackward, row_sums: Optional[torch.Tensor], width_total: int):
    """-> ``(gx, g_ln_weight, g_ln_bias)`` with ``row_sums`` summed over the ranks (``None``: the layer has no LayerNorm)"""
    if st.lw is None:
        return st.gx, None, None
    n, fin, fout, ng = st.dims
    glw = torch.empty(fin, dtype=torch.float32, device=st.gx.device)
    glb = torch.empty_like(glw)
    with _device_of(st.gx):
        _call("kagnn_fastkan_shard_bwd_finish", _ptr(st.x), _ld(st.x), n, fin, int(width_total), fout, ng, _ptr(st.lw), _ptr(st.lb),
              _ptr(st.stats), _ptr(row_sums), _ptr(st.gx), fin, _ptr(glw), _ptr(glb), st.mode, _ptr(st.ws), st.ws.numel(), _stream())
    return st.gx, glw, glb


# ======================================================================== harness loss
class _SoftmaxXentFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, logits, labels, mask, pre_softmax):
        _need_cuda(logits, labels, mask)
        z = _rows(logits)
        n, c = z.shape
        y = labels.to(torch.int64).contiguous()
        m = None if mask is None else mask.contiguous()
        ws = _ws(_sizes("kagnn_softmax_xent_workspace_bytes", n), z.device)
        out = torch.empty(2, dtype=torch.float32, device=z.device)          # (loss, row count)
        stats = torch.empty((n, 3), dtype=torch.float32, device=z.device)
        _call("kagnn_softmax_xent_fwd", _ptr(z), _ld(z), n, c, _ptr(y), _ptr(m), int(pre_softmax), _ptr(out),
              _ptr(stats), ctypes.c_void_p(out.data_ptr() + 4), _ptr(ws), ws.numel(), _stream())
        ctx.save_for_backward(z, y, m, stats, out)
        ctx.pre = int(pre_softmax)
        return out[0]

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gloss):
        z, y, m, stats, out = ctx.saved_tensors
        n, c = z.shape
        g = gloss.to(torch.float32).contiguous()
        gz = torch.empty((n, c), dtype=torch.float32, device=z.device)
        _call("kagnn_softmax_xent_bwd", _ptr(z), _ld(z), n, c, _ptr(y), _ptr(m), ctx.pre, _ptr(stats),
              ctypes.c_void_p(out.data_ptr() + 4), _ptr(g), _ptr(gz), c, _stream())
        return gz, None, None, None


def softmax_cross_entropy(logits, labels, mask=None, pre_softmax: bool = False) -> torch.Tensor:
    """``CrossEntropyLoss()(f(logits)[mask], labels[mask])`` (mean over the masked rows) in one kernel each way, with
    ``f = softmax(dim=1)`` when ``pre_softmax`` (the reference harness, ``time_model.py:43-45``) and the identity
    otherwise.  ``mask``: bool ``[N]`` or ``None``; an int64 index tensor of distinct rows is accepted too.  No
    device-to-host sync (boolean indexing has one), so a whole epoch can be captured in a HIP graph."""
    if mask is not None and mask.dtype != torch.bool:
        picked = torch.zeros(logits.size(0), dtype=torch.bool, device=logits.device)
        picked[mask] = True
        mask = picked
    return _SoftmaxXentFn.apply(logits, labels, mask, bool(pre_softmax))


class _L1LossFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, pred, target):
        _need_cuda(pred, target)
        p = pred.to(torch.float32).contiguous()
        t = target.to(torch.float32).contiguous()
        out = torch.empty(1, dtype=torch.float32, device=p.device)
        _call("kagnn_l1_loss_fwd", _ptr(p), _ptr(t), p.numel(), _ptr(out), _stream())
        ctx.save_for_backward(p, t)
        return out[0]

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gloss):
        p, t = ctx.saved_tensors
        g = gloss.to(torch.float32).contiguous()
        gp = torch.empty_like(p)
        _call("kagnn_l1_loss_bwd", _ptr(p), _ptr(t), p.numel(), _ptr(g), _ptr(gp), _stream())
        return gp, None


def l1_loss(pred, target) -> torch.Tensor:
    """``torch.nn.L1Loss()(pred, target)`` (mean absolute error; the loss of the reference's graph-regression scripts,
    ``graph_regression/optuna_zinc.py:58``) as one kernel each way; same shapes required (no broadcasting -- L1Loss warns about it
    and the scripts squeeze the prediction for that reason); the target gets no gradient."""
    if pred.shape != target.shape:
        raise ValueError(f"l1_loss: prediction {tuple(pred.shape)} and target {tuple(target.shape)} must have the same shape")
    return _L1LossFn.apply(pred, target)


# ======================================================================== GAT attention aggregation
class _GatFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, xh, att_src, att_dst, bias, g, heads, channels):
        _need_cuda(xh, att_src, att_dst)
        xh = _rows(xh)
        n = xh.size(0)
        dev = xh.device
        a_s = att_src.reshape(heads, channels).contiguous()
        a_d = att_dst.reshape(heads, channels).contiguous()
        b = None if bias is None else bias.contiguous()
        ls = torch.empty((n, heads), dtype=torch.float32, device=dev)
        ld_ = torch.empty((n, heads), dtype=torch.float32, device=dev)
        _call("kagnn_gat_logits", _ptr(xh), _ld(xh), n, heads, channels, _ptr(a_s), _ptr(a_d), _ptr(ls), _ptr(ld_), _stream())
        out = torch.empty((n, heads * channels), dtype=torch.float32, device=dev)
        m = torch.empty((n, heads), dtype=torch.float32, device=dev)
        z = torch.empty((n, heads), dtype=torch.float32, device=dev)
        _call("kagnn_gat_fwd", _ptr(xh), _ld(xh), _ptr(ls), _ptr(ld_), _ptr(g.rowptr), _ptr(g.col), n, heads, channels,
              _ptr(b), _ptr(out), heads * channels, _ptr(m), _ptr(z), _ptr(g.hub_seg) if g.num_hub_seg else None,
              g.num_hub_seg, g.hub_threshold, _stream())
        ctx.save_for_backward(xh, a_s, a_d, b, ls, ld_, m, z, out)
        ctx.g, ctx.hc = g, (heads, channels)
        ctx.att_shape = att_src.shape
        return out

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gout):
        xh, a_s, a_d, b, ls, ld_, m, z, out = ctx.saved_tensors
        g = ctx.g
        heads, channels = ctx.hc
        gout = _rows(gout)
        n, dev = xh.size(0), xh.device
        f32 = dict(dtype=torch.float32, device=dev)
        gpre = torch.empty(max(g.num_edges, 1) * heads, **f32)
        gself = torch.empty((n, heads), **f32)
        gd = torch.empty((n, heads), **f32)
        gs = torch.empty((n, heads), **f32)
        gx = torch.empty((n, heads * channels), **f32)
        _call("kagnn_gat_bwd", _ptr(xh), _ld(xh), _ptr(gout), _ld(gout), _ptr(out), heads * channels, _ptr(b), _ptr(ls),
              _ptr(ld_), _ptr(m), _ptr(z), _ptr(g.rowptr), _ptr(g.col), _ptr(g.perm), _ptr(g.rowptr_t), _ptr(g.col_t),
              _ptr(g.perm_t), _ptr(a_s), _ptr(a_d), n, heads, channels, _ptr(gpre), _ptr(gself), _ptr(gd), _ptr(gs),
              _ptr(gx), heads * channels, _ptr(g.hub_seg) if g.num_hub_seg else None, g.num_hub_seg, g.hub_threshold,
              _stream())
        if heads * channels <= 1024:                      # [H, C] contractions over the nodes: one pass over xh
            ws = _ws(_sizes("kagnn_gat_att_grad_workspace_bytes", n, heads, channels), dev)
            g_att = torch.empty((2, heads * channels), **f32)
            _call("kagnn_gat_att_grad", _ptr(xh), _ld(xh), _ptr(gs), _ptr(gd), n, heads, channels, _ptr(g_att),
                  ctypes.c_void_p(g_att.data_ptr() + 4 * heads * channels), _ptr(ws), ws.numel(), _stream())
            g_att_src, g_att_dst = g_att[0].reshape(ctx.att_shape), g_att[1].reshape(ctx.att_shape)
        else:
            x3 = xh.reshape(n, heads, channels) if xh.is_contiguous() else xh.contiguous().view(n, heads, channels)
            g_att_src = torch.einsum("nh,nhc->hc", gs, x3).reshape(ctx.att_shape)
            g_att_dst = torch.einsum("nh,nhc->hc", gd, x3).reshape(ctx.att_shape)
        g_bias = gout.sum(0) if b is not None else None
        return gx, g_att_src, g_att_dst, g_bias, None, None, None


def gat_aggregate(xh, att_src, att_dst, bias, g: GraphIndex, heads: int, channels: int) -> torch.Tensor:
    """GATConv message passing on ``xh = lin(x)`` (heads concatenated): see csrc/gat.hip."""
    return _GatFn.apply(xh, att_src, att_dst, bias, g, int(heads), int(channels))


# ======================================================================== BatchNorm1d (conv epilogue)
def _batchnorm_fwd_raw(x, weight, bias, running_mean, running_var, training, momentum, eps, moments=None,
                       dropout_p=0.0, dropout_seed=0):
    """-> (y, save_mean, save_rstd); running statistics are updated in place by the kernel.  ``moments``: the [2, F]
    column moments of x from its producer (no statistics pass); ``dropout_p`` > 0: y = dropout(bn(x)) in the same pass"""
    n, f = x.shape
    ws = _ws(_sizes("kagnn_batchnorm_workspace_bytes", n, f), x.device)
    y = torch.empty((n, f), dtype=torch.float32, device=x.device)
    mean = torch.empty(f, dtype=torch.float32, device=x.device)
    rstd = torch.empty(f, dtype=torch.float32, device=x.device)
    w = None if weight is None else weight.contiguous()
    b = None if bias is None else bias.contiguous()
    use_mom = moments is not None and training
    _call("kagnn_batchnorm_fwd", _ptr(x), _ld(x), n, f, _ptr(w), _ptr(b), _ptr(running_mean), _ptr(running_var),
          float(momentum), float(eps), int(bool(training)), _ptr(moments[0]) if use_mom else None,
          _ptr(moments[1]) if use_mom else None, float(dropout_p), int(dropout_seed), _ptr(y), f, _ptr(mean),
          _ptr(rstd), _ptr(ws), ws.numel(), _stream())
    return y, mean, rstd


def _batchnorm_bwd_raw(x, gy, weight, mean, rstd, training, want_gx, want_bias, dropout_p=0.0, dropout_seed=0):
    n, f = x.shape
    w = None if weight is None else weight.contiguous()
    ws = _ws(_sizes("kagnn_batchnorm_workspace_bytes", n, f), x.device)
    gx = torch.empty((n, f), dtype=torch.float32, device=x.device) if want_gx else None
    gw = torch.empty(f, dtype=torch.float32, device=x.device) if w is not None else None
    gb = torch.empty(f, dtype=torch.float32, device=x.device) if want_bias else None
    _call("kagnn_batchnorm_bwd", _ptr(x), _ld(x), _ptr(gy), _ld(gy), n, f, _ptr(w), _ptr(mean), _ptr(rstd),
          int(training), float(dropout_p), int(dropout_seed), _ptr(gx), f, _ptr(gw), _ptr(gb), _ptr(ws), ws.numel(),
          _stream())
    return gx, gw, gb


class _BatchNormFn(Function):
    @staticmethod
    @_on_operand_device
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, moments=None, dropout_p=0.0,
                dropout_seed=0):
        _need_cuda(x, weight, bias, running_mean, running_var, moments)
        x = _rows(x)
        y, mean, rstd = _batchnorm_fwd_raw(x, weight, bias, running_mean, running_var, training, momentum, eps, moments,
                                           dropout_p, dropout_seed)
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.training = bool(training)
        ctx.dropout = (float(dropout_p), int(dropout_seed))
        ctx.has_bias = bias is not None
        return y                      # running statistics are grad-free buffers, updated in place by the kernel

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gy):
        x, w, mean, rstd = ctx.saved_tensors
        gx, gw, gb = _batchnorm_bwd_raw(x, _rows(gy), w, mean, rstd, ctx.training, ctx.needs_input_grad[0], ctx.has_bias,
                                        *ctx.dropout)
        return gx, gw, gb, None, None, None, None, None, None, None, None


def dropout_seed() -> int:
    """a fresh 63-bit seed for the fused dropout, drawn from torch's CPU default generator (``torch.manual_seed``
    reproduces it; no device work)"""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF


def batch_norm(x, weight, bias, running_mean, running_var, training: bool, momentum: float, eps: float,
               moments: Optional[torch.Tensor] = None, dropout_p: float = 0.0, seed: Optional[int] = None):
    """torch.nn.functional.batch_norm on [N, F] rows (reference ``models.py:195-202`` epilogue).

    The fused epilogue of SURVEY.md 8(f) rank 1: ``moments`` = the [2, F] column moments of ``x`` left by the kernel
    that produced it (``gin_kan_layer(..., moments=True)``) replaces the statistics pass in training; ``dropout_p`` > 0
    (training) returns ``dropout(batch_norm(x), p)`` from the same kernel, the mask being a hash of (seed, row, column)
    that the backward regenerates -- same distribution as ``F.dropout``, not torch's random stream."""
    if x.size(0) == 0:
        return x.new_empty(x.shape)
    if not 0.0 <= dropout_p <= 1.0:
        raise ValueError(f"dropout probability has to be between 0 and 1, but got {dropout_p}")
    p = float(dropout_p) if training else 0.0
    if torch.compiler.is_compiling():
        from . import library
        y = library.batch_norm_traced(x, weight, bias, running_mean, running_var, bool(training), float(momentum), float(eps))
        return torch.nn.functional.dropout(y, p, True) if p > 0.0 else y
    if p > 0.0 and seed is None:
        seed = dropout_seed()
    return _BatchNormFn.apply(x, weight, bias, running_mean, running_var, training, momentum, eps, moments, p,
                              seed if p > 0.0 else 0)


class _ConcatColumnsFn(Function):
    """torch.cat(dim=1) whose backward hands every branch a CONTIGUOUS gradient (stock cat backward returns
    strided column slices, which sends BatchNorm1d's backward down a ~15x slower non-contiguous kernel)."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, *parts):
        ctx.widths = [p.size(1) for p in parts]
        return torch.cat(parts, dim=1)

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, g):
        return tuple(c.contiguous() for c in torch.split(g, ctx.widths, dim=1))


def concat_columns(parts) -> torch.Tensor:
    """Skip-concatenation of the node models (reference ``models.py:200-201``: ``torch.cat(outs, dim=1)``)."""
    return _ConcatColumnsFn.apply(*parts)


def _fits32(x, width) -> bool:
    """The split kernels address activations with 32-bit byte offsets inside per-tile buffer windows: any number of
    rows, row strides up to 7680 floats (wider inputs take the exact-fp32 kernels)."""
    return max(x.stride(0) if x.dim() == 2 else 0, x.size(-1), width) <= 7680


def fastkan_layer(x, ln_weight, ln_bias, spline_weight, base_weight, base_bias, centers,
                  denominator: float, ln_eps: float = 1e-5, mode: Optional[int] = None) -> torch.Tensor:
    """FastKANLayer.forward (fastkan.py:76-85) on 2-D input."""
    if mode is None:
        mode = default_precision()
    if split_like(mode) and not _fits32(x, spline_weight.size(0)):
        mode = PREC_FP32
    ng = centers.numel()
    if split_like(mode) and ng > 16:
        # more than 16 centres: the layer is a sum over groups of centres (each <= 16, split-precision kernels);
        # every group normalises x the same way, the base branch rides with the first
        fout, fin = spline_weight.size(0), spline_weight.size(1) // ng
        w3 = spline_weight.view(fout, fin, ng)
        groups = -(-ng // 16)
        size, extra = divmod(ng, groups)
        y, c0 = None, 0
        for g in range(groups):
            cg = size + (1 if g < extra else 0)
            part = _FastKANFn.apply(x, ln_weight, ln_bias, w3[:, :, c0:c0 + cg].reshape(fout, fin * cg),
                                    base_weight if g == 0 else None, base_bias if g == 0 else None,
                                    centers[c0:c0 + cg], float(denominator), float(ln_eps), int(mode))
            y = part if y is None else y + part
            c0 += cg
        return y
    if torch.compiler.is_compiling():
        from . import library
        return library.fastkan_layer(x, ln_weight, ln_bias, spline_weight, base_weight, base_bias, centers,
                                     float(denominator), float(ln_eps), int(mode))[0]
    return _FastKANFn.apply(x, ln_weight, ln_bias, spline_weight, base_weight, base_bias, centers,
                            float(denominator), float(ln_eps), int(mode))
