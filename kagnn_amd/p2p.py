"""Peer-mapped exchange buffers for the direct peer-to-peer steps of the feature-sharded layer.

One process per GPU (the launch contract of bench.py); every rank allocates its exchange buffer with the caching allocator,
exports it as an IPC handle (hipIpcGetMemHandle underneath ``UntypedStorage._share_cuda_``; the pool's driver speaks dmabuf IPC:
``HSA_ENABLE_IPC_MODE_LEGACY=0``), the handles travel through ``torch.distributed.all_gather_object`` (control plane only -- any
backend) and every rank maps its peers' buffers (hipIpcOpenMemHandle underneath ``UntypedStorage._new_shared_cuda``).  The data
path is then ``kagnn_p2p_reduce_scatter`` / ``kagnn_p2p_all_gather`` (csrc/p2p.hip): kernels that READ the peers' buffers over
xGMI.  The reference has no multi-GPU code."""
from __future__ import annotations

import ctypes
from typing import List

import torch
import torch.distributed as dist

from . import ops


class PeerBuffers:
    """``numel`` fp32 elements per rank, mapped on every rank: ``local`` is this rank's buffer, ``views[p]`` rank p's."""

    def __init__(self, numel: int, device: torch.device, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        # a storage of its own (not a slice of a cached block shared with other tensors: the whole allocation is exported)
        self.local = torch.empty(max(int(numel), 4), dtype=torch.float32, device=self.device)
        handle = self.local.untyped_storage()._share_cuda_()
        handles = [None] * self.world
        dist.all_gather_object(handles, handle, group=group)
        self.views: List[torch.Tensor] = []
        self._keep = []
        for p, h in enumerate(handles):
            if p == self.rank:
                self.views.append(self.local)
                continue
            st = torch.UntypedStorage._new_shared_cuda(*h)
            self._keep.append(st)
            t = torch.empty(0, dtype=torch.float32, device=st.device).set_(st, 0, (self.local.numel(),))
            if t.device != self.device:                       # first touch enables peer access from this device to the peer's
                torch.empty(1, dtype=torch.float32, device=self.device).copy_(t[:1])
            self.views.append(t)
        self._ptrs = (ctypes.c_void_p * self.world)(*[v.data_ptr() for v in self.views])
        dist.barrier(group=group)                             # nobody proceeds (and frees) before everybody has mapped

    def ptr_array(self):
        return self._ptrs


_BARRIER_TOKEN: dict = {}


def rank_barrier(group=None) -> None:
    """every rank's device work enqueued so far on the CURRENT stream is complete on all ranks after this (stream-ordered with
    RCCL: a one-element all-reduce on the current stream -- the token tensor is allocated once per device; host-side with any
    other backend)"""
    if dist.get_backend(group) == "nccl":
        dev = torch.cuda.current_device()
        t = _BARRIER_TOKEN.get(dev)
        if t is None:
            t = _BARRIER_TOKEN[dev] = torch.zeros(1, device=torch.device("cuda", dev))
        dist.all_reduce(t, group=group)
    else:
        torch.cuda.current_stream().synchronize()
        dist.barrier(group=group)


def reduce_scatter(bufs: PeerBuffers, n: int, out: int, rows=None, y: torch.Tensor = None) -> torch.Tensor:
    """rank r's column block of the sum over ranks of the ``[n, out]`` partial matrices sitting in ``bufs``; ``rows=(r0, r1)``
    pulls that row range only (into ``y[r0:r1]`` when ``y`` is given: the row-chunked exchange)"""
    w = out // bufs.world
    r0, r1 = (0, n) if rows is None else rows
    if y is None:
        y = torch.empty((n, w), dtype=torch.float32, device=bufs.device)
    if r1 > r0:
        with ops._device_of(y):
            ops._call("kagnn_p2p_reduce_scatter", _shifted(bufs, [r0 * out] * bufs.world), bufs.world, bufs.rank, r1 - r0, out, out,
                      y.data_ptr() + 4 * r0 * w, w, ops._stream())
    return y


def all_gather(bufs: PeerBuffers, n: int, w: int, rows=None, g: torch.Tensor = None) -> torch.Tensor:
    """``[n, world*w]``: the ``[n, w]`` shards sitting in ``bufs``, side by side in rank order; ``rows=(r0, r1)`` pulls that row
    range only (into ``g[r0:r1]`` when ``g`` is given)"""
    r0, r1 = (0, n) if rows is None else rows
    if g is None:
        g = torch.empty((n, w * bufs.world), dtype=torch.float32, device=bufs.device)
    if r1 > r0:
        with ops._device_of(g):
            ops._call("kagnn_p2p_all_gather", _shifted(bufs, [r0 * w] * bufs.world), bufs.world, r1 - r0, w, w,
                      g.data_ptr() + 4 * r0 * w * bufs.world, w * bufs.world, ops._stream())
    return g


def _shifted(bufs: PeerBuffers, offsets_elems):
    """host array of the peers' buffer pointers advanced by per-peer element offsets"""
    return (ctypes.c_void_p * len(offsets_elems))(*[bufs.views[p].data_ptr() + 4 * int(o) for p, o in enumerate(offsets_elems)])


def cols_to_rows(bufs: PeerBuffers, n: int, w: int, splits) -> torch.Tensor:
    """all-to-all, column shards -> row shards: every rank's ``[n, w]`` column shard sits in ``bufs``; returns this rank's rows of
    ALL columns ``[n_me, world*w]`` (rank order) -- one pull kernel (``kagnn_p2p_all_gather`` on row-shifted peer pointers)"""
    r0 = sum(splits[:bufs.rank])
    n_me = splits[bufs.rank]
    out = torch.empty((n_me, w * bufs.world), dtype=torch.float32, device=bufs.device)
    if n_me:
        with ops._device_of(out):
            ops._call("kagnn_p2p_all_gather", _shifted(bufs, [r0 * w] * bufs.world), bufs.world, n_me, w, w, ops._ptr(out),
                      w * bufs.world, ops._stream())
    return out


def rows_to_cols(bufs: PeerBuffers, n: int, w: int, splits) -> torch.Tensor:
    """all-to-all, row shards -> column shards: rank p's ``[n_p, world*w]`` row block sits in ``bufs``; returns this rank's columns
    of ALL rows ``[n, w]`` -- one strided pull per peer (``kagnn_p2p_all_gather`` with world = 1)"""
    out = torch.empty((n, w), dtype=torch.float32, device=bufs.device)
    f = w * bufs.world
    r0 = 0
    with ops._device_of(out):
        for p, n_p in enumerate(splits):
            if n_p:
                src = (ctypes.c_void_p * 1)(bufs.views[p].data_ptr() + 4 * bufs.rank * w)
                ops._call("kagnn_p2p_all_gather", src, 1, n_p, w, f, out.data_ptr() + 4 * r0 * w, w, ops._stream())
            r0 += n_p
    return out
