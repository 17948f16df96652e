"""ctypes host of libkagnn_rccl.so (C ABI: include/kagnn_rccl.h): the feature-sharded KANLinear whose exchange step --
rank-major staging, ``ncclReduceScatter`` (forward) / ``ncclAllGather`` (backward), row chunks overlapped with the KAN
kernels -- is library code on an ``ncclComm_t``, not a sequence of ``torch.distributed`` calls.  SURVEY.md 8(b) lists it
("the sharded variants taking an ncclComm_t / process-group"); ``ShardedGIKANLayer(comm="rccl_c")`` uses it.

The reference has no multi-GPU code.  What every rank computes is ``KANLinear.forward`` on its
input-feature slice (``node_classification_clean/ekan.py:154-162``) and its backward.  No fallback: a missing library or
a failing call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int32, c_int64, c_size_t, c_void_p, POINTER
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KAGNN_RCCL_LIB") or os.path.join(_HERE, "lib", "libkagnn_rccl.so")

_P = c_void_p
_SIGNATURES = {
    "kagnn_rccl_version": (c_int32, []),
    "kagnn_rccl_last_error": (ctypes.c_char_p, []),
    "kagnn_rccl_unique_id": (c_int32, [_P]),
    "kagnn_rccl_comm_init": (c_int32, [_P, c_int32, c_int32, POINTER(c_void_p)]),
    "kagnn_rccl_comm_destroy": (c_int32, [_P]),
    "kagnn_sharded_kan_linear_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                           POINTER(c_size_t), POINTER(c_size_t)]),
    "kagnn_sharded_kan_linear_fwd": (c_int32, [_P, c_int64, c_int64, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                               _P, c_int32, c_int32, c_int32, _P, c_size_t, _P, _P]),
    "kagnn_sharded_kan_linear_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P,
                                               _P, _P, _P, c_int64, _P, _P, _P,
                                               _P, c_int32, c_int32, c_int32, _P, c_size_t, _P, _P]),
}
EXPORTED = tuple(_SIGNATURES)
_rccl = None


def load() -> ctypes.CDLL:
    """Load libkagnn_rccl.so (once; it pulls in libkagnn_hip.so and librccl).  RuntimeError when it has not been built."""
    global _rccl
    if _rccl is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m kagnn_amd._build` (needs hipcc and RCCL). "
                               "kagnn_amd has no CPU or eager-torch fallback.")
        _lib.load()                                   # (same libkagnn_hip.so instance the rest of the package calls)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)                   # AttributeError here = header / library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _rccl = lib
    return _rccl


def _call(name: str, *args) -> None:
    rc = getattr(load(), name)(*args)
    if rc != 0:
        msg = load().kagnn_rccl_last_error()
        raise RuntimeError(f"libkagnn_rccl {name} failed (code {rc}): {msg.decode() if msg else '?'}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class Communicator:
    """An ``ncclComm_t`` of this process's own (``kagnn_rccl_comm_init`` on the current device).  ``from_group`` makes one
    that spans a ``torch.distributed`` group: rank 0 draws the unique id, the 128 bytes travel through the group's object
    broadcast (torch's process group does not hand out its own ``ncclComm_t``)."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: torch.device):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of an ncclUniqueId")
        self.world, self.rank, self.device = int(world), int(rank), torch.device(device)
        self._side = None
        out = c_void_p()
        buf = ctypes.create_string_buffer(unique_id, 128)
        with torch.cuda.device(self.device):
            _call("kagnn_rccl_comm_init", ctypes.cast(buf, c_void_p), self.world, self.rank, ctypes.byref(out))
        self.handle = out.value

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _call("kagnn_rccl_unique_id", ctypes.cast(buf, c_void_p))
        return buf.raw

    @classmethod
    def from_group(cls, group, device) -> "Communicator":
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], world, rank, device)

    @property
    def side(self) -> torch.cuda.Stream:
        """ONE side stream per communicator: RCCL orders the collectives of a communicator by issue order anyway"""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def close(self) -> None:
        if getattr(self, "handle", None):
            _call("kagnn_rccl_comm_destroy", self.handle)
            self.handle = None

    def __del__(self):                                # (best effort; interpreter shutdown may already have unloaded things)
        try:
            self.close()
        except Exception:                             # noqa: BLE001
            pass


def default_row_chunks(n: int) -> int:
    return 4 if n >= 262144 else 1


class _ShardedKANLinearFn(Function):
    """``y_shard = reduce_scatter(KANLinear_slice(x_slice))`` and its backward as ONE library call each way."""

    @staticmethod
    def forward(ctx, x, base_weight, spline_weight, spline_scaler, knots, grid_size, spline_order, mode, comm, row_chunks, packed):
        from . import ops
        ops._need_cuda(x, base_weight, spline_weight, spline_scaler, knots)
        x = ops._rows(x)
        n, fin = x.shape
        fout = spline_weight.size(0)
        if fout % comm.world:
            raise ValueError("out_features must be divisible by the world size")
        bw = None if base_weight is None else base_weight.contiguous()
        sw = spline_weight.contiguous()
        sc = None if spline_scaler is None else spline_scaler.contiguous()
        key = ops._weights_key(base_weight, spline_weight, spline_scaler)
        with torch.cuda.device(x.device):
            if packed is not None and packed[2] == key:
                pack_f, pack_d = packed[0], packed[1]
            else:
                fb, db = ops._sizes("kagnn_kan_pack_bytes", fin, fout, grid_size, spline_order, mode, outputs=2)
                pack_f, pack_d = ops._ws(fb, x.device), ops._ws(db, x.device)
                ops._call("kagnn_kan_pack", _ptr(bw), _ptr(sw), _ptr(sc), fin, fout, grid_size, spline_order, mode,
                          _ptr(pack_f), _ptr(pack_d), ops._stream())
            fwd_b, bwd_b = c_size_t(), c_size_t()
            _call("kagnn_sharded_kan_linear_workspace_bytes", n, fin, fout, grid_size, spline_order, mode, comm.world, row_chunks,
                  ctypes.byref(fwd_b), ctypes.byref(bwd_b))
            ws = ops._ws(fwd_b.value, x.device)
            y = torch.empty((n, fout // comm.world), dtype=torch.float32, device=x.device)
            side = comm.side
            _call("kagnn_sharded_kan_linear_fwd", _ptr(x), ops._ld(x), n, _ptr(knots), fin, fout, grid_size, spline_order, mode,
                  _ptr(pack_f), _ptr(y), comm.handle, comm.world, comm.rank, row_chunks, _ptr(ws), ws.numel(),
                  ops._stream(), side.cuda_stream)
            ws.record_stream(side)                    # the side stream's collectives read the staging blocks
        ctx.save_for_backward(x, sw, sc, knots, pack_d)
        ctx.dims = (fin, fout, grid_size, spline_order, mode, row_chunks, bwd_b.value)
        ctx.comm, ctx.has_base = comm, bw is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        from . import ops
        x, sw, sc, knots, pack_d = ctx.saved_tensors
        fin, fout, G, K, mode, row_chunks, bwd_bytes = ctx.dims
        comm = ctx.comm
        n = x.size(0)
        gy = gy.contiguous()
        with torch.cuda.device(x.device):
            ws = ops._ws(bwd_bytes, x.device)
            gx = torch.empty((n, fin), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[0] else None
            gbw = torch.empty((fout, fin), dtype=torch.float32, device=x.device) if ctx.has_base else None
            gsw = torch.empty((fout, fin, G + K), dtype=torch.float32, device=x.device)
            gsc = None if sc is None else torch.empty((fout, fin), dtype=torch.float32, device=x.device)
            side = comm.side
            _call("kagnn_sharded_kan_linear_bwd", _ptr(x), ops._ld(x), _ptr(gy), n, _ptr(knots), fin, fout, G, K, mode,
                  _ptr(pack_d), _ptr(sw), _ptr(sc), _ptr(gx), fin, _ptr(gbw), _ptr(gsw), _ptr(gsc),
                  comm.handle, comm.world, comm.rank, row_chunks, _ptr(ws), ws.numel(), ops._stream(), side.cuda_stream)
            ws.record_stream(side)
            gy.record_stream(side)                    # the all-gathers read it on the side stream
        return gx, gbw, gsw, gsc, None, None, None, None, None, None, None


def sharded_kan_linear(x_slice, base_weight, spline_weight, spline_scaler, knots, grid_size: int, spline_order: int,
                       mode: Optional[int], comm: Communicator, row_chunks: Optional[int] = None, packed=None) -> torch.Tensor:
    """This rank's ``[N, out/world]`` column block of the KANLinear whose input features are split over the ranks of ``comm``."""
    from . import ops
    mode = ops.default_precision() if mode is None else mode
    rc = default_row_chunks(x_slice.size(0)) if row_chunks is None else int(row_chunks)
    return _ShardedKANLinearFn.apply(x_slice, base_weight, spline_weight, spline_scaler, knots, int(grid_size), int(spline_order),
                                     int(mode), comm, max(1, rc), packed)
