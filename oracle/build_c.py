"""Build the oracle's C restatement (oracle/kan_ref.c) with gcc -> oracle/_build/libkagnn_ref.so and
bind it with ctypes.  Test infrastructure only (see the header of kan_ref.c)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "kan_ref.c")
OUT = os.path.join(HERE, "_build", "libkagnn_ref.so")


def build(verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O2", "-std=c11", "-shared", "-fPIC", "-o", OUT, SRC, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def bspline_bases(x, grid, G, k):
    x, grid = _f32(x), _f32(grid)
    n, fin = x.shape
    out = np.empty((n, fin, G + k), dtype=np.float64)
    lib().kagnn_ref_bspline_bases(_p(x), ctypes.c_int64(n), fin, _p(grid), G, k, _p(out))
    return out


def kan_linear_fwd(x, p, G, k):
    x = _f32(x)
    n, fin = x.shape
    bw, sw, sc, grid = (_f32(p[q]) for q in ("base_weight", "spline_weight", "spline_scaler", "grid"))
    fout = bw.shape[0]
    y = np.empty((n, fout), dtype=np.float64)
    lib().kagnn_ref_kan_linear_fwd(_p(x), ctypes.c_int64(n), fin, fout, G, k, _p(grid), _p(bw), _p(sw), _p(sc), _p(y))
    return y


def kan_linear_bwd(x, gy, p, G, k):
    x, gy = _f32(x), _f32(gy)
    n, fin = x.shape
    bw, sw, sc, grid = (_f32(p[q]) for q in ("base_weight", "spline_weight", "spline_scaler", "grid"))
    fout = bw.shape[0]
    gx = np.empty((n, fin)); gbw = np.empty((fout, fin)); gsw = np.empty((fout, fin, G + k)); gsc = np.empty((fout, fin))
    lib().kagnn_ref_kan_linear_bwd(_p(x), _p(gy), ctypes.c_int64(n), fin, fout, G, k, _p(grid), _p(bw), _p(sw),
                                   _p(sc), _p(gx), _p(gbw), _p(gsw), _p(gsc))
    return gx, gbw, gsw, gsc


def csr_build(key, val, n):
    key = np.ascontiguousarray(key, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.int64)
    e = key.shape[0]
    rowptr = np.empty(n + 1, dtype=np.int64); col = np.empty(e, dtype=np.int64); perm = np.empty(e, dtype=np.int64)
    rc = lib().kagnn_ref_csr_build(_p(key), _p(val), ctypes.c_int64(e), ctypes.c_int64(n), _p(rowptr), _p(col), _p(perm))
    if rc:
        raise ValueError("node id out of range")
    return rowptr, col, perm


def aggregate(x, src, dst, w=None, self_scale=1.0):
    x = _f32(x)
    n, f = x.shape
    src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
    out = np.empty((n, f), dtype=np.float64)
    wp = None if w is None else _p(_f32(w))
    lib().kagnn_ref_aggregate(_p(x), ctypes.c_int64(n), f, _p(src), _p(dst), ctypes.c_int64(src.shape[0]), wp,
                              ctypes.c_double(self_scale), _p(out))
    return out


if __name__ == "__main__":
    print(build(verbose=True))
