"""Build libkagnn_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/kagnn_hip.h).  libkagnn_rccl.so (include/kagnn_rccl.h: the sharded KANLinear on a caller-owned
ncclComm_t) is a second shared object next to it, linking librccl and libkagnn_hip.so."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIBPATH = os.path.join(LIBDIR, "libkagnn_hip.so")
RCCL_LIBPATH = os.path.join(LIBDIR, "libkagnn_rccl.so")
RCCL_SOURCE = "rccl_sharded.hip"
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
SOURCES = ["api.hip", "csr.hip", "aggregate.hip", "aggregate_bf16.hip", "kan_fp32.hip", "kan_split.hip", "kan_sparse_fwd.hip", "kan_split_bwd.hip", "kan_grid.hip", "fastkan.hip", "bn.hip", "gat.hip", "loss.hip", "p2p.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wno-unused-result", "-DNDEBUG"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libkagnn_hip.so cannot be built on this machine")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "kagnn_hip.h"))
    headers.append(os.path.join(os.path.dirname(PKG), "include", "kagnn_rccl.h"))

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIBPATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBPATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    # libkagnn_rccl.so is OPTIONAL (only ShardedGIKANLayer(comm="rccl_c") needs it): a machine without the RCCL headers or
    # librccl still gets the core library; kagnn_amd.rccl.load() raises a clear error when the .so is missing
    try:
        rccl_obj = compile_one(RCCL_SOURCE)
        if force or _stale(RCCL_LIBPATH, [rccl_obj, LIBPATH]):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", RCCL_LIBPATH, rccl_obj,
                   "-L" + LIBDIR, "-lkagnn_hip", "-L" + ROCM_LIB, "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + ROCM_LIB]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    except (subprocess.CalledProcessError, OSError) as ex:
        if os.path.exists(RCCL_LIBPATH):
            os.remove(RCCL_LIBPATH)                  # never leave a stale one next to a newer core library
        sys.stderr.write(f"kagnn_amd._build: libkagnn_rccl.so NOT built ({ex}); the core library is complete, "
                         "comm='rccl_c' will raise at load\n")
    return LIBPATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
