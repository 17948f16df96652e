"""How does kagnn_embedding_bwd's time scale with rows and table size?  (diagnostic for the config-4 step)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kagnn_amd import ops
from kagnn_amd.ops import _call, _ptr, _ld, _stream, _sizes, _ws
dev = "cuda:0"
for n in (128, 512, 2048, 6000, 12700, 100_000):
    for v in (4, 21, 119):
        f = 64
        idx = torch.randint(0, v, (n, 1), device=dev)
        g = torch.randn(n, f, device=dev)
        gt = torch.empty(v, f, device=dev)
        nbytes = _sizes("kagnn_embedding_bwd_workspace_bytes", n, v, f)
        ws = _ws(nbytes, dev)
        def run():
            _call("kagnn_embedding_bwd", idx.data_ptr(), 1, n, _ptr(g), _ld(g), v, f, _ptr(gt), _ptr(ws), ws.numel(), _stream())
        for _ in range(5): run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): run()
        b.record(); torch.cuda.synchronize()
        want = torch.zeros(v, f, device=dev).index_add_(0, idx[:, 0], g)
        err = float((gt - want).abs().max() / want.abs().max())
        print(f"n {n:7d} V {v:4d}: {a.elapsed_time(b) / 50 * 1e3:7.1f} us per call (two launches), rel err {err:.1e}", flush=True)
