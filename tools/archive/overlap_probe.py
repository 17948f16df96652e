"""Does a memory-bound aggregation overlap with the VALU-bound KAN forward when launched on two streams?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
dev = 'cuda'
n, e, f = 1_000_000, 10_000_000, 64
ei = orc.powerlaw_graph(n, e, seed=0).to(dev)
g = ops.GraphIndex(ei, n)
x = (torch.randn(n, f) * 0.25).to(dev)
h = (torch.randn(n, f) * 0.25).to(dev)
lay = kagnn_amd.KANLinear(f, f, grid_size=5, spline_order=3).to(dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
from ctypes import c_size_t, byref
fb, db = c_size_t(0), c_size_t(0)
ops._call("kagnn_kan_pack_bytes", f, f, 5, 3, 1, byref(fb), byref(db))
pack_f, pack_d = ops._ws(fb.value, x.device), ops._ws(db.value, x.device)
knots = lay.grid[0].contiguous()
ops._call("kagnn_kan_pack", ops._ptr(lay.base_weight), ops._ptr(lay.spline_weight), ops._ptr(lay.spline_scaler), f, f, 5, 3, 1,
          ops._ptr(pack_f), ops._ptr(pack_d), ops._stream())
yk = torch.empty(n, f, device=dev); ya = torch.empty(n, f, device=dev)
def kan():
    ops._call("kagnn_kan_linear_fwd", ops._ptr(h), f, n, ops._ptr(knots), f, f, 5, 3, 1, ops._ptr(pack_f), ops._ptr(yk), f, None, 0, ops._stream())
def agg():
    ops._aggregate_raw(x, g, False, 1.0, None, None, None, None, False, out=ya) if False else ops.aggregate_sum(x, g, self_scale=1.0)
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
def both(order):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        first, second = (kan, agg) if order == 'kan_first' else (agg, kan)
        with torch.cuda.stream(s1): first()
        with torch.cuda.stream(s2): second()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run
def chunks(C):
    rows = [(i * n // C, (i + 1) * n // C) for i in range(C)]
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        for a, b in rows:
            with torch.cuda.stream(s1):
                with torch.no_grad(): lay(h[a:b])
            with torch.cuda.stream(s2): agg()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run
print("agg alone", timeit(agg), "kan alone", timeit(kan), flush=True)
print("concurrent kan_first", timeit(both('kan_first')), "agg_first", timeit(both('agg_first')), flush=True)
