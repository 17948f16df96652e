"""GPU box: does the fabric-bound aggregation overlap with the VALU-bound weight-gradient kernel (352 of a SIMD's 512 registers:
the row kernel's 52-register waves fit beside it) when the two are launched on two streams?  And with the input-gradient kernel
(2 x 219 registers: nothing fits beside it)?  usage: python tools/overlap_dw_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
dev = 'cuda'
n, e, f = 1_000_000, 10_000_000, 64
g = ops.GraphIndex(orc.powerlaw_graph(n, e, seed=0).to(dev), n)
x = (torch.randn(n, f) * 0.25).to(dev)
h = (torch.randn(n, f) * 0.25).to(dev)
gy = torch.randn(n, f).to(dev)
lay = kagnn_amd.KANLinear(f, f, grid_size=5, spline_order=3).to(dev)
knots = lay.grid[0].contiguous()
bw, sw, sc = lay.base_weight.detach(), lay.spline_weight.detach(), lay.spline_scaler.detach()
_, pack_d = ops._kan_fwd_raw(h, bw, sw, sc, knots, 5, 3, ops.PREC_SPLIT)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def dw(): ops._kan_bwd_weight_raw(h, gy, knots, sw, sc, f, f, 5, 3, ops.PREC_SPLIT, True)
def dx(): ops._kan_bwd_input_raw(h, gy, knots, pack_d, f, f, 5, 3, ops.PREC_SPLIT)
def agg(): ops._aggregate_raw(x, g, True, 1.0, None, None, None, None, False)
def timeit(fn, it=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
def both(a, b):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): a()
        with torch.cuda.stream(s2): b()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run
def seq(a, b):
    def run(): a(); b()
    return run
ta, tw, tx = timeit(agg), timeit(dw), timeit(dx)
print(f"alone: agg {ta:.3f}  dW {tw:.3f}  dX {tx:.3f} ms", flush=True)
print(f"dW then agg, one stream {timeit(seq(dw, agg)):.3f}   two streams: dW first {timeit(both(dw, agg)):.3f}  agg first {timeit(both(agg, dw)):.3f}", flush=True)
print(f"dX then agg, one stream {timeit(seq(dx, agg)):.3f}   two streams: dX first {timeit(both(dx, agg)):.3f}  agg first {timeit(both(agg, dx)):.3f}", flush=True)
print(f"dW then dX,  one stream {timeit(seq(dw, dx)):.3f}   two streams: dW first {timeit(both(dw, dx)):.3f}  dX first {timeit(both(dx, dw)):.3f}", flush=True)
for thr in (4,):
    pass
