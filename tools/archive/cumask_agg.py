"""How many CUs does the (HBM-bound) aggregation need?  And the dW kernel?  (CU-masked streams)"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(ncu, total=256):
    words = (ctypes.c_uint32 * (total // 32))()
    for i in range(ncu): words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(total // 32), words) == 0
    return torch.cuda.ExternalStream(s.value)
dev = 'cuda'
n, e, f = 1_000_000, 10_000_000, 64
ei = orc.powerlaw_graph(n, e, seed=0).to(dev)
g = ops.GraphIndex(ei, n)
x = (torch.randn(n, f) * 0.25).to(dev)
lay = kagnn_amd.KANLinear(f, f, grid_size=5, spline_order=3).to(dev)
xr = x.clone().requires_grad_(True)
gy = torch.randn(n, f, device=dev)
def agg(): return ops.aggregate_sum(x, g, self_scale=1.0)
def aggT():
    return ops._aggregate_raw(x, g, True, 1.0, None, None, None, None, False)
def kan_fb():
    lay.zero_grad(); xr.grad = None
    lay(xr).backward(gy)
def timeit(fn, stream, it=5):
    with torch.cuda.stream(stream):
        fn(); stream.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        stream.synchronize()
    return (time.perf_counter() - t0) / it * 1e3
print("default: agg", timeit(agg, torch.cuda.current_stream()), "kan fwd+bwd", timeit(kan_fb, torch.cuda.current_stream()), flush=True)
for c in (256, 192, 128, 96, 64, 48, 32):
    s = masked_stream(c)
    print(c, "CUs: agg ms", round(timeit(agg, s), 3), " kan fwd+bwd ms", round(timeit(kan_fb, s), 3), flush=True)
