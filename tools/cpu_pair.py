"""Build container only (/root/reference present): the CPU baseline's cost-faithfulness check of BASELINE.md section 4.2 --
oracle/kan_oracle.py's KAN-GIN layer fwd+bwd timed next to the SAME layer built from the live reference's ekan.KAN
(torch_geometric is absent, so both sides use the oracle's index_select + scatter_add_ aggregation), N=100k / E=1M, hidden 64,
grid 5, 8 threads, 1 warm-up + best of 3.   python tools/cpu_pair.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, "/root/reference/node_classification_clean")
sys.dont_write_bytecode = True
import ekan as ref_ekan
from oracle import kan_oracle as orc

torch.set_num_threads(8)
n, e, f = 100_000, 1_000_000, 64
ei = orc.powerlaw_graph(n, e, seed=0)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, f, generator=g) * 0.25
layers = [orc.init_kan_linear(f, f, 5, 3, g) for _ in range(2)]
kan = ref_ekan.KAN([f, f, f], grid_size=5, spline_order=3)
for l, p in zip(kan.layers, layers):
    l.load_state_dict(p)

def ref_pass():
    xr = x.clone().requires_grad_(True)
    kan.zero_grad()
    orc.gin_conv(xr, ei, kan).sum().backward()

def port_pass():
    orc.kan_gin_layer_fwd_bwd(x, ei, layers, 3)

def best(fn):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)

r, p = best(ref_pass), best(port_pass)
print(f"live reference ekan.KAN: {r:.2f} s   oracle port: {p:.2f} s   ratio {p / r:.3f}   ({e / r:.3g} vs {e / p:.3g} edges/s, 8 threads)")
