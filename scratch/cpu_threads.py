import sys, time, os, torch
sys.path.insert(0, '.')
from oracle import kan_oracle as orc
n, e, f = 50000, 500000, 64
ei = orc.powerlaw_graph(n, e, seed=0)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, f, generator=g) * 0.25
layers = [orc.init_kan_linear(f, f, 5, 3, g) for _ in range(2)]
for th in (8, 32, 64, 128, 256):
    torch.set_num_threads(th)
    orc.kan_gin_layer_fwd_bwd(x, ei, layers, 3)
    t0 = time.perf_counter(); orc.kan_gin_layer_fwd_bwd(x, ei, layers, 3); dt = time.perf_counter() - t0
    print(th, "threads:", round(dt, 2), "s ->", round(e / dt), "edges/s", flush=True)
