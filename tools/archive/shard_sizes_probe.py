import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
dev='cuda'; n,e=1_000_000,10_000_000
ei=orc.powerlaw_graph(n,e,seed=0).to(dev); g=ops.GraphIndex(ei,n)
for f in (8,16,32,64):
    x=torch.randn(n,f,device=dev)
    for tr in (False, True):
        fn=lambda: ops._aggregate_raw(x,g,tr,1.0,None,None,None,None,False)
        fn(); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); print(f, "transposed" if tr else "fwd", round((time.perf_counter()-t0)/10*1e3,3), "ms", flush=True)
# row-shard sized KAN layer (N/8 rows)
lay=kagnn_amd.KANLinear(64,64,grid_size=5,spline_order=3).to(dev)
for rows in (125_000, 250_000, 500_000):
    h=(torch.randn(rows,64,device=dev)*0.3).requires_grad_(True); gy=torch.randn(rows,64,device=dev)
    def fb():
        lay.zero_grad(); h.grad=None; lay(h).backward(gy)
    fb(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): fb()
    torch.cuda.synchronize(); print("KANLinear fwd+bwd rows", rows, round((time.perf_counter()-t0)/10*1e3,3), "ms", flush=True)
