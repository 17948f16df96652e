import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import kagnn_amd
from kagnn_amd import ops
dev='cuda'; n=1_000_000
for fin in (8,16,32,64):
    lay=kagnn_amd.KANLinear(fin,64,grid_size=5,spline_order=3).to(dev)
    h=(torch.randn(n,fin,device=dev)*0.3).requires_grad_(True); gy=torch.randn(n,64,device=dev)
    def fb():
        lay.zero_grad(); h.grad=None; lay(h).backward(gy)
    for _ in range(3): fb()
    tm=ops.EntryPointTimer(); ops.set_timer(tm)
    for _ in range(5): fb()
    torch.cuda.synchronize(); ops.set_timer(None)
    print("in", fin, {k[6:]: round(v["total_ms"]/5,3) for k,v in tm.summary().items()}, flush=True)
