import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kagnn_amd
from kagnn_amd import ops
dev='cuda'; n=1_000_000
for (fi, fo, G) in ((128,128,8),(64,64,8),(128,128,13)):
    x=(torch.randn(n,fi,device=dev)*0.5)
    lay=kagnn_amd.KANLinear(fi,fo,grid_size=G,spline_order=3).to(dev); lay.precision=ops.PREC_SPLIT
    with torch.no_grad():
        for _ in range(3): lay(x)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20): lay(x)
        torch.cuda.synchronize(); print(fi,fo,G,"fwd ms",(time.perf_counter()-t0)/20*1e3, flush=True)
