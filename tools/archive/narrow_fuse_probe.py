"""GPU box: the KAN-GIN layer's FORWARD (aggregation + KAN chain, one library call) on narrow first layers -- the per-rank
slices of the feature-sharded layer -- with the aggregation fused into the first KANLinear's kernel (default) or as its own
launch (KAGNN_FUSE_AGG=0; read once per process, so run the script once per setting).  N = 1M, E = 10M, ms per call.
    PYTHONPATH=. python tools/narrow_fuse_probe.py"""
import os, sys, torch
import bench, kagnn_amd
from kagnn_amd import ops
dev = torch.device("cuda", 0)
n, e = 1_000_000, 10_000_000
g = ops.GraphIndex(bench.powerlaw_graph(n, e, 0).to(dev), n)
out = []
for fin in (8, 16, 32):
    torch.manual_seed(0)
    conv = kagnn_amd.GIKANLayer(fin, 64, grid_size=5, spline_order=3, hidden_dim=64, nb_layers=2).to(dev)
    x = (torch.randn(n, fin, generator=torch.Generator().manual_seed(0)) * 0.25).to(dev)
    with torch.no_grad():
        for _ in range(5):
            conv(x, g)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20):
            conv(x, g)
        b.record(); torch.cuda.synchronize()
    out.append(f"in={fin}: {a.elapsed_time(b) / 20:.3f}")
print("KAGNN_FUSE_AGG=" + os.environ.get("KAGNN_FUSE_AGG", "1"), " ".join(out))
