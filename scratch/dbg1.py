import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import kagnn_amd
from kagnn_amd import ops
z = np.load('tests/golden/g2_kanlinear.npz')
tag = '64_64_5_3'
layer = kagnn_amd.KANLinear(64, 64, grid_size=5, spline_order=3)
layer.load_state_dict({n: torch.from_numpy(z[f"{tag}.{n}"]) for n in ("base_weight","spline_weight","spline_scaler","grid")})
layer = layer.cuda(); layer.precision = ops.PREC_SPLIT
x = torch.from_numpy(z[f"{tag}.x"]).cuda()
y = layer(x).detach().cpu().numpy()
want = z[f"{tag}.y"]
d = np.abs(y - want)
bad = np.argwhere(d > 1e-3)
print("bad count", len(bad), "of", d.size)
rows = sorted(set(bad[:,0].tolist())); cols = sorted(set(bad[:,1].tolist()))
print("rows", rows[:40], len(rows)); print("cols", cols[:70], len(cols))
for r in rows[:3]:
    print(r, x[r].cpu().numpy()[:8], np.isnan(x[r].cpu().numpy()).any(), y[r,:4], want[r,:4])
