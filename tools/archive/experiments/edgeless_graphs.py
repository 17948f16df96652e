"""Do the node-level paths take a graph WITHOUT edges?  (GIN = (1 + eps) x, GCN = self loops only, GAT = self attention)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kagnn_amd
from kagnn_amd import ops
dev = "cuda:0"
for n in (1, 2, 50, 70000):
    ei = torch.zeros((2, 0), dtype=torch.int64, device=dev)
    x = torch.randn(n, 24, device=dev)
    for name, make in (("GIKANLayer", lambda: kagnn_amd.GIKANLayer(24, 16, grid_size=5, spline_order=3, hidden_dim=16, nb_layers=2)),
                       ("GIFASTKANLayer", lambda: kagnn_amd.GIFASTKANLayer(24, 16, grid_size=4, hidden_dim=16, nb_layers=2)),
                       ("KAGCNConv", lambda: kagnn_amd.KAGCNConv(24, 16, grid_size=5, spline_order=3)),
                       ("KAGATConv", lambda: kagnn_amd.KAGATConv(24, 8, heads=2, grid_size=5, spline_order=3)),
                       ("GKAN_Nodes gin", lambda: kagnn_amd.GKAN_Nodes("gin", 2, 24, 16, 5, grid_size=5, spline_order=3)),
                       ("GKAN_Nodes gcn", lambda: kagnn_amd.GKAN_Nodes("gcn", 2, 24, 16, 5, grid_size=5, spline_order=3))):
        try:
            torch.manual_seed(0)
            m = make().to(dev)
            if n == 1 and "Nodes" in name:
                m.eval()                      # (BatchNorm in training mode needs two rows, as in torch)
            xr = x.clone().requires_grad_(True)
            y = m(xr, ei)
            y.sum().backward()
            ok = bool(torch.isfinite(y).all()) and bool(torch.isfinite(xr.grad).all())
            print(f"n={n:6d} {name:16s}: out {tuple(y.shape)} finite {ok}", flush=True)
        except Exception as ex:
            print(f"n={n:6d} {name:16s}: FAIL {type(ex).__name__}: {str(ex)[:160]}", flush=True)
