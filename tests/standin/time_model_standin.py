"""Stand-in for the reference's ``node_classification_clean/time_model.py`` (test data for kagnn_amd.run_reference; NOT a copy: the
three import lines below are the contract under test -- they are the reference's, time_model.py:13-15 -- the rest is this repo's own
minimal timing loop of the same shape as SURVEY.md 3.4 on a synthetic graph, because no dataset exists offline).  Run as

    python -m kagnn_amd.run_reference tests/standin/time_model_standin.py <out.json>

and it must work WITHOUT editing those lines."""
import json
import sys

import torch

from utils import *
import time

from models import GNN_Nodes, GKAN_Nodes, GFASTKAN_Nodes

device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
mp_layers = dataset_layers["standin"]
gen = torch.Generator().manual_seed(0)
n, e, fin, classes = 500, 4000, 12, 5
x = torch.randn(n, fin, generator=gen).to(device)
edge_index = torch.randint(0, n, (2, e), generator=gen).to(device)
y = torch.randint(0, classes, (n,), generator=gen).to(device)
mask = (torch.rand(n, generator=gen) < 0.5).to(device)
report = {"device": str(device), "classes": {}}
for cls, kw in ((GKAN_Nodes, dict(grid_size=4, spline_order=3)), (GFASTKAN_Nodes, dict(grid_size=4))):
    for conv_type in ("gcn", "gin"):
        model = cls(conv_type=conv_type, mp_layers=mp_layers, num_features=fin, hidden_channels=16, num_classes=classes, skip=True,
                    hidden_layers=2, dropout=0.0, **kw).to(device)
        entry = {"module": type(model).__module__, "params": int(sum(p.numel() for p in model.parameters()))}
        try:
            optimizer = torch.optim.Adam(model.parameters(), lr=0.001)
            criterion = torch.nn.CrossEntropyLoss()
            t0 = time.time()
            losses = []
            for _ in range(3):
                optimizer.zero_grad()
                out = torch.softmax(model(x, edge_index), dim=1)
                loss = criterion(out[mask], y[mask])
                loss.backward()
                optimizer.step()
                losses.append(float(loss))
            entry.update(seconds_per_epoch=(time.time() - t0) / 3, losses=losses)
        except RuntimeError as ex:                   # (no GPU: the package refuses CPU tensors, loudly)
            entry["error"] = str(ex)[:200]
        report["classes"][f"{cls.__name__}/{conv_type}"] = entry
try:
    GNN_Nodes("gcn", 2, fin, 16, classes)
    report["GNN_Nodes"] = "constructed (the reference's own class: torch_geometric is installed)"
except ImportError as ex:
    report["GNN_Nodes"] = "ImportError: " + str(ex)[:120]
with open(sys.argv[1], "w") as f:
    json.dump(report, f)
