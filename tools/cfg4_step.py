#!/usr/bin/env python3
"""BASELINE config 4's ZINC-shaped training step exactly as bench.py's secondary.graph_level_step measures it (harness.train_graph_batches,
8 distinct 256-molecule batches), several repetitions: wall ms per step.  `KAGNN_CFG4_MT=1`: with autograd's worker threads (the
pre-round-6 behaviour) for A/B."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = []
r = bench.graph_level_step_figures(dev, epochs=int(os.environ.get("EPOCHS", "10")))
print(json.dumps({"ms_per_step": round(r["ms_per_step"], 4), "repeats_ms_per_step": [round(v, 4) for v in r["repeats_ms_per_step"]],
                  "loss": r["final_epoch_mean_loss"]}))
