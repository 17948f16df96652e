"""Timing harness: the working counterpart of the reference's ``node_classification_clean/time_model.py``.

``time_model`` reproduces the reference loop :35-48 -- Adam(lr=1e-3), 20 x {zero_grad, forward, softmax,
CrossEntropyLoss on the masked nodes, backward, step} -- including its softmax-before-CE quirk, and fixes
what makes the original unusable as a benchmark: it warms up, synchronises the device
around the timed region, and forwards ``grid_size`` to the model.  Returns seconds per epoch.
"""
from __future__ import annotations

import time

import numpy as np
import torch


def time_model(model, x, edge_index, y, mask, nb_epochs: int = 20, warmup: int = 2):
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)
    criterion = torch.nn.CrossEntropyLoss()
    losses = []

    def epoch():
        optimizer.zero_grad()
        out = model(x, edge_index)
        out = torch.softmax(out, dim=1)              # the reference applies softmax before CE (:43-44)
        loss = criterion(out[mask], y[mask])
        loss.backward()
        optimizer.step()
        return loss

    for _ in range(warmup):
        epoch()
    if x.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nb_epochs):
        losses.append(epoch())
    if x.is_cuda:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nb_epochs
    return float(np.round(dt, 6)), [float(l.detach()) for l in losses]


def count_params(model) -> int:
    return int(sum(p.numel() for p in model.parameters()))
