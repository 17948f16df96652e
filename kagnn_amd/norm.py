"""BatchNorm1d with the stock module surface (same parameters, buffers and state_dict keys as
``torch.nn.BatchNorm1d``, so ``bns.{i}.*`` checkpoints of the reference load unchanged), computed by
``kagnn_batchnorm_fwd/bwd`` of libkagnn_hip.so -- the epilogue of every convolution in the node and graph
models (reference ``node_classification_clean/models.py:195-202``, ``graph_regression/models.py:107-119``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class BatchNorm1d(nn.BatchNorm1d):
    def step(self):
        """the per-call bookkeeping of ``nn.BatchNorm1d.forward``: counts the batch, -> (momentum factor of this call, whether
        the running statistics take part).  Also used by the node that fuses the norm into its convolution
        (``models.conv_bn_dropout``)."""
        factor = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            factor = 1.0 / float(self.num_batches_tracked) if self.momentum is None else self.momentum
        return factor, (not self.training) or self.track_running_stats

    def forward(self, input: torch.Tensor, moments=None, dropout_p: float = 0.0) -> torch.Tensor:
        """``moments`` / ``dropout_p``: the fused epilogue of ``ops.batch_norm`` (column moments of ``input`` from its
        producer; ``dropout(bn(input), p)`` in one pass while training)"""
        self._check_input_dim(input)
        if input.dim() != 2:
            raise NotImplementedError("kagnn_amd.BatchNorm1d normalises [N, F] node rows only")
        factor, use_running = self.step()
        bn_training = self.training or (self.running_mean is None and self.running_var is None)
        if bn_training and input.size(0) == 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(input.shape)}")
        return ops.batch_norm(input, self.weight, self.bias,
                              self.running_mean if use_running else None,
                              self.running_var if use_running else None,
                              bn_training, factor, self.eps, moments=moments if bn_training else None,
                              dropout_p=dropout_p if self.training else 0.0)
