"""FastKAN layers with the reference's module surface, computed by libkagnn_hip.so.

Drop-in for ``node_classification_clean/fastkan.py`` of the reference (``SplineLinear`` :22-28,
``RadialBasisFunction`` :30-47, ``FastKANLayer`` :49-85, ``FastKAN`` :118-145): same constructor
signatures and the same state_dict keys (``layernorm.weight, layernorm.bias, rbf.grid,
spline_linear.weight, base_linear.weight, base_linear.bias``).  ``FastKANLayer.forward`` is one
call to ``kagnn_fastkan_fwd`` -- LayerNorm statistics, the Gaussian RBF expansion and both linear
maps are fused in the kernel; ``plot_curve`` and the attention helper of the reference are not
part of the KAGNN path and are not provided.

Attribution: the module surface restated here (class names, constructor signatures, parameter layout and the
trunc-normal / linspace initialisation of ``SplineLinear`` / ``RadialBasisFunction`` / ``FastKANLayer`` / ``FastKAN``)
follows the reference's ``node_classification_clean/fastkan.py``, which is
    Copyright 2024 Li, Ziyao -- Licensed under the Apache License, Version 2.0
    (http://www.apache.org/licenses/LICENSE-2.0); distributed there on an "AS IS" BASIS, WITHOUT WARRANTIES OR
    CONDITIONS OF ANY KIND.
The state_dict / constructor contract is what forces the shared lines; the forward and backward
are this package's HIP kernels.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class SplineLinear(nn.Linear):
    """Bias-free linear map over the (in major, grid minor) RBF columns; trunc-normal init."""

    def __init__(self, in_features: int, out_features: int, init_scale: float = 0.1, **kw) -> None:
        self.init_scale = init_scale
        super().__init__(in_features, out_features, bias=False, **kw)

    def reset_parameters(self) -> None:
        nn.init.trunc_normal_(self.weight, mean=0, std=self.init_scale)


class RadialBasisFunction(nn.Module):
    """Holder of the RBF centres (``grid``, a frozen Parameter as in the reference so it shows up
    in ``parameters()`` / ``state_dict()``) and the common width ``denominator``."""

    def __init__(self, grid_min: float = -2., grid_max: float = 2., num_grids: int = 8,
                 denominator: float = None):
        super().__init__()
        self.grid_min, self.grid_max, self.num_grids = grid_min, grid_max, num_grids
        self.grid = nn.Parameter(torch.linspace(grid_min, grid_max, num_grids), requires_grad=False)
        self.denominator = denominator or (grid_max - grid_min) / (num_grids - 1)


class FastKANLayer(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, grid_min: float = -2., grid_max: float = 2.,
                 num_grids: int = 8, use_base_update: bool = True, use_layernorm: bool = True,
                 base_activation=F.silu, spline_weight_init_scale: float = 0.1) -> None:
        super().__init__()
        if use_base_update and base_activation is not F.silu:
            raise NotImplementedError("the fused kernel implements the SiLU base branch only")
        self.input_dim, self.output_dim = input_dim, output_dim
        self.layernorm = None
        if use_layernorm:
            assert input_dim > 1, "Do not use layernorms on 1D inputs. Set `use_layernorm=False`."
            self.layernorm = nn.LayerNorm(input_dim)
        self.rbf = RadialBasisFunction(grid_min, grid_max, num_grids)
        self.spline_linear = SplineLinear(input_dim * num_grids, output_dim, spline_weight_init_scale)
        self.use_base_update = use_base_update
        self.precision = None                     # None -> ops.default_precision()
        if use_base_update:
            self.base_activation = base_activation
            self.base_linear = nn.Linear(input_dim, output_dim)

    def forward(self, x, use_layernorm=True):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        ln = self.layernorm if (self.layernorm is not None and use_layernorm) else None
        y = ops.fastkan_layer(
            x2,
            None if ln is None else ln.weight, None if ln is None else ln.bias,
            self.spline_linear.weight,
            self.base_linear.weight if self.use_base_update else None,
            self.base_linear.bias if self.use_base_update else None,
            self.rbf.grid, self.rbf.denominator, 1e-5 if ln is None else ln.eps, self.precision)
        return y.reshape(*lead, self.output_dim)


class FastKAN(nn.Module):
    def __init__(self, layers_hidden: List[int], grid_min: float = -2., grid_max: float = 2.,
                 num_grids: int = 8, use_base_update: bool = True, base_activation=F.silu,
                 spline_weight_init_scale: float = 0.1) -> None:
        super().__init__()
        self.layers = nn.ModuleList(
            FastKANLayer(a, b, grid_min=grid_min, grid_max=grid_max, num_grids=num_grids,
                         use_base_update=use_base_update, base_activation=base_activation,
                         spline_weight_init_scale=spline_weight_init_scale)
            for a, b in zip(layers_hidden[:-1], layers_hidden[1:]))

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x
