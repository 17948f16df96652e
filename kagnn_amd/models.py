"""KAN-GNN convolutions and node-level models with the reference's class surface.

Mirrors ``node_classification_clean/models.py`` of the reference: ``KANLayer`` :27-29,
``KAGCNConv`` :31-37, ``GIKANLayer`` :48-56, ``FKANLayer`` :58-66, ``FASTKAGCNConv`` :68-74,
``GIFASTKANLayer`` :85-92, ``GKAN_Nodes`` :150-203, ``GFASTKAN_Nodes`` :205-257 -- same constructor
arguments, attribute names (``nn``/``eps`` for the GIN flavour, ``lin``/``bias`` for the GCN
flavour, ``convs``/``bns``/``lay_out`` for the models) and therefore the same state_dict keys.

The reference subclasses torch_geometric's ``GINConv`` / ``GCNConv``; torch_geometric is a
third-party dependency that is not available here, so the two message-passing schemes are
restated on top of ``kagnn_amd.ops`` (CSR built once per ``edge_index``, HIP aggregation kernel):

* GIN  (``GINConv(nn, eps=0, train_eps=False)``):  ``nn((1+eps) * x_i + sum_{j->i} x_j)``
* GCN  (``GCNConv(in, out)`` defaults: self loops added, symmetric normalisation, bias, not
  cached):  ``bias + sum_{j->i or j=i} d_i^-1/2 d_j^-1/2 * lin(x)_j``,  ``d`` = in-degree incl. the loop.

The GAT variants (``KAGATConv`` / ``FASTKAGATConv``) are outside the hot path.
"""
from __future__ import annotations

import os

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .ekan import KAN as eKAN, KANLinear
from .fastkan import FastKAN, FastKANLayer
from .norm import BatchNorm1d


def make_kan(num_features, hidden_dim, out_dim, hidden_layers, grid_size, spline_order):
    sizes = [num_features] + [hidden_dim] * (hidden_layers - 1) + [out_dim]
    return eKAN(layers_hidden=sizes, grid_size=grid_size, spline_order=spline_order)


def make_fastkan(num_features, hidden_dim, out_dim, hidden_layers, grid_size):
    sizes = [num_features] + [hidden_dim] * (hidden_layers - 1) + [out_dim]
    return FastKAN(layers_hidden=sizes, num_grids=grid_size)


class KANLayer(KANLinear):
    def __init__(self, input_dim, output_dim, grid_size=4, spline_order=3):
        super().__init__(in_features=input_dim, out_features=output_dim, grid_size=grid_size,
                         spline_order=spline_order)


class FKANLayer(FastKANLayer):
    def __init__(self, input_dim, output_dim, num_grids=4):
        super().__init__(input_dim=input_dim, output_dim=output_dim, num_grids=num_grids)
        self.num_grids = num_grids

    def reset_parameters(self):
        self.__init__(self.input_dim, self.output_dim, self.num_grids)


# ---------------------------------------------------------------------------------- conv bases
_SPLIT_READOUT = True       # (module attributes: tests and tools/fuzz_models.py move them; no longer environment switches)
_FUSED_LAYER = os.environ.get("KAGNN_FUSED_LAYER", "1") != "0"       # GIN + KAN chain as one autograd node (ops.gin_kan_layer)
_SPLIT_READOUT_MIN_ROWS = 400000
# ... from this many rows when the blocks run as ONE forward launch and hand their gradients to the convolutions
# (ops.kan_linear_parts / ops.SkipGradient; crossover measured in round 3, profiles/r03_experiments.md: 100k rows even, 170k -6 %)
_SPLIT_READOUT_MIN_ROWS_ONE_LAUNCH = 120000
_SKIP_GRADIENT = os.environ.get("KAGNN_SKIP_GRADIENT", "1") != "0"     # skip-branch gradient added inside the next convolution's backward
_FUSED_EPILOGUE = os.environ.get("KAGNN_FUSED_EPILOGUE", "1") != "0"  # conv -> BatchNorm1d -> dropout: statistics + mask fused
_LAZY_NORM = os.environ.get("KAGNN_LAZY_NORM", "1") != "0"             # the norm's forward pass folded into its consumers (ops.AffineRows)


class _SumAggregateConv(nn.Module):
    """GIN message passing: ``nn((1 + eps) * x_i + sum_{j -> i} x_j)`` with a fixed eps buffer."""

    def __init__(self, net: nn.Module, eps: float = 0.0):
        super().__init__()
        self.nn = net
        self.register_buffer("eps", torch.full((1,), float(eps)))
        self._eps_key = None
        self._eps_val = float(eps)

    def _eps(self) -> float:
        if torch.compiler.is_compiling():
            return self._eps_val          # the data_ptr-keyed refresh below is not traceable; eps is a fixed buffer
        key = (self.eps.data_ptr(), self.eps._version)
        if key != self._eps_key:              # one host read per change of the buffer, not per call
            self._eps_val = float(self.eps)
            self._eps_key = key
        return self._eps_val

    def forward_with_moments(self, x: torch.Tensor, edge_index: torch.Tensor, want_moments: bool = True, skip_gradient=None):
        """``(y, moments)``: ``self(x, edge_index)`` (hooks included -- the module output stays a tensor) plus the
        [2, out] column moments of y when the fused node produced them in its last forward kernel, else ``None``; for the
        BatchNorm1d that follows: ``bn(y, moments=...)``.  ``skip_gradient``: an ``ops.SkipGradient`` the fused node
        registers with when it can add a second gradient of ``x`` in its own backward (the skip-concat models)."""
        self.__dict__["_want_moments"] = want_moments
        self.__dict__["_skip_gradient"] = skip_gradient
        try:
            y = self(x, edge_index)
        finally:
            self.__dict__.pop("_want_moments", None)
            self.__dict__.pop("_skip_gradient", None)
        return y, self.__dict__.pop("_moments", None)

    def forward_fused_norm(self, x, edge_index: torch.Tensor, bn: BatchNorm1d, skip_gradient=None, lazy: bool = False):
        """``bn(self(x, edge_index))`` for a training-mode ``BatchNorm1d`` as ONE tape node (``ops._GinKanBnLayerFn``: the
        norm's element-wise backward runs inside the chain's last input-gradient kernel), or ``None`` when the chain is
        outside what the node covers -- nothing has been touched then.  Module hooks of the two modules do NOT run on this
        path (their intermediate tensor is never exposed): ``conv_bn_dropout`` only takes it when neither has any."""
        in_affine = in_stats = None
        if isinstance(x, ops.AffineRows):                 # the previous layer's norm, to be folded into this aggregation
            x, in_affine, in_stats = x.y, x.affine, x.stats
        if not (_FUSED_LAYER and isinstance(self.nn, eKAN) and x.is_cuda and x.size(0) > 1):
            return None
        g = edge_index if isinstance(edge_index, ops.GraphIndex) else ops.graph_index(edge_index, x.size(0))

        def stage():
            factor, use_running = bn.step()
            return (bn.weight, bn.bias, bn.running_mean if use_running else None, bn.running_var if use_running else None, factor, bn.eps)

        return ops.gin_kan_layer(x, g, 1.0 + self._eps(), self.nn, skip_gradient=skip_gradient, batch_norm=stage,
                                 in_affine=in_affine, lazy_norm=lazy, in_stats=in_stats)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        g = edge_index if isinstance(edge_index, ops.GraphIndex) else ops.graph_index(edge_index, x.size(0))
        if _FUSED_LAYER and isinstance(self.nn, eKAN) and x.is_cuda and x.size(0) > 0 and not torch.compiler.is_compiling():
            want = self.__dict__.get("_want_moments", False) and x.size(0) > 1
            y = ops.gin_kan_layer(x, g, 1.0 + self._eps(), self.nn, moments=want,    # one tape node: aggregate + KAN chain
                                  skip_gradient=self.__dict__.get("_skip_gradient"))
            if y is not None:
                if want:
                    y, self.__dict__["_moments"] = y
                return y
        if (x.dtype == torch.bfloat16 or ops.default_activation_dtype() == torch.bfloat16) and not torch.compiler.is_compiling():
            # bf16 gather operands outside the fused node (FastKAN chains): the aggregation takes the bf16 rows and hands
            # fp32 sums to the chain.  Traced code (torch.compile) keeps fp32 rows: the opaque kagnn::aggregate_sum op is
            # registered for fp32 operands and the bf16 conversion is a ctypes call dynamo cannot trace
            xg = x if x.dtype == torch.bfloat16 else ops.to_bf16_rows(x) if not x.requires_grad else x.to(torch.bfloat16)
            return self.nn(ops.aggregate_sum(xg, g, self_scale=1.0 + self._eps()))
        return self.nn(ops.aggregate_sum(x, g, self_scale=1.0 + self._eps()))


class _NormalisedConv(nn.Module):
    """GCN message passing around a KAN transform ``lin`` (transform first, then aggregate)."""

    def __init__(self, lin: nn.Module, out_channels: int):
        super().__init__()
        self.lin = lin
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor,
                edge_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
        if edge_weight is not None or (isinstance(edge_index, torch.Tensor) and edge_index.is_sparse):
            # weighted edges / the sparse adjacency of the reference's gcn timing branch (time_model.py:70-80)
            wg = ops.weighted_gcn_graph(edge_index, edge_weight, x.size(0))
            return ops.aggregate_sum(self.lin(x), wg.graph, self_scale=0.0, edge_weight=wg.weight,
                                     in_scale=wg.dis, out_scale=wg.dis, bias=self.bias)
        g = edge_index if isinstance(edge_index, ops.GraphIndex) else ops.graph_index(edge_index, x.size(0))
        dis = g.gcn_dis
        return ops.aggregate_sum(self.lin(x), g, self_scale=1.0, in_scale=dis, out_scale=dis,
                                 bias=self.bias, skip_self_loops=True)


class KAGCNConv(_NormalisedConv):
    def __init__(self, in_feat: int, out_feat: int, grid_size: int = 4, spline_order: int = 3):
        super().__init__(KANLayer(in_feat, out_feat, grid_size, spline_order), out_feat)


class _AttentionConv(nn.Module):
    """GAT message passing around a KAN transform ``lin`` (torch_geometric 2.5.3 ``GATConv`` with int in_channels,
    ``concat=True``, negative slope 0.2, self loops re-added, no attention dropout): same attribute names and
    state_dict keys (``att_src, att_dst, bias, lin.*``)."""

    def __init__(self, lin: nn.Module, out_channels: int, heads: int):
        super().__init__()
        self.lin = lin
        self.heads, self.out_channels = heads, out_channels
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        nn.init.xavier_uniform_(self.att_src)         # glorot, as GATConv.reset_parameters
        nn.init.xavier_uniform_(self.att_dst)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        g = edge_index if isinstance(edge_index, ops.GraphIndex) else ops.graph_index(edge_index, x.size(0))
        return ops.gat_aggregate(self.lin(x), self.att_src, self.att_dst, self.bias, g, self.heads, self.out_channels)


class KAGATConv(_AttentionConv):
    def __init__(self, in_feat: int, out_feat: int, heads: int, grid_size: int = 4, spline_order: int = 3):
        super().__init__(KANLayer(in_feat, out_feat * heads, grid_size, spline_order), out_feat, heads)


class FASTKAGATConv(_AttentionConv):
    def __init__(self, in_feat: int, out_feat: int, heads: int, grid_size: int = 4):
        super().__init__(FKANLayer(in_feat, out_feat * heads, num_grids=grid_size), out_feat, heads)
        self.grid_size = grid_size


class GIKANLayer(_SumAggregateConv):
    def __init__(self, in_feat: int, out_feat: int, grid_size: int = 4, spline_order: int = 3,
                 hidden_dim: int = 16, nb_layers: int = 2):
        super().__init__(make_kan(in_feat, hidden_dim, out_feat, nb_layers, grid_size, spline_order))


class FASTKAGCNConv(_NormalisedConv):
    def __init__(self, in_feat: int, out_feat: int, grid_size: int = 4):
        super().__init__(FKANLayer(in_feat, out_feat, num_grids=grid_size), out_feat)
        self.grid_size = grid_size


class GIFASTKANLayer(_SumAggregateConv):
    def __init__(self, in_feat: int, out_feat: int, grid_size: int = 4, hidden_dim: int = 16,
                 nb_layers: int = 2):
        super().__init__(make_fastkan(in_feat, hidden_dim, out_feat, nb_layers, grid_size))


# ---------------------------------------------------------------------------------- node models
_FUSED_NORM_BACKWARD = os.environ.get("KAGNN_FUSED_NORM_BACKWARD", "1") != "0"    # conv + BatchNorm1d as one tape node


def _has_hooks(m: nn.Module) -> bool:
    mod = torch.nn.modules.module
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None)
                or mod._global_forward_hooks or mod._global_forward_pre_hooks or mod._global_backward_hooks
                or getattr(mod, "_global_backward_pre_hooks", None))


def conv_bn_dropout(conv, bn, dropout, x, g, *conv_args, skip_gradient=None, lazy=False):
    """The epilogue ``dropout(bn(conv(x)))`` of every message-passing layer (reference
    ``node_classification_clean/models.py:198-201``, ``graph_regression/models.py:107-119``), fused (SURVEY.md 8(f)
    rank 1): the batch statistics come out of the convolution's last forward kernel (``_SumAggregateConv`` over a KAN
    chain) and the dropout mask is applied inside the normalising pass and regenerated in the backward
    (``ops.batch_norm``).  Stock modules under torch.compile and, with dropout on, during stream capture (a captured
    seed would repeat the mask on every replay)."""
    # ``lazy`` (only _NodeModel.forward asks): return an ops.AffineRows -- the convolution's raw output plus the norm's per-column
    # affine -- instead of the normalised rows, when the one-node form below runs; ``x`` may itself be one (the previous layer's)
    x_in = x
    if isinstance(x, ops.AffineRows):
        ok = (_FUSED_EPILOGUE and _FUSED_NORM_BACKWARD and not torch.compiler.is_compiling() and type(dropout) is nn.Dropout
              and isinstance(bn, BatchNorm1d) and isinstance(conv, _SumAggregateConv) and not conv_args and bn.training and bn.affine
              and not (dropout.training and dropout.p > 0.0) and not _has_hooks(conv) and not _has_hooks(bn)
              and type(conv).forward is _SumAggregateConv.forward and type(bn).forward is BatchNorm1d.forward
              and ops.default_activation_dtype() == torch.float32)
        if ok:
            h = conv.forward_fused_norm(x, g, bn, skip_gradient, lazy=lazy)
            if h is not None:
                return h
        x = x_in.materialise()                            # a consumer that cannot fold the affine: write the rows out
    fused = (_FUSED_EPILOGUE and x.is_cuda and not torch.compiler.is_compiling() and type(dropout) is nn.Dropout
             and isinstance(bn, BatchNorm1d)
             and not (dropout.p > 0.0 and dropout.training and torch.cuda.is_current_stream_capturing())
             # a frozen BatchNorm (bn.eval()) under an active dropout: the fused pass ties the mask to the norm's training
             # flag, the reference's dropout(bn(.)) does not -- stock modules then
             and (bn.training or not (dropout.training and dropout.p > 0.0)))
    if not fused:
        return dropout(bn(conv(x, g, *conv_args)))
    if (_FUSED_NORM_BACKWARD and isinstance(conv, _SumAggregateConv) and not conv_args and bn.training and bn.affine
            and not (dropout.training and dropout.p > 0.0) and not _has_hooks(conv) and not _has_hooks(bn)
            # (a subclass with its own forward must see its forward called)
            and type(conv).forward is _SumAggregateConv.forward and type(bn).forward is BatchNorm1d.forward):
        h = conv.forward_fused_norm(x, g, bn, skip_gradient,      # convolution + norm as one tape node
                                    lazy=lazy and ops.default_activation_dtype() == torch.float32 and x.dtype == torch.float32)
        if h is not None:
            return h
    # GINE message passing (graph_regression/models.py:107-119): convolution + norm as one tape node, one library call each way
    if (_FUSED_NORM_BACKWARD and len(conv_args) == 1 and hasattr(conv, "forward_fused_norm") and not isinstance(conv, _SumAggregateConv)
            and bn.training and bn.affine and not (dropout.training and dropout.p > 0.0) and not _has_hooks(conv) and not _has_hooks(bn)
            and type(bn).forward is BatchNorm1d.forward and x.size(0) > 1):
        h = conv.forward_fused_norm(x, g, conv_args[0], bn)
        if h is not None:
            return h
    if isinstance(conv, _SumAggregateConv) and not conv_args and (bn.training or skip_gradient is not None):
        y, mom = conv.forward_with_moments(x, g, want_moments=bn.training, skip_gradient=skip_gradient)
    else:
        y, mom = conv(x, g, *conv_args), None
    return bn(y, moments=mom, dropout_p=dropout.p if dropout.training else 0.0)


class _NodeModel(nn.Module):
    """mp_layers x {conv -> BatchNorm1d -> dropout}, skip-concat of the input and every layer
    output, then a KAN / FastKAN read-out (reference ``models.py:192-203,246-257``)."""

    def _build(self, conv_type, mp_layers, num_features, hidden_channels, skip, dropout, make_conv, heads=1):
        if conv_type not in ("gcn", "gin", "gat"):
            raise ValueError("unknown conv_type")
        if conv_type != "gat":
            heads = 1
        width = hidden_channels * heads                  # conv output width (reference models.py:165-190)
        self.convs = nn.ModuleList()
        self.bns = nn.ModuleList()
        for i in range(mp_layers):
            self.convs.append(make_conv(num_features if i == 0 else width))
            self.bns.append(BatchNorm1d(width))
        self.skip = skip
        self.dropout = nn.Dropout(dropout)
        return num_features + mp_layers * width if skip else width

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        if isinstance(edge_index, ops.GraphIndex):
            g = edge_index
        elif isinstance(edge_index, torch.Tensor) and edge_index.is_sparse:
            # the sparse adjacency of the reference's gcn timing branch (time_model.py:70-80) goes to the convs as it
            # is: _NormalisedConv indexes it once (ops.weighted_gcn_graph caches per tensor identity)
            if not all(isinstance(c, _NormalisedConv) for c in self.convs):
                raise ValueError("a sparse adjacency edge_index is only defined for the gcn convolutions")
            g = edge_index
        else:
            g = ops.graph_index(edge_index, x.size(0))
        outs = [x if x.dtype == torch.float32 else x.float()]      # (bf16 activation storage: the read-out is fp32)
        # large graphs: read-out over [x | h1 | ... ] without concatenating (12.2 vs 13.3 ms per step at 1M nodes;
        # on a 170k-node graph the extra launches cancel the saved copies, so small graphs keep the concat)
        split = self.skip and _SPLIT_READOUT and isinstance(self.lay_out, KANLinear) and (
            x.size(0) >= _SPLIT_READOUT_MIN_ROWS
            or (x.size(0) >= _SPLIT_READOUT_MIN_ROWS_ONE_LAUNCH and x.is_cuda and not torch.compiler.is_compiling()
                and self.lay_out.read_out_blocks_in_one_launch([x.size(1)] + [bn.num_features for bn in self.bns])))
        # h_l feeds the next convolution and the read-out: the read-out's gradient of h_l is added inside the convolution's
        # backward instead of by the tape (ops.SkipGradient; only between the two fused nodes, everything else sums as usual)
        carry = split and _SKIP_GRADIENT and x.is_cuda and torch.is_grad_enabled() and not torch.compiler.is_compiling()
        # the norm's normalising pass folded into its consumers: every layer output then
        # travels as an ops.AffineRows -- raw convolution output + per-column affine -- which the next convolution's aggregation
        # and the read-out's three kernels apply to the rows they load.  Only where every consumer can: the one-launch skip
        # read-out on the split-precision cubic kernels, <= 64 classes, blocks wider than 32 columns, dropout off (the
        # per-layer conditions -- training-mode norm, no hooks, fp32 rows -- are checked by conv_bn_dropout)
        lazy = (_LAZY_NORM and split and x.is_cuda and not torch.compiler.is_compiling() and isinstance(self.lay_out, KANLinear) and self.lay_out.spline_order == 3
                and self.lay_out.grid_size + 3 <= 8 and self.lay_out.out_features <= 64 and not (self.dropout.training and self.dropout.p > 0.0)
                and all(bn.num_features > 32 for bn in self.bns)
                and ops.split_like(self.lay_out.precision if self.lay_out.precision is not None else ops.default_precision()))
        skips = []
        for conv, bn in zip(self.convs, self.bns):
            grad_in = x.y.requires_grad if isinstance(x, ops.AffineRows) else x.requires_grad
            sk = ops.SkipGradient() if carry and grad_in else None
            skips.append(sk)
            x = conv_bn_dropout(conv, bn, self.dropout, x, g, skip_gradient=sk, lazy=lazy)
            outs.append(x)
        if self.skip:
            if split:
                return self.lay_out.forward_parts(outs, skips + [None] if carry else None)
            x = ops.concat_columns(outs)
        return self.lay_out(x)


class GKAN_Nodes(_NodeModel):
    def __init__(self, conv_type: str, mp_layers: int, num_features: int, hidden_channels: int,
                 num_classes: int, skip: bool = True, grid_size: int = 4, spline_order: int = 3,
                 hidden_layers: int = 2, dropout: float = 0., heads=4):
        super().__init__()

        def make_conv(width):
            if conv_type == "gcn":
                return KAGCNConv(width, hidden_channels, grid_size, spline_order)
            if conv_type == "gat":
                return KAGATConv(width, hidden_channels, heads, grid_size, spline_order)
            return GIKANLayer(width, hidden_channels, grid_size, spline_order, hidden_channels, hidden_layers)

        dim = self._build(conv_type, mp_layers, num_features, hidden_channels, skip, dropout, make_conv, heads)
        self.lay_out = KANLinear(dim, num_classes, grid_size=grid_size, spline_order=spline_order)


class GFASTKAN_Nodes(_NodeModel):
    def __init__(self, conv_type: str, mp_layers: int, num_features: int, hidden_channels: int,
                 num_classes: int, skip: bool = True, grid_size: int = 4, hidden_layers: int = 2,
                 dropout: float = 0., heads=4):
        super().__init__()

        def make_conv(width):
            if conv_type == "gcn":
                return FASTKAGCNConv(width, hidden_channels, grid_size)
            if conv_type == "gat":
                return FASTKAGATConv(width, hidden_channels, heads, grid_size)
            return GIFASTKANLayer(width, hidden_channels, grid_size, hidden_channels, hidden_layers)

        dim = self._build(conv_type, mp_layers, num_features, hidden_channels, skip, dropout, make_conv, heads)
        self.lay_out = FastKANLayer(dim, num_classes, num_grids=grid_size)
