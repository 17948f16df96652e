"""Graph-level KAN-GNN models with the reference's class surface (mini-batches of many small graphs).

Mirrors ``graph_classification/models.py`` (``KAGIN`` :95-119, ``FASTKAGIN`` :125-151, ``KAGCN`` / ``KAGAT`` /
``FASTKAGCN`` / ``FASTKAGAT`` :174-288) and
``graph_regression/models.py`` (``KAGIN`` :86-119 with GINE messages and node/edge encoders): same
constructor arguments, attribute names (``conv``, ``bn``, ``kan``, ``atom_encoder``, ``bond_encoder``)
and state_dict keys (``conv.{i}.eps``, ``conv.{i}.nn.layers.{j}.*``, ``bn.{i}.*``, ``kan.layers.{j}.*``).
``forward(data)`` takes anything with ``x, edge_index, batch`` (and ``edge_attr`` for the regression
model) -- e.g. a torch_geometric ``Batch`` -- and runs message passing + pooling on the HIP kernels:
the whole mini-batch is one CSR (the disjoint union torch_geometric's DataLoader already builds), the
read-out is a segmented sum over the sorted ``batch`` vector (``kagnn_segment_pool``).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import graph_ops, ops
from .models import (FASTKAGATConv, FASTKAGCNConv, GIFASTKANLayer, GIKANLayer, KAGATConv, KAGCNConv, _has_hooks, conv_bn_dropout,
                     make_fastkan, make_kan)
from .norm import BatchNorm1d


# OGB molecule feature cardinalities (lengths of the category lists the reference tabulates in
# graph_regression/models.py:283-345): atomic number, chirality, degree, formal charge, #H, #radical e, hybridisation,
# aromatic, in-ring; bond type, stereo, conjugated
ATOM_FEATURE_DIMS = [119, 5, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


class _SumOfEmbeddings(nn.Module):
    """sum over the integer feature columns of one embedding table each (host-side torch: an index lookup, not part
    of the hot path)"""

    def __init__(self, dims, emb_dim, list_name):
        super().__init__()
        tables = nn.ModuleList()
        for d in dims:
            emb = nn.Embedding(d, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            tables.append(emb)
        setattr(self, list_name, tables)
        self._list_name = list_name

    def _fast_tables(self, x):
        """the tables' weights when ``x`` takes the embedding kernels (``kagnn_embedding_fwd / _bwd``), else ``None``"""
        tables = getattr(self, self._list_name)
        if (x.is_cuda and x.dtype == torch.int64 and x.dim() == 2 and 1 <= x.size(1) <= len(tables) and not torch.compiler.is_compiling()
                and all(type(t) is nn.Embedding and t.padding_idx is None and t.max_norm is None and not t.sparse
                        and not t.scale_grad_by_freq and t.weight.dtype == torch.float32 and t.num_embeddings <= 512 for t in tables)):
            return [tables[i].weight for i in range(x.shape[1])]
        return None

    def forward(self, x):
        tables = getattr(self, self._list_name)
        weights = self._fast_tables(x)
        if weights is not None:
            # one launch per feature column each way (graph_ops._EmbeddingSumFn) instead of gather + add / aten's sort-based backward
            return graph_ops.embedding_sum(x, weights)
        out = 0
        for i in range(x.shape[1]):
            out = out + tables[i](x[:, i])
        return out


class AtomEncoder(_SumOfEmbeddings):
    """``graph_regression/models.py:244-262`` (state_dict keys ``atom_embedding_list.{i}.weight``)."""

    def __init__(self, emb_dim, optional_full_atom_features_dims=None):
        super().__init__(optional_full_atom_features_dims or ATOM_FEATURE_DIMS, emb_dim, "atom_embedding_list")


class BondEncoder(_SumOfEmbeddings):
    """``graph_regression/models.py:264-281`` (state_dict keys ``bond_embedding_list.{i}.weight``)."""

    def __init__(self, emb_dim):
        super().__init__(BOND_FEATURE_DIMS, emb_dim, "bond_embedding_list")


def _num_graphs(data) -> int:
    n = getattr(data, "num_graphs", None)
    return int(n) if n is not None else int(data.batch.max()) + 1


def _segment_ptr(data) -> torch.Tensor:
    """node offsets of the batch's graphs: torch_geometric's ``Batch.ptr`` when the loader supplies it (int64 -> int32, one cast),
    else one binary search over the sorted ``batch`` vector"""
    ptr = getattr(data, "ptr", None)
    if ptr is not None and ptr.numel() == _num_graphs(data) + 1:
        return ptr.to(torch.int32)
    return ops.segment_ptr(data.batch, _num_graphs(data))


# the reference's graph-level files name their convolution layers differently from the node-level file
# (graph_classification/models.py:157-172,... `KAGCN_Layer(GCNConv)`, `KAGAT_Layer(GATConv)`): same layers, same
# constructor arguments
KAGCN_Layer = KAGCNConv
KAGAT_Layer = KAGATConv
FASTKAGCN_Layer = FASTKAGCNConv
FASTKAGAT_Layer = FASTKAGATConv


class _GraphLevel(nn.Module):
    def _message_passing(self, x, g, edge_attr=None):
        # GINE stacks (graph_regression/models.py:107-119): all convolutions + norms as ONE tape node where the shapes allow it
        if (edge_attr is not None and self.training and not (self.dropout.p > 0.0) and all(isinstance(c, GINEKANLayer) for c in self.conv)
                and all(type(b) is BatchNorm1d for b in self.bn) and not any(_has_hooks(m) for m in list(self.conv) + list(self.bn))):
            h = graph_ops.gine_kan_stack(x, edge_attr, g, list(self.conv), list(self.bn))
            if h is not None:
                return h
        for conv, bn in zip(self.conv, self.bn):
            x = conv_bn_dropout(conv, bn, self.dropout, x, g, *(() if edge_attr is None else (edge_attr,)))
        return x

    def _pool(self, x, data):
        return ops.segment_pool(x, _segment_ptr(data))


class KAGIN(_GraphLevel):
    """graph classification: GIN(KAN) stack -> global_add_pool -> KAN read-out -> log_softmax."""

    def __init__(self, gnn_layers, num_features, hidden_dim, num_classes, hidden_layers, grid_size,
                 spline_order, dropout):
        super().__init__()
        self.n_layers = gnn_layers
        self.conv = nn.ModuleList(
            GIKANLayer(num_features if i == 0 else hidden_dim, hidden_dim, grid_size, spline_order, hidden_dim,
                       hidden_layers) for i in range(gnn_layers))
        self.bn = nn.ModuleList(BatchNorm1d(hidden_dim) for _ in range(gnn_layers))
        self.kan = make_kan(hidden_dim, hidden_dim, num_classes, hidden_layers, grid_size, spline_order)
        self.dropout = nn.Dropout(dropout)

    def forward(self, data):
        g = ops.graph_index(data.edge_index, data.x.size(0), cache=False)
        x = self._message_passing(data.x, g)
        return F.log_softmax(self.kan(self._pool(x, data)), dim=1)


class FASTKAGIN(_GraphLevel):
    def __init__(self, gnn_layers, num_features, hidden_dim, num_classes, hidden_layers, grid_size, dropout):
        super().__init__()
        self.n_layers = gnn_layers
        self.conv = nn.ModuleList(
            GIFASTKANLayer(num_features if i == 0 else hidden_dim, hidden_dim, grid_size, hidden_dim, hidden_layers)
            for i in range(gnn_layers))
        self.bn = nn.ModuleList(BatchNorm1d(hidden_dim) for _ in range(gnn_layers))
        self.kan = make_fastkan(hidden_dim, hidden_dim, num_classes, hidden_layers, grid_size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, data):
        g = ops.graph_index(data.edge_index, data.x.size(0), cache=False)
        x = self._message_passing(data.x, g)
        return F.log_softmax(self.kan(self._pool(x, data)), dim=1)


class GINEKANLayer(nn.Module):
    """GINE message passing around a KAN: ``nn((1+eps) x_i + sum_{j->i} relu(x_j + e_ij))``."""

    def __init__(self, net: nn.Module, eps: float = 0.0):
        super().__init__()
        self.nn = net
        self.register_buffer("eps", torch.full((1,), float(eps)))
        self._eps_key, self._eps_val = None, float(eps)

    def _eps(self) -> float:
        key = (self.eps.data_ptr(), self.eps._version)
        if key != self._eps_key:                  # one host read per change of the buffer, not per call
            self._eps_val, self._eps_key = float(self.eps), key
        return self._eps_val

    def forward_fused_norm(self, x, edge_index, edge_attr, bn):
        """``bn(self(x, edge_index, edge_attr))`` for a training-mode BatchNorm1d as ONE tape node (``graph_ops._GineKanLayerFn``: one
        library call each way), or ``None`` when the chain is outside what the node covers -- nothing has been touched then."""
        g = edge_index if isinstance(edge_index, ops.GraphIndex) else ops.graph_index(edge_index, x.size(0))
        return graph_ops.gine_kan_layer(x, edge_attr, g, 1.0 + self._eps(), self.nn, batch_norm=bn)

    def forward(self, x, edge_index, edge_attr):
        g = edge_index if isinstance(edge_index, ops.GraphIndex) else ops.graph_index(edge_index, x.size(0))
        y = graph_ops.gine_kan_layer(x, edge_attr, g, 1.0 + self._eps(), self.nn)          # one tape node: aggregate + KAN chain
        if y is not None:
            return y
        return self.nn(ops.aggregate_gine(x, edge_attr, g, self_scale=1.0 + self._eps()))


class KAGINRegression(_GraphLevel):
    """graph regression (ZINC / QM9 flavour of the reference, ``graph_regression/models.py:86-119``):
    linear (or caller-supplied) node / edge encoders -> GINE(KAN) stack -> global_add_pool -> KAN.
    ``ogb_encoders=True`` uses the OGB Atom/BondEncoder embedding tables on integer features."""

    def __init__(self, num_node_features, num_edge_features, gnn_layers, hidden_dim, hidden_layers, grid_size,
                 spline_order, num_classes, dropout, ogb_encoders=False):
        super().__init__()
        self.n_layers = gnn_layers
        self.atom_encoder = AtomEncoder(hidden_dim) if ogb_encoders else nn.Linear(num_node_features, hidden_dim)
        self.bond_encoder = BondEncoder(hidden_dim) if ogb_encoders else nn.Linear(num_edge_features, hidden_dim)
        self.conv = nn.ModuleList(
            GINEKANLayer(make_kan(hidden_dim, hidden_dim, hidden_dim, hidden_layers, grid_size, spline_order))
            for _ in range(gnn_layers))
        self.bn = nn.ModuleList(BatchNorm1d(hidden_dim) for _ in range(gnn_layers))
        self.kan = make_kan(hidden_dim, hidden_dim, num_classes, hidden_layers, grid_size, spline_order)
        self.dropout = nn.Dropout(dropout)

    def forward(self, data):
        if type(self) is KAGINRegression:                 # the whole forward as ONE tape node where the model and the batch allow it
            out = graph_ops.kagin_regression_forward(self, data)
            if out is not None:
                return out
        x, edge_attr = data.x, data.edge_attr
        if edge_attr.dim() == 1:
            edge_attr = edge_attr.unsqueeze(1)
        x = self.atom_encoder(x)
        edge_attr = self.bond_encoder(edge_attr)
        g = ops.graph_index(data.edge_index, x.size(0), cache=False)
        x = self._message_passing(x, g, edge_attr)
        return self.kan(self._pool(x, data))


class FASTKAGINRegression(KAGINRegression):
    """the FastKAN flavour of the same model (``graph_regression/models.py:125-160``, class ``FASTKAGIN`` there):
    GINE message passing around ``FastKAN`` chains, FastKAN read-out; same attribute names and state_dict keys."""

    def __init__(self, num_node_features, num_edge_features, gnn_layers, hidden_dim, hidden_layers, grid_size,
                 num_classes, dropout, ogb_encoders=False):
        _GraphLevel.__init__(self)
        self.n_layers = gnn_layers
        self.atom_encoder = AtomEncoder(hidden_dim) if ogb_encoders else nn.Linear(num_node_features, hidden_dim)
        self.bond_encoder = BondEncoder(hidden_dim) if ogb_encoders else nn.Linear(num_edge_features, hidden_dim)
        self.conv = nn.ModuleList(
            GINEKANLayer(make_fastkan(hidden_dim, hidden_dim, hidden_dim, hidden_layers, grid_size)) for _ in range(gnn_layers))
        self.bn = nn.ModuleList(BatchNorm1d(hidden_dim) for _ in range(gnn_layers))
        self.kan = make_fastkan(hidden_dim, hidden_dim, num_classes, hidden_layers, grid_size)
        self.dropout = nn.Dropout(dropout)


# ---------------------------------------------------------------------------------- GCN / GAT flavours
class _ConvSiluStack(_GraphLevel):
    """``{conv -> SiLU -> dropout} x L -> pool -> read-out`` of the reference's graph-level KAGCN / KAGAT families
    (``graph_classification/models.py:174-216,245-288``, ``graph_regression/models.py:174-243``): attribute names
    ``conv``, ``readout`` (and ``atom_encoder`` for the regression flavour) as there."""

    def _stack(self, x, g):
        for conv in self.conv:
            x = self.dropout(F.silu(conv(x, g)))
        return x


class KAGCN(_ConvSiluStack):
    """graph classification, mean pooling (``graph_classification/models.py:174-194``)."""

    def __init__(self, gnn_layers, num_features, hidden_dim, num_classes, grid_size, spline_order, dropout):
        super().__init__()
        self.n_layers = gnn_layers
        self.conv = nn.ModuleList(KAGCNConv(num_features if i == 0 else hidden_dim, hidden_dim, grid_size, spline_order)
                                  for i in range(gnn_layers))
        self.readout = make_kan(hidden_dim, hidden_dim, num_classes, 1, grid_size, spline_order)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, data):
        x = self._stack(data.x, ops.graph_index(data.edge_index, data.x.size(0), cache=False))
        ptr = _segment_ptr(data)
        return F.log_softmax(self.readout(ops.segment_pool(x, ptr, mean=True)), dim=1)


class KAGAT(_ConvSiluStack):
    """graph classification, sum pooling (``graph_classification/models.py:196-216``)."""

    def __init__(self, gnn_layers, num_features, hidden_dim, num_classes, grid_size, spline_order, dropout, heads):
        super().__init__()
        self.n_layers = gnn_layers
        self.conv = nn.ModuleList(
            KAGATConv(num_features if i == 0 else hidden_dim * heads, hidden_dim, heads, grid_size, spline_order)
            for i in range(gnn_layers))
        self.readout = make_kan(hidden_dim * heads, hidden_dim, num_classes, 1, grid_size, spline_order)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, data):
        x = self._stack(data.x, ops.graph_index(data.edge_index, data.x.size(0), cache=False))
        return F.log_softmax(self.readout(self._pool(x, data)), dim=1)


class FASTKAGCN(_ConvSiluStack):
    """``graph_classification/models.py:245-265``."""

    def __init__(self, gnn_layers, num_features, hidden_dim, num_classes, grid_size, dropout):
        super().__init__()
        self.n_layers = gnn_layers
        self.conv = nn.ModuleList(FASTKAGCNConv(num_features if i == 0 else hidden_dim, hidden_dim, grid_size)
                                  for i in range(gnn_layers))
        self.readout = make_fastkan(hidden_dim, hidden_dim, num_classes, 1, grid_size)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, data):
        x = self._stack(data.x, ops.graph_index(data.edge_index, data.x.size(0), cache=False))
        ptr = _segment_ptr(data)
        return F.log_softmax(self.readout(ops.segment_pool(x, ptr, mean=True)), dim=1)


class FASTKAGAT(_ConvSiluStack):
    """``graph_classification/models.py:267-288``."""

    def __init__(self, gnn_layers, num_features, hidden_dim, num_classes, grid_size, dropout, heads):
        super().__init__()
        self.n_layers, self.heads = gnn_layers, heads
        self.conv = nn.ModuleList(
            FASTKAGATConv(num_features if i == 0 else hidden_dim * heads, hidden_dim, heads, grid_size)
            for i in range(gnn_layers))
        self.readout = make_fastkan(hidden_dim * heads, hidden_dim, num_classes, 1, grid_size)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, data):
        x = self._stack(data.x, ops.graph_index(data.edge_index, data.x.size(0), cache=False))
        return F.log_softmax(self.readout(self._pool(x, data)), dim=1)


class KAGCNRegression(_ConvSiluStack):
    """graph regression (``graph_regression/models.py:174-198``): linear node encoder, GCN(KAN) stack, sum pooling,
    KAN read-out.  As in the reference the conv layers are built with the DEFAULT grid (4) and order (3) -- the
    constructor's grid_size / spline_order only reach the read-out."""

    def __init__(self, num_node_features, gnn_layers, hidden_dim, grid_size, spline_order, num_classes, dropout,
                 ogb_encoders=False):
        super().__init__()
        self.n_layers = gnn_layers
        self.atom_encoder = AtomEncoder(hidden_dim) if ogb_encoders else nn.Linear(num_node_features, hidden_dim)
        self.conv = nn.ModuleList(KAGCNConv(hidden_dim, hidden_dim) for _ in range(gnn_layers))
        self.readout = make_kan(hidden_dim, hidden_dim, num_classes, 1, grid_size, spline_order)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, data):
        x = self.atom_encoder(data.x)
        x = self._stack(x, ops.graph_index(data.edge_index, x.size(0), cache=False))
        return self.readout(self._pool(x, data))


class FASTKAGCNRegression(_ConvSiluStack):
    """``graph_regression/models.py:218-243``."""

    def __init__(self, num_node_features, gnn_layers, hidden_dim, grid_size, num_classes, dropout, ogb_encoders=False):
        super().__init__()
        self.n_layers = gnn_layers
        self.atom_encoder = AtomEncoder(hidden_dim) if ogb_encoders else nn.Linear(num_node_features, hidden_dim)
        self.conv = nn.ModuleList(FASTKAGCNConv(hidden_dim, hidden_dim, grid_size) for _ in range(gnn_layers))
        self.readout = make_fastkan(hidden_dim, hidden_dim, num_classes, 1, grid_size)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, data):
        x = self.atom_encoder(data.x)
        x = self._stack(x, ops.graph_index(data.edge_index, x.size(0), cache=False))
        return self.readout(self._pool(x, data))
