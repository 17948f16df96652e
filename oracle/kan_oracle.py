"""CPU oracle for the KAN-GNN layer hot path -- TEST INFRASTRUCTURE ONLY.

This module is a plain-torch, dense-materialising restatement of the algorithm the
reference (RomanBresson/KAGNN) runs for the path SURVEY.md section 8 names.  It is the
checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  ``kagnn_amd`` never does.

Parity status
-------------
* KAN / FastKAN layer math: PINNED.  ``tests/golden/*.npz`` were produced by importing
  the reference's own ``ekan.py`` / ``fastkan.py`` (``tests/golden/make_golden.py``);
  ``tests/test_oracle_golden.py`` checks this file against them (and against the live
  import when ``/root/reference`` is present).
* Message passing (GIN / GCN / GAT / GINE / pool): the arithmetic lives in the un-vendored
  third-party dependency ``torch_geometric==2.5.3`` (reference ``requirements.txt:4``),
  which is neither under ``/root/reference`` nor installable here.  Its published
  semantics are restated below from the reference's call sites
  (``node_classification_clean/models.py:31-92``, ``graph_regression/models.py:86-119``)
  -- **parity unpinned** for that part (the reference holds no tests or vectors for it).

Every function follows the *op sequence* of the reference (whole-tensor elementwise torch
ops, materialised ``[N, in, C]`` bases, one ``F.linear``, stock autograd for backward) so
that timing it is a fair "reference CPU path" baseline.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# efficient-KAN  (reference: node_classification_clean/ekan.py)
# --------------------------------------------------------------------------------------

def make_knots(in_features: int, grid_size: int, spline_order: int,
               grid_range: Sequence[float] = (-1.0, 1.0), dtype=torch.float32) -> Tensor:
    """Uniform extended knot vector, one identical row per input feature.

    Follows ``ekan.py:28-37``: ``arange(-k, G+k+1) * h + lo`` with ``h = (hi-lo)/G``;
    shape ``[in, G + 2k + 1]``.  The arange is integer, the product is fp32 -- keep that
    order so the knots are bit-identical to the reference buffer.
    """
    lo, hi = grid_range
    step = (hi - lo) / grid_size
    row = torch.arange(-spline_order, grid_size + spline_order + 1) * step + lo
    return row.to(dtype).expand(in_features, -1).contiguous()


def bspline_bases(x: Tensor, knots: Tensor, spline_order: int) -> Tensor:
    """Dense Cox-de Boor evaluation, ``ekan.py:79-112``.

    x ``[N, in]``, knots ``[in, G+2k+1]`` -> ``[N, in, G+k]``.  Order-0 is the half-open
    indicator ``t_j <= x < t_{j+1}`` (``ekan.py:95``); each of the k blending steps drops
    one column (``ekan.py:96-105``).
    """
    assert x.dim() == 2 and x.size(1) == knots.size(0)
    xe = x.unsqueeze(-1)
    t = knots
    b = ((xe >= t[:, :-1]) & (xe < t[:, 1:])).to(x.dtype)
    for p in range(1, spline_order + 1):
        rise = (xe - t[:, : -(p + 1)]) / (t[:, p:-1] - t[:, : -(p + 1)])
        fall = (t[:, p + 1:] - xe) / (t[:, p + 1:] - t[:, 1:(-p)])
        b = rise * b[:, :, :-1] + fall * b[:, :, 1:]
    return b.contiguous()


def kan_linear_forward(x: Tensor, base_weight: Tensor, spline_weight: Tensor,
                       spline_scaler: Optional[Tensor], knots: Tensor,
                       spline_order: int) -> Tensor:
    """``KANLinear.forward``, ``ekan.py:146-162``.

    ``silu(x) @ base_weight.T  +  bases.view(N, in*C) @ (spline_weight*scaler).view(out, in*C).T``
    with the flattened K axis ordered (in major, C minor).
    """
    out_features = base_weight.size(0)
    w = spline_weight if spline_scaler is None else spline_weight * spline_scaler.unsqueeze(-1)
    base = F.linear(F.silu(x), base_weight)
    bases = bspline_bases(x, knots, spline_order)
    spline = F.linear(bases.view(x.size(0), -1), w.view(out_features, -1))
    return base + spline


def kan_forward(x: Tensor, layers: Sequence[dict], spline_order: int) -> Tensor:
    """``KAN.forward`` with ``update_grid=False``, ``ekan.py:270-275``: a bare chain of
    KANLinear layers (no activation / norm in between).  ``layers`` is a list of dicts with
    keys ``base_weight, spline_weight, spline_scaler, grid`` (the reference state_dict keys).
    """
    for p in layers:
        x = kan_linear_forward(x, p["base_weight"], p["spline_weight"], p["spline_scaler"],
                               p["grid"], spline_order)
    return x


def curve2coeff(xs: Tensor, ys: Tensor, knots: Tensor, spline_order: int) -> Tensor:
    """Least-squares spline fit used only at init, ``ekan.py:114-144``.
    xs ``[M, in]``, ys ``[M, in, out]`` -> ``[out, in, C]``."""
    a = bspline_bases(xs, knots, spline_order).transpose(0, 1)
    sol = torch.linalg.lstsq(a, ys.transpose(0, 1)).solution
    return sol.permute(2, 0, 1).contiguous()


def update_grid(x: Tensor, layer: dict, grid_size: int, spline_order: int, grid_eps: float = 0.02,
                margin: float = 0.01, solve_dtype=None) -> Tuple[Tensor, Tensor]:
    """``KANLinear.update_grid``, ``ekan.py:164-211``: returns ``(new_grid [in, G+2k+1], new_spline_weight
    [out, in, C])`` for a layer dict with the state_dict keys.

    The per-feature curves the layer currently draws, ``bases_old(x) @ (spline_weight*scaler)`` laid out
    ``[N, in, out]`` (``:169-177``), are re-fitted on the new knots by ``curve2coeff`` (``:211``).  New knots:
    ``G+1`` order statistics of each column at ``linspace(0, N-1, G+1)`` (int64, ``:180-186``), blended with an
    even grid over ``[min - margin, max + margin]`` by ``grid_eps`` (``:188-198``), extended by ``k`` even steps
    either side (``:199-209``).  ``solve_dtype=torch.float64`` runs the fit in double (tighter checker for the
    device's fp64 normal-equation solve); ``None`` keeps the reference's fp32 ``lstsq``."""
    n = x.size(0)
    w = layer["spline_weight"]
    if layer.get("spline_scaler") is not None:
        w = w * layer["spline_scaler"].unsqueeze(-1)
    curves = torch.bmm(bspline_bases(x, layer["grid"], spline_order).permute(1, 0, 2),
                       w.permute(1, 2, 0)).permute(1, 0, 2)                       # [N, in, out]
    xs = torch.sort(x, dim=0)[0]
    adaptive = xs[torch.linspace(0, n - 1, grid_size + 1, dtype=torch.int64)]
    step = (xs[-1] - xs[0] + 2 * margin) / grid_size
    even = torch.arange(grid_size + 1, dtype=torch.float32).unsqueeze(1) * step + xs[0] - margin
    inner = grid_eps * even + (1 - grid_eps) * adaptive
    grid = torch.cat([inner[:1] - step * torch.arange(spline_order, 0, -1).unsqueeze(1), inner,
                      inner[-1:] + step * torch.arange(1, spline_order + 1).unsqueeze(1)], dim=0).T.contiguous()
    if solve_dtype is None:
        return grid, curve2coeff(x, curves, grid, spline_order)
    fit = curve2coeff(x.to(solve_dtype), curves.to(solve_dtype), grid.to(solve_dtype), spline_order)
    return grid, fit.to(torch.float32)


# --------------------------------------------------------------------------------------
# FastKAN  (reference: node_classification_clean/fastkan.py)
# --------------------------------------------------------------------------------------

def rbf_bases(z: Tensor, centers: Tensor, denominator: float) -> Tensor:
    """Gaussian RBF expansion, ``fastkan.py:46-47``: ``exp(-((z[...,None]-c)/den)**2)``."""
    return torch.exp(-((z[..., None] - centers) / denominator) ** 2)


def fastkan_layer_forward(x: Tensor, ln_weight: Optional[Tensor], ln_bias: Optional[Tensor],
                          centers: Tensor, denominator: float, spline_weight: Tensor,
                          base_weight: Optional[Tensor], base_bias: Optional[Tensor],
                          ln_eps: float = 1e-5) -> Tensor:
    """``FastKANLayer.forward``, ``fastkan.py:76-85``.

    LayerNorm(x) -> RBF -> bias-free linear over (in major, grid minor) columns, plus a
    biased linear of ``silu(x)`` on the RAW (not layer-normed) input (``fastkan.py:82-84``).
    """
    z = x
    if ln_weight is not None:
        z = F.layer_norm(x, (x.size(-1),), ln_weight, ln_bias, ln_eps)
    phi = rbf_bases(z, centers, denominator)
    ret = F.linear(phi.view(*phi.shape[:-2], -1), spline_weight)
    if base_weight is not None:
        ret = ret + F.linear(F.silu(x), base_weight, base_bias)
    return ret


def fastkan_forward(x: Tensor, layers: Sequence[dict]) -> Tensor:
    """``FastKAN.forward``, ``fastkan.py:142-145``.  ``layers``: dicts with the reference
    state_dict keys ``layernorm.weight, layernorm.bias, rbf.grid, spline_linear.weight,
    base_linear.weight, base_linear.bias``."""
    for p in layers:
        c = p["rbf.grid"]
        den = (float(c[-1].detach()) - float(c[0].detach())) / (c.numel() - 1)
        x = fastkan_layer_forward(x, p.get("layernorm.weight"), p.get("layernorm.bias"), c, den,
                                  p["spline_linear.weight"], p.get("base_linear.weight"),
                                  p.get("base_linear.bias"))
    return x


# --------------------------------------------------------------------------------------
# Message passing (third-party torch_geometric 2.5.3 semantics, restated; parity unpinned)
# --------------------------------------------------------------------------------------

def sum_aggregate(x: Tensor, edge_index: Tensor, num_nodes: Optional[int] = None,
                  edge_weight: Optional[Tensor] = None) -> Tensor:
    """``MessagePassing.propagate`` with ``aggr='add'``, flow source->target: gather rows
    ``edge_index[0]``, scatter-add at ``edge_index[1]``."""
    n = x.size(0) if num_nodes is None else num_nodes
    msg = x.index_select(0, edge_index[0])
    if edge_weight is not None:
        msg = edge_weight.view(-1, 1) * msg
    out = x.new_zeros((n, x.size(1)))
    out.scatter_add_(0, edge_index[1].view(-1, 1).expand_as(msg), msg)
    return out


def gin_conv(x: Tensor, edge_index: Tensor, nn_fn, eps: float = 0.0) -> Tensor:
    """PyG ``GINConv.forward`` as wrapped by ``GIKANLayer`` / ``GIFASTKANLayer``
    (``node_classification_clean/models.py:48-56,85-92``): ``nn((1+eps)*x_i + sum_j x_j)``."""
    return nn_fn(sum_aggregate(x, edge_index) + (1.0 + eps) * x)


def gcn_norm(edge_index: Tensor, num_nodes: int, dtype=torch.float32,
             edge_weight: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """PyG ``gcn_norm(improved=False, add_self_loops=True)`` restated:
    keep non-loop edges first, then exactly one loop per node (an existing loop's weight is
    kept, otherwise 1.0); ``deg`` = weighted in-degree at ``edge_index[1]``;
    ``w = deg^-1/2[row] * w * deg^-1/2[col]`` with inf -> 0."""
    row, col = edge_index[0], edge_index[1]
    if edge_weight is None:
        edge_weight = torch.ones(row.numel(), dtype=dtype)
    keep = row != col
    loop_w = torch.ones(num_nodes, dtype=dtype)
    inv = ~keep
    if inv.any():
        loop_w[row[inv]] = edge_weight[inv]
    ar = torch.arange(num_nodes, dtype=row.dtype)
    row2 = torch.cat([row[keep], ar])
    col2 = torch.cat([col[keep], ar])
    w2 = torch.cat([edge_weight[keep], loop_w])
    deg = torch.zeros(num_nodes, dtype=dtype).scatter_add_(0, col2, w2)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    return torch.stack([row2, col2]), dis[row2] * w2 * dis[col2]


def gat_conv(x: Tensor, edge_index: Tensor, lin_fn, att_src: Tensor, att_dst: Tensor,
             bias: Optional[Tensor], heads: int, negative_slope: float = 0.2) -> Tensor:
    """PyG 2.5.3 ``GATConv.forward`` (int ``in_channels``, ``concat=True``, ``add_self_loops=True``, no attention
    dropout) as subclassed by ``KAGATConv`` / ``FASTKAGATConv`` (``models.py:39-46,76-83``), restated
    (third-party semantics: parity unpinned): ``xh = lin(x).view(N, H, C)``; logits ``(xh*att).sum(-1)``;
    existing self loops removed, one added per node; ``alpha = softmax_j(leaky_relu(a_src[j] + a_dst[i]))`` over
    the incoming edges of i; ``out_i = sum_j alpha_ij xh_j``, heads concatenated, plus bias."""
    n = x.size(0)
    xh = lin_fn(x).view(n, heads, -1)
    a_s = (xh * att_src.view(1, heads, -1)).sum(-1)
    a_d = (xh * att_dst.view(1, heads, -1)).sum(-1)
    keep = edge_index[0] != edge_index[1]
    ar = torch.arange(n, dtype=edge_index.dtype)
    src = torch.cat([edge_index[0][keep], ar])
    dst = torch.cat([edge_index[1][keep], ar])
    e = F.leaky_relu(a_s[src] + a_d[dst], negative_slope)                       # [E', H]
    emax = torch.full((n, heads), float("-inf"), dtype=e.dtype).scatter_reduce(0, dst.view(-1, 1).expand_as(e), e, "amax")
    p = torch.exp(e - emax[dst])
    z = torch.zeros((n, heads), dtype=e.dtype).index_add_(0, dst, p)
    alpha = p / z[dst]
    out = torch.zeros_like(xh).index_add_(0, dst, alpha.unsqueeze(-1) * xh[src])
    out = out.reshape(n, -1)
    return out if bias is None else out + bias


def gcn_conv(x: Tensor, edge_index: Tensor, lin_fn, bias: Optional[Tensor]) -> Tensor:
    """PyG ``GCNConv.forward`` as subclassed by ``KAGCNConv`` / ``FASTKAGCNConv``
    (``models.py:31-37,68-74``): transform first, then normalised aggregate, then bias."""
    ei, w = gcn_norm(edge_index, x.size(0), x.dtype)
    out = sum_aggregate(lin_fn(x), ei, x.size(0), w)
    return out if bias is None else out + bias


def gine_conv(x: Tensor, edge_index: Tensor, edge_attr: Tensor, nn_fn, eps: float = 0.0) -> Tensor:
    """PyG ``GINEConv`` as used by ``graph_regression/models.py:98,113``:
    message ``relu(x_j + e_ij)``, sum at target, ``+ (1+eps) x_i``, then ``nn``."""
    msg = F.relu(x.index_select(0, edge_index[0]) + edge_attr)
    agg = x.new_zeros(x.shape)
    agg.scatter_add_(0, edge_index[1].view(-1, 1).expand_as(msg), msg)
    return nn_fn(agg + (1.0 + eps) * x)


def global_add_pool(x: Tensor, batch: Tensor, num_graphs: Optional[int] = None) -> Tensor:
    """PyG ``global_add_pool`` (``graph_regression/models.py:117``): scatter-sum over ``batch``."""
    b = int(batch.max()) + 1 if num_graphs is None else num_graphs
    out = x.new_zeros((b, x.size(1)))
    out.scatter_add_(0, batch.view(-1, 1).expand_as(x), x)
    return out


def global_mean_pool(x: Tensor, batch: Tensor, num_graphs: Optional[int] = None) -> Tensor:
    b = int(batch.max()) + 1 if num_graphs is None else num_graphs
    cnt = torch.zeros(b, dtype=x.dtype).scatter_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return global_add_pool(x, batch, b) / cnt.clamp(min=1).view(-1, 1)


def graph_regression_forward(x: Tensor, edge_index: Tensor, edge_attr: Tensor, batch: Tensor, num_graphs: int,
                             state: dict, arch: str, gnn_layers: int, spline_order: int = 3, bn_eps: float = 1e-5) -> Tensor:
    """``KAGIN.forward`` / ``FASTKAGIN.forward`` of ``graph_regression/models.py:107-119,148-160`` in TRAINING mode with dropout 0
    on a reference-keyed ``state`` dict (``atom_encoder.atom_embedding_list.{i}.weight``, ``conv.{l}.nn.layers.{i}.*``,
    ``bn.{l}.weight|bias``, ``kan.layers.{i}.*``): embedding-table encoders over integer features (``models.py:244-281``),
    ``GINEConv`` around the KAN / FastKAN chain, BatchNorm1d on batch statistics, ``global_add_pool``, read-out chain.
    Runs in the dtype of ``state`` (fp64 for the GPU parity tests)."""
    def chain(prefix, h):
        n = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in state if k.startswith(prefix + "layers."))
        if arch == "kan":
            return kan_forward(h, [{q: state[f"{prefix}layers.{i}.{q}"] for q in ("base_weight", "spline_weight", "spline_scaler", "grid")}
                                   for i in range(n)], spline_order)
        keys = ("layernorm.weight", "layernorm.bias", "rbf.grid", "spline_linear.weight", "base_linear.weight", "base_linear.bias")
        return fastkan_forward(h, [{q: state[f"{prefix}layers.{i}.{q}"] for q in keys if f"{prefix}layers.{i}.{q}" in state}
                                   for i in range(n)])

    def tables(prefix, idx):
        out, i = 0, 0
        while f"{prefix}.{i}.weight" in state:
            out = out + state[f"{prefix}.{i}.weight"][idx[:, i]]
            i += 1
        return out

    h = tables("atom_encoder.atom_embedding_list", x)
    e = tables("bond_encoder.bond_embedding_list", edge_attr)
    for l in range(gnn_layers):
        h = gine_conv(h, edge_index, e, lambda t: chain(f"conv.{l}.nn.", t), float(state.get(f"conv.{l}.eps", torch.zeros(1))[0]))
        mu, var = h.mean(0), h.var(0, unbiased=False)
        h = (h - mu) / torch.sqrt(var + bn_eps) * state[f"bn.{l}.weight"] + state[f"bn.{l}.bias"]
    return chain("kan.", global_add_pool(h, batch, num_graphs))


def graph_classification_forward(x: Tensor, edge_index: Tensor, batch: Tensor, num_graphs: int, state: dict, arch: str,
                                 family: str, gnn_layers: int, spline_order: int = 3, bn_eps: float = 1e-5) -> Tensor:
    """The graph-CLASSIFICATION callers of the hot path, ``graph_classification/models.py``, in TRAINING mode with dropout 0 on a
    reference-keyed ``state`` dict:
    ``family='gin'`` -- ``KAGIN.forward`` :107-119 / ``FASTKAGIN.forward`` :141-151: ``gnn_layers x {GINConv(chain) -> BatchNorm1d}``
    -> ``global_add_pool`` -> chain ``kan`` -> ``log_softmax`` (keys ``conv.{l}.nn.layers.{i}.*``, ``bn.{l}.weight|bias``,
    ``kan.layers.{i}.*``);
    ``family='gcn'`` -- ``KAGCN.forward`` :184-194 / ``FASTKAGCN.forward`` :255-265: ``gnn_layers x {GCNConv(lin) -> SiLU}`` ->
    ``global_mean_pool`` -> one-layer chain ``readout`` -> ``log_softmax`` (keys ``conv.{l}.lin.*``, ``conv.{l}.bias``,
    ``readout.layers.0.*``).  ``arch``: 'kan' | 'fastkan'.  Runs in the dtype of ``state`` (fp64 for the GPU parity tests)."""
    kan_keys = ("base_weight", "spline_weight", "spline_scaler", "grid")
    fk_keys = ("layernorm.weight", "layernorm.bias", "rbf.grid", "spline_linear.weight", "base_linear.weight", "base_linear.bias")

    def one(prefix, h):
        if arch == "kan":
            p = {q: state[prefix + q] for q in kan_keys}
            return kan_linear_forward(h, p["base_weight"], p["spline_weight"], p["spline_scaler"], p["grid"], spline_order)
        return fastkan_forward(h, [{q: state[prefix + q] for q in fk_keys if prefix + q in state}])

    def chain(prefix, h):
        n = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in state if k.startswith(prefix + "layers."))
        for i in range(n):
            h = one(f"{prefix}layers.{i}.", h)
        return h

    h = x
    if family == "gin":
        for l in range(gnn_layers):
            eps = float(state[f"conv.{l}.eps"][0]) if f"conv.{l}.eps" in state else 0.0
            h = gin_conv(h, edge_index, lambda t: chain(f"conv.{l}.nn.", t), eps)
            mu, var = h.mean(0), h.var(0, unbiased=False)
            h = (h - mu) / torch.sqrt(var + bn_eps) * state[f"bn.{l}.weight"] + state[f"bn.{l}.bias"]
        return F.log_softmax(chain("kan.", global_add_pool(h, batch, num_graphs)), dim=1)
    if family == "gcn":
        for l in range(gnn_layers):
            h = F.silu(gcn_conv(h, edge_index, lambda t: one(f"conv.{l}.lin.", t), state[f"conv.{l}.bias"]))
        return F.log_softmax(chain("readout.", global_mean_pool(h, batch, num_graphs)), dim=1)
    raise ValueError("family must be 'gin' or 'gcn'")


# --------------------------------------------------------------------------------------
# Node-level models (reference: node_classification_clean/models.py:150-257)
# --------------------------------------------------------------------------------------

def gcn_norm_sparse(adj_t: Tensor) -> Tuple[Tensor, Tensor]:
    """PyG 2.5.3 ``gcn_norm`` for a TORCH SPARSE ``edge_index`` (the form the reference's gcn timing branch hands to
    ``GCNConv``, ``time_model.py:70-80``), restated (third-party semantics, parity unpinned): entry ``(i, j)`` is the
    weight of edge ``j -> i``; self loops are ADDED with ``add_self_loops`` -- weight 1 summed onto whatever the
    diagonal already holds (the tensor is coalesced afterwards), NOT ``add_remaining_self_loops`` as for a dense
    ``edge_index``; ``deg[i] = sum_j adj_t[i, j]``; ``w = deg^-1/2[i] * w * deg^-1/2[j]`` (inf -> 0).
    Returns ``(edge_index[2, E'] as (src j, dst i), weight[E'])``."""
    n = adj_t.size(0)
    a = adj_t.coalesce()
    idx, val = a.indices(), a.values()
    ar = torch.arange(n, dtype=idx.dtype)
    full = torch.sparse_coo_tensor(torch.cat([idx, torch.stack([ar, ar])], dim=1),
                                   torch.cat([val, torch.ones(n, dtype=val.dtype)]), (n, n)).coalesce()
    i, j, w = full.indices()[0], full.indices()[1], full.values()
    deg = torch.zeros(n, dtype=w.dtype).scatter_add_(0, i, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    return torch.stack([j, i]), dis[i] * w * dis[j]


def _rows_chunked(fn, x: Tensor, chunk: Optional[int]) -> Tensor:
    """``fn`` applied to row blocks of ``x`` under activation checkpointing.  A KAN layer treats rows independently,
    so this changes no per-row arithmetic; it only bounds the oracle's memory (the dense ``[N, in, C]`` bases and the
    ~20 elementwise temporaries autograd keeps per layer are ~100 GB at ogbn-arxiv size otherwise)."""
    if chunk is None or x.size(0) <= chunk:
        return fn(x)
    from torch.utils.checkpoint import checkpoint
    return torch.cat([checkpoint(fn, x[i:i + chunk], use_reentrant=False) for i in range(0, x.size(0), chunk)], dim=0)


def node_model_forward(x: Tensor, edge_index, state: dict, arch: str, conv_type: str, mp_layers: int,
                       spline_order: int = 3, skip: bool = True, bn_eps: float = 1e-5, training: bool = True,
                       chunk: Optional[int] = None) -> Tensor:
    """``GKAN_Nodes.forward`` (``models.py:192-203``) / ``GFASTKAN_Nodes.forward`` (``:246-257``) on a state_dict with
    the reference's keys (``convs.{i}.nn.layers.{j}.* | convs.{i}.lin.* + convs.{i}.bias``, ``bns.{i}.*``,
    ``lay_out.*``): ``mp_layers`` x {conv -> BatchNorm1d -> dropout(p=0)}, skip-concat of the input and every layer
    output, KANLinear / FastKANLayer read-out.  ``arch``: 'kan' | 'fastkan'; ``conv_type``: 'gin' | 'gcn'.
    ``edge_index`` may be a torch sparse ``adj_t`` for 'gcn' (``gcn_norm_sparse``).  BatchNorm in training mode uses
    batch statistics (running buffers are not updated here)."""
    def sub(prefix):
        return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}

    def layer_list(d):
        n = 1 + max(int(k.split(".")[1]) for k in d if k.startswith("layers."))
        return [{k[len(f"layers.{i}."):]: v for k, v in d.items() if k.startswith(f"layers.{i}.")} for i in range(n)]

    def kan_fn(p):
        return lambda h: _rows_chunked(lambda r: kan_linear_forward(r, p["base_weight"], p["spline_weight"],
                                                                    p.get("spline_scaler"), p["grid"], spline_order), h, chunk)

    def fk_fn(p):
        return lambda h: _rows_chunked(lambda r: fastkan_forward(r, [p]), h, chunk)

    one = kan_fn if arch == "kan" else fk_fn

    def chain(ps):
        def run(h):
            for p in ps:
                h = one(p)(h)
            return h
        return run

    outs = [x]
    for i in range(mp_layers):
        c = sub(f"convs.{i}.")
        if conv_type == "gin":
            x = gin_conv(x, edge_index, chain(layer_list(sub(f"convs.{i}.nn."))), eps=float(c["eps"]) if "eps" in c else 0.0)
        elif conv_type == "gcn":
            lin = one(sub(f"convs.{i}.lin."))
            if isinstance(edge_index, Tensor) and edge_index.is_sparse:
                ei, w = gcn_norm_sparse(edge_index.to(x.dtype))
                x = sum_aggregate(lin(x), ei, x.size(0), w) + c["bias"]
            else:
                x = gcn_conv(x, edge_index, lin, c["bias"])
        else:
            raise ValueError("unknown conv_type")
        b = sub(f"bns.{i}.")
        x = F.batch_norm(x, None if training else b["running_mean"], None if training else b["running_var"],
                         b["weight"], b["bias"], training, 0.0, bn_eps)
        outs.append(x)
    x = torch.cat(outs, dim=1) if skip else x
    return one(sub("lay_out."))(x)


# --------------------------------------------------------------------------------------
# Integer work: CSR of the edge list (bit-exact contract)
# --------------------------------------------------------------------------------------

def csr_by_key(keys: Tensor, vals: Tensor, num_nodes: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Stable counting sort of the edge list by ``keys``.

    Returns ``(rowptr[int64, N+1], col[int64, E], perm[int64, E])`` with
    ``perm = argsort(keys, stable=True)``, ``col = vals[perm]``,
    ``rowptr = [0, cumsum(bincount(keys, N))]`` -- the layout section 8(c) G7 pins bit-exact.
    ``csr_by_key(dst, src, N)`` is the forward structure, ``csr_by_key(src, dst, N)`` its transpose.
    """
    perm = torch.argsort(keys, stable=True)
    counts = torch.bincount(keys, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr, vals[perm], perm


# --------------------------------------------------------------------------------------
# Whole layers, used by fixtures and by bench.py's cpu_baseline
# --------------------------------------------------------------------------------------

def init_kan_linear(in_features: int, out_features: int, grid_size: int, spline_order: int,
                    gen: torch.Generator, scale_noise: float = 0.1) -> dict:
    """Random KANLinear parameters with the reference's *distributions*
    (``ekan.py:57-77``: kaiming-uniform(a=sqrt5) base and scaler, lstsq-fitted noise spline).
    RNG-stream parity with the reference is NOT promised."""
    knots = make_knots(in_features, grid_size, spline_order)
    bound = 1.0 / math.sqrt(in_features)  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)
    bw = (torch.rand(out_features, in_features, generator=gen) * 2 - 1) * bound
    sc = (torch.rand(out_features, in_features, generator=gen) * 2 - 1) * bound
    noise = (torch.rand(grid_size + 1, in_features, out_features, generator=gen) - 0.5) \
        * scale_noise / grid_size
    sw = curve2coeff(knots.T[spline_order:-spline_order], noise, knots, spline_order)
    return {"base_weight": bw, "spline_weight": sw, "spline_scaler": sc, "grid": knots}


def harness_loss(logits: Tensor, labels: Tensor, mask: Optional[Tensor] = None, pre_softmax: bool = True) -> Tensor:
    """The loss of the timing harness, ``node_classification_clean/time_model.py:43-45``: ``out = softmax(out, dim=1)``
    then ``CrossEntropyLoss()(out[mask], y[mask])`` -- cross-entropy of the *probabilities* (log_softmax applied on
    top of softmax), mean over the masked rows.  ``pre_softmax=False`` is the plain cross-entropy of the training
    loop (``utils.py``)."""
    out = F.softmax(logits, dim=1) if pre_softmax else logits
    if mask is not None:
        out, labels = out[mask], labels[mask]
    return F.cross_entropy(out, labels)


def powerlaw_graph(num_nodes: int, num_edges: int, seed: int = 0) -> Tensor:
    """The synthetic graph recipe of SURVEY.md section 8(d) (seeded, duplicates and
    self-loops kept, unsorted).  Returns ``edge_index[2, E]`` int64."""
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(num_nodes, generator=g)
    u = torch.rand(num_edges, generator=g, dtype=torch.float64)
    dst = perm[torch.floor(num_nodes * u * u).long().clamp_(max=num_nodes - 1)]
    src = torch.randint(0, num_nodes, (num_edges,), generator=g)
    return torch.stack([src, dst])


def kan_gin_layer_fwd_bwd(x: Tensor, edge_index: Tensor, layers: List[dict], spline_order: int,
                          gy: Optional[Tensor] = None):
    """One KAN-GIN conv (aggregate + KAN chain) forward AND backward with stock autograd --
    the unit of work of BASELINE.json's metric.  Returns ``(y, gx, [param grads per layer])``."""
    x = x.detach().clone().requires_grad_(True)
    ps = [{k: (v.detach().clone().requires_grad_(True) if k != "grid" else v)
           for k, v in p.items()} for p in layers]
    y = gin_conv(x, edge_index, lambda h: kan_forward(h, ps, spline_order))
    if gy is None:
        y.sum().backward()
    else:
        y.backward(gy)
    grads = [{k: p[k].grad for k in ("base_weight", "spline_weight", "spline_scaler")} for p in ps]
    return y.detach(), x.grad, grads
