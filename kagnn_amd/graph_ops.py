"""Graph-level (mini-batch) operations on the HIP library: GINE convolutions around KAN chains as one library call each way
(``kagnn_gine_kan_layer_*``), the whole ``n_layers x {GINE conv -> BatchNorm1d}`` stack as ONE call and one tape node each way
(``kagnn_gine_kan_stack_*``), and the embedding-table encoders (``kagnn_embedding_*``).  Reference: ``graph_regression/models.py:98,
107-119,244-281`` as called per mini-batch from ``optuna_zinc.py:56-66`` (BASELINE config 4).  Split out of ``ops.py`` in round 5;
the per-batch CSR build lives in ``ops.GraphIndex`` (``kagnn_csr_build_small``)."""
from __future__ import annotations

import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .ops import (PREC_FP32, GraphIndex, _defer_flag_check, _same_knots, _validate_pending, _batchnorm_fwd_raw, _call, _fits32, _kan_bwd_input_raw, _kan_bwd_weight_raw, _kan_fwd_raw, _ld,
                  _need_cuda, _on_operand_device, _ptr, _ptr_array, _rows, _segment_broadcast_raw, _segment_pool_raw, _sizes, _stream,
                  _weights_key, _ws, default_precision, graph_index, kan_pack_chain, split_like)


class _GineKanLayerFn(Function):
    """One KAN-GINE convolution -- ``KAN((1 + eps) x_i + sum_{j->i} relu(x_j + e_ij))`` -- and, optionally, the training-mode
    ``BatchNorm1d`` that follows it (reference ``graph_regression/models.py:98,107-119``) as ONE tape node over
    ``kagnn_gine_kan_layer_fwd / _bwd``: one library call each way (round 5; BASELINE config 4's mini-batch step is launch- and
    host-bound -- a conv used to be ~7 calls forward and ~10 backward).  Forward: aggregation + one pack launch + the chain, the
    norm's batch statistics from the last kernel's epilogue, then the normalising pass; backward: the norm's statistics pass, its
    element-wise backward inside the last input-gradient kernel, dW / dX per layer, the transposed GINE aggregation (``gx`` and the
    edge-attribute gradient from the same kernel).  Same kernels and summation orders as the composition: same bits."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x, edge_attr, g, self_scale, knots, grid_size, spline_order, mode, bn_w, bn_b, rm, rv, momentum, eps, *params):
        """``bn_w`` .. ``eps``: the training-mode BatchNorm1d (affine) that follows the convolution, or ``bn_w is None``: none"""
        _need_cuda(x, edge_attr, bn_w, bn_b, rm, rv, *params)
        bn = None if bn_w is None else True
        nl = len(params) // 3
        layers = [(params[3 * i].contiguous(), params[3 * i + 1].contiguous(), params[3 * i + 2].contiguous()) for i in range(nl)]
        xg, ea = _rows(x), _rows(edge_attr)
        n, dev = xg.size(0), xg.device
        widths = [layers[0][1].size(1)] + [sw.size(0) for _, sw, _ in layers]
        if n != g.num_nodes or ea.shape != (g.num_edges, widths[0]) or xg.size(1) != widths[0]:
            raise ValueError("x must be [N, F] and edge_attr [E, F] with F the chain's input width, N / E those of the graph")
        acts = [torch.empty((n, w), dtype=torch.float32, device=dev) for w in widths]
        pfs, pds = [], []
        for i in range(nl):
            fb, db = _sizes("kagnn_kan_pack_bytes", widths[i], widths[i + 1], grid_size, spline_order, mode, outputs=2)
            pfs.append(_ws(fb, dev)); pds.append(_ws(db, dev))
        warr = (ctypes.c_int32 * (nl + 1))(*widths)
        wf, _ = _sizes("kagnn_gin_kan_layer_workspace_bytes", n, nl, tuple(widths), grid_size, spline_order, mode, 0, 0, outputs=2)
        ws = _ws(wf, dev)
        mom = torch.empty((2, widths[nl]), dtype=torch.float32, device=dev) if bn is not None else None
        if ea.numel() == 0:              # a batch without edges: the library wants non-null edge arrays; it reads no element of them
            ea = torch.zeros((1, widths[0]), dtype=torch.float32, device=dev)
        perm = g.perm if g.num_edges else torch.zeros(1, dtype=torch.int32, device=dev)
        _call("kagnn_gine_kan_layer_fwd", _ptr(xg), _ld(xg), _ptr(ea), _ld(ea), n, _ptr(g.rowptr), _ptr(g.col), _ptr(perm),
              float(self_scale), nl, warr, _ptr_array([l[0] for l in layers]), _ptr_array([l[1] for l in layers]),
              _ptr_array([l[2] for l in layers]), _ptr(knots), grid_size, spline_order, mode, _ptr_array(acts), _ptr_array(pfs),
              _ptr_array(pds), _ptr(mom[0]) if mom is not None else None, _ptr(mom[1]) if mom is not None else None,
              _ptr(ws), ws.numel(), _stream())
        y = acts[nl]
        saved = [xg, ea]
        for i in range(nl):
            saved += [acts[i], layers[i][1], layers[i][2], pds[i]]
        ctx.meta = (g, self_scale, grid_size, spline_order, mode, nl, widths, bn is not None)
        if bn is None:
            ctx.save_for_backward(*saved, knots)
            return y
        h, mean, rstd = _batchnorm_fwd_raw(y, bn_w, bn_b, rm, rv, True, momentum, eps, mom)
        ctx.save_for_backward(*saved, knots, y, bn_w, mean, rstd)
        return h

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gh):
        g, self_scale, G, K, mode, nl, widths, has_bn = ctx.meta
        t = ctx.saved_tensors
        xg, ea = t[0], t[1]
        knots = t[2 + 4 * nl]
        gh = _rows(gh)
        n, dev = gh.size(0), gh.device
        f32 = dict(dtype=torch.float32, device=dev)
        acts = [t[2 + 4 * i] for i in range(nl)]
        sws, scs, pds = [t[3 + 4 * i] for i in range(nl)], [t[4 + 4 * i] for i in range(nl)], [t[5 + 4 * i] for i in range(nl)]
        gbw = [torch.empty((widths[i + 1], widths[i]), **f32) for i in range(nl)]
        gsw = [torch.empty((widths[i + 1], widths[i], G + K), **f32) for i in range(nl)]
        gsc = [torch.empty((widths[i + 1], widths[i]), **f32) for i in range(nl)]
        gx = torch.empty((n, widths[0]), **f32)
        gea = torch.empty((g.num_edges, widths[0]), **f32) if ctx.needs_input_grad[1] else None
        _, wb = _sizes("kagnn_gin_kan_layer_workspace_bytes", n, nl, tuple(widths), G, K, mode, 0, 0, outputs=2)
        y = bn_w = mean = rstd = g_bnw = g_bnb = None
        if has_bn:
            y, bn_w, mean, rstd = t[3 + 4 * nl:7 + 4 * nl]
            wb += _sizes("kagnn_gin_kan_layer_bwd_bn_workspace_bytes", n, widths[nl])
            g_bnw, g_bnb = torch.empty(widths[nl], **f32), torch.empty(widths[nl], **f32)
        ws = _ws(wb, dev)
        warr = (ctypes.c_int32 * (nl + 1))(*widths)
        perm_t = g.perm_t if g.num_edges else torch.zeros(1, dtype=torch.int32, device=dev)
        _call("kagnn_gine_kan_layer_bwd", _ptr(gh), _ld(gh), _ptr(y), _ld(y) if y is not None else 0, _ptr(bn_w), _ptr(mean), _ptr(rstd),
              _ptr(g_bnw), _ptr(g_bnb), _ptr(xg), _ld(xg), _ptr(ea), _ld(ea), n, _ptr(g.rowptr_t), _ptr(g.col_t), _ptr(perm_t),
              float(self_scale), nl, warr, _ptr_array(sws), _ptr_array(scs), _ptr(knots), G, K, mode, _ptr_array(acts), _ptr_array(pds),
              _ptr(gx), widths[0], _ptr(gea), widths[0], _ptr_array(gbw), _ptr_array(gsw), _ptr_array(gsc), _ptr(ws), ws.numel(), _stream())
        grads = []
        for i in range(nl):
            grads += [gbw[i], gsw[i], gsc[i]]
        return (gx, gea, None, None, None, None, None, None, g_bnw, g_bnb, None, None, None, None, *grads)


class _StackState:
    """what the backward of the GINE stack needs (kept by the tape node that ran the forward)"""
    __slots__ = ("g", "self_scales", "G", "K", "mode", "nconv", "nl", "H", "fb", "db", "xg", "ea", "acts_all", "h_all", "stats", "packs",
                 "knots", "bnw", "sws", "scs")


def _gine_stack_fwd_raw(x, edge_attr, g, self_scales, knots, grid_size, spline_order, mode, running, momentum, eps, nconv, nl, params):
    """``kagnn_gine_kan_stack_fwd`` -> (h of the last convolution, _StackState).  ``params``: per convolution ``bn_weight, bn_bias``
    then per layer ``base_weight, spline_weight, spline_scaler``; ``running``: per convolution ``(running_mean, running_var)`` or
    ``(None, None)``.  Everything a step allocates comes from a handful of tensors (activations, normalised outputs, statistics)
    that the per-layer pointers index into."""
    _need_cuda(x, edge_attr, *params)
    xg, ea = _rows(x), _rows(edge_attr)
    n, dev, H = xg.size(0), xg.device, xg.size(1)
    per = 2 + 3 * nl
    bnw = [params[i * per].contiguous() for i in range(nconv)]
    bnb = [params[i * per + 1].contiguous() for i in range(nconv)]
    bws, sws, scs = [], [], []
    for i in range(nconv):
        for l in range(nl):
            b, w, c = params[i * per + 2 + 3 * l:i * per + 5 + 3 * l]
            bws.append(b.contiguous()); sws.append(w.contiguous()); scs.append(c.contiguous())
    if n != g.num_nodes or ea.shape != (g.num_edges, H):
        raise ValueError("x must be [N, H] and edge_attr [E, H] with N / E those of the graph")
    f32 = dict(dtype=torch.float32, device=dev)
    if ea.numel() == 0:                  # a batch without edges (single atoms): the library wants a non-null pointer; it reads no row
        ea = torch.zeros((1, H), **f32)
    acts_all = torch.empty((nconv, nl + 1, n, H), **f32)
    h_all = torch.empty((nconv, n, H), **f32)
    stats = torch.empty((nconv, 2, H), **f32)
    fb, db = _sizes("kagnn_kan_pack_bytes", H, H, grid_size, spline_order, mode, outputs=2)
    fb, db = (fb + 255) & ~255, (db + 255) & ~255
    packs = torch.empty(nconv * nl * (fb + db), dtype=torch.uint8, device=dev)
    pf_ptr = [packs.data_ptr() + k * fb for k in range(nconv * nl)]
    pd_ptr = [packs.data_ptr() + nconv * nl * fb + k * db for k in range(nconv * nl)]
    widths = (H,) * (nl + 1)
    warr = (ctypes.c_int32 * (nl + 1))(*widths)
    wf, _ = _sizes("kagnn_gine_kan_stack_workspace_bytes", n, nconv, nl, widths, grid_size, spline_order, mode, outputs=2)
    ws = _ws(wf, dev)
    VP, FA = ctypes.c_void_p * (nconv * nl), ctypes.c_float * nconv
    VC = ctypes.c_void_p * nconv
    a0, hs = acts_all.data_ptr(), n * H * 4
    acts_ptr = (ctypes.c_void_p * (nconv * (nl + 1)))(*[a0 + k * hs for k in range(nconv * (nl + 1))])
    h_ptr = VC(*[h_all.data_ptr() + i * hs for i in range(nconv)])
    mean_ptr = VC(*[stats.data_ptr() + (2 * i) * H * 4 for i in range(nconv)])
    rstd_ptr = VC(*[stats.data_ptr() + (2 * i + 1) * H * 4 for i in range(nconv)])
    scale_arr = FA(*[float(v) for v in self_scales])
    _call("kagnn_gine_kan_stack_fwd", _ptr(xg), _ld(xg), _ptr(ea), _ld(ea), n, _ptr(g.rowptr), _ptr(g.col), _ptr(g.perm), scale_arr,
          nconv, nl, warr, _ptr_array(bws), _ptr_array(sws), _ptr_array(scs), _ptr(knots), grid_size, spline_order, mode, acts_ptr,
          VP(*pf_ptr), VP(*pd_ptr), _ptr_array(bnw), _ptr_array(bnb), _ptr_array([r[0] for r in running]),
          _ptr_array([r[1] for r in running]), FA(*[float(v) for v in momentum]), FA(*[float(v) for v in eps]), h_ptr, mean_ptr,
          rstd_ptr, _ptr(ws), ws.numel(), _stream())
    st = _StackState()
    st.g, st.self_scales, st.G, st.K, st.mode, st.nconv, st.nl, st.H, st.fb, st.db = (
        g, tuple(float(v) for v in self_scales), grid_size, spline_order, mode, nconv, nl, H, fb, db)
    st.xg, st.ea, st.acts_all, st.h_all, st.stats, st.packs, st.knots, st.bnw, st.sws, st.scs = xg, ea, acts_all, h_all, stats, packs, knots, bnw, sws, scs
    return h_all[nconv - 1], st


def _gine_stack_bwd_raw(gh, st, need_gea):
    """``kagnn_gine_kan_stack_bwd`` -> (gx, g_edge_attr or None, parameter gradients in the order of ``params``)"""
    g, self_scales, G, K, mode, nconv, nl, H, fb, db = st.g, st.self_scales, st.G, st.K, st.mode, st.nconv, st.nl, st.H, st.fb, st.db
    xg, ea, acts_all, h_all, stats, packs, knots, bnw, sws, scs = st.xg, st.ea, st.acts_all, st.h_all, st.stats, st.packs, st.knots, st.bnw, st.sws, st.scs
    gh = _rows(gh)
    n, dev = gh.size(0), gh.device
    f32 = dict(dtype=torch.float32, device=dev)
    C = G + K
    gx = torch.empty((n, H), **f32)
    gea = torch.empty((g.num_edges, H), **f32) if need_gea else None
    g_bn = torch.empty((nconv, 2, H), **f32)
    g_bw = torch.empty((nconv * nl, H, H), **f32)
    g_sw = torch.empty((nconv * nl, H, H, C), **f32)
    g_sc = torch.empty((nconv * nl, H, H), **f32)
    widths = (H,) * (nl + 1)
    warr = (ctypes.c_int32 * (nl + 1))(*widths)
    _, wb = _sizes("kagnn_gine_kan_stack_workspace_bytes", n, nconv, nl, widths, G, K, mode, outputs=2)
    ws = _ws(wb, dev)
    VP, FA, VC = ctypes.c_void_p * (nconv * nl), ctypes.c_float * nconv, ctypes.c_void_p * nconv
    a0, hs = acts_all.data_ptr(), n * H * 4
    acts_ptr = (ctypes.c_void_p * (nconv * (nl + 1)))(*[a0 + k * hs for k in range(nconv * (nl + 1))])
    pd_ptr = VP(*[packs.data_ptr() + nconv * nl * fb + k * db for k in range(nconv * nl)])
    _call("kagnn_gine_kan_stack_bwd", _ptr(gh), _ld(gh), _ptr(xg), _ld(xg), _ptr(ea), _ld(ea), n, _ptr(g.rowptr_t), _ptr(g.col_t),
          _ptr(g.perm_t), FA(*self_scales), nconv, nl, warr, _ptr_array(sws), _ptr_array(scs), _ptr(knots), G, K, mode, acts_ptr, pd_ptr,
          VC(*[h_all.data_ptr() + i * hs for i in range(nconv)]), _ptr_array(bnw),
          VC(*[stats.data_ptr() + (2 * i) * H * 4 for i in range(nconv)]), VC(*[stats.data_ptr() + (2 * i + 1) * H * 4 for i in range(nconv)]),
          _ptr(gx), H, _ptr(gea), H, VC(*[g_bn.data_ptr() + (2 * i) * H * 4 for i in range(nconv)]),
          VC(*[g_bn.data_ptr() + (2 * i + 1) * H * 4 for i in range(nconv)]),
          VP(*[g_bw.data_ptr() + k * H * H * 4 for k in range(nconv * nl)]), VP(*[g_sw.data_ptr() + k * H * H * C * 4 for k in range(nconv * nl)]),
          VP(*[g_sc.data_ptr() + k * H * H * 4 for k in range(nconv * nl)]), _ptr(ws), ws.numel(), _stream())
    grads = []
    for i in range(nconv):
        grads += [g_bn[i, 0], g_bn[i, 1]]
        for l in range(nl):
            k = i * nl + l
            grads += [g_bw[k], g_sw[k], g_sc[k]]
    return gx, gea, grads


class _GineKanStackFn(Function):
    """The whole message-passing stack of a graph-level model -- ``nconv x {GINE convolution around a KAN chain -> training-mode
    BatchNorm1d}``, all chains hidden -> ... -> hidden (reference ``graph_regression/models.py:107-119``) -- as ONE tape node over
    ``kagnn_gine_kan_stack_fwd / _bwd`` (round 5; see include/kagnn_hip.h: on a 256-molecule batch the per-convolution nodes cost the
    host as much as the device)."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x, edge_attr, g, self_scales, knots, grid_size, spline_order, mode, running, momentum, eps, nconv, nl, *params):
        h, st = _gine_stack_fwd_raw(x, edge_attr, g, self_scales, knots, grid_size, spline_order, mode, running, momentum, eps, nconv, nl, params)
        ctx.meta = (st.g, st.self_scales, st.G, st.K, st.mode, st.nconv, st.nl, st.H, st.fb, st.db)
        ctx.save_for_backward(st.xg, st.ea, st.acts_all, st.h_all, st.stats, st.packs, st.knots, *st.bnw, *st.sws, *st.scs)
        return h

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gh):
        st = _StackState()
        st.g, st.self_scales, st.G, st.K, st.mode, st.nconv, st.nl, st.H, st.fb, st.db = ctx.meta
        t = ctx.saved_tensors
        st.xg, st.ea, st.acts_all, st.h_all, st.stats, st.packs, st.knots = t[:7]
        nconv, nl = st.nconv, st.nl
        st.bnw = t[7:7 + nconv]
        st.sws = t[7 + nconv:7 + nconv + nconv * nl]
        st.scs = t[7 + nconv + nconv * nl:7 + nconv + 2 * nconv * nl]
        gx, gea, grads = _gine_stack_bwd_raw(gh, st, ctx.needs_input_grad[1])
        return (gx, gea, None, None, None, None, None, None, None, None, None, None, None, *grads)


_GINE_STACK_ABI = True      # False: one tape node per convolution (bit-identical)


def _gine_stack_plan(x, convs, bns):
    """Is ``for conv, bn in zip(convs, bns): x = bn(conv(x, g, edge_attr))`` within what ``_GineKanStackFn`` covers?  -> ``(first
    layer, layers per chain, mode)`` or ``None``.  No side effects (the norms' batch counters move in ``_gine_stack_args``)."""
    if (not _GINE_STACK_ABI or not _GINE_LAYER_ABI or torch.compiler.is_compiling() or not x.is_cuda or x.dtype != torch.float32
            or x.size(0) < 2 or len(convs) < 2):
        return None
    H = x.size(1)
    first = None
    nl = None
    all_layers = []
    for conv, bn in zip(convs, bns):
        layers = list(getattr(conv.nn, "layers", []))
        if nl is None:
            nl = len(layers)
        if not (1 <= len(layers) == nl) or any(type(l).__name__ != "KANLinear" for l in layers):
            return None
        if first is None:
            first = layers[0]
        for l in layers:
            if (l.in_features != H or l.out_features != H or l.precision != first.precision or l.grid_size != first.grid_size
                    or l.spline_order != first.spline_order or not l.enable_standalone_scale_spline):
                return None
        all_layers += layers
        if not (bn.training and bn.affine and bn.num_features == H):
            return None
    #  ONE knot vector for every layer of every convolution, verified -- not assumed from the first layer
    knots = [l._knots() for l in all_layers]
    if any(k.dim() != 1 or k.numel() != knots[0].numel() for k in knots) or not _same_knots(all_layers, knots):
        return None
    mode = first.precision if first.precision is not None else default_precision()
    if not split_like(mode) or first.spline_order != 3 or first.grid_size + first.spline_order > 8 or H > 64 or len(convs) * nl > 16:
        return None
    return first, nl, mode


def _gine_stack_args(convs, bns):
    """the per-convolution arguments of the stack node; counts the batch in every norm (``BatchNorm1d.step``)"""
    params, scales, running, momentum, eps = [], [], [], [], []
    # the norms' batch counters: one multi-tensor increment for all of them where BatchNorm1d.step would do `counter.add_(1)` each
    # (training, tracked statistics, a fixed momentum -- with momentum=None the factor is read back from the counter: left to step())
    together = len(bns) > 1 and all(bn.training and bn.track_running_stats and bn.num_batches_tracked is not None and bn.momentum is not None
                                    for bn in bns)
    if together:
        torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
    for conv, bn in zip(convs, bns):
        factor, use_running = (bn.momentum, True) if together else bn.step()
        params += [bn.weight, bn.bias]
        for l in conv.nn.layers:
            params += [l.base_weight, l.spline_weight, l.spline_scaler]
        scales.append(1.0 + conv._eps())
        running.append((bn.running_mean, bn.running_var) if use_running else (None, None))
        momentum.append(factor); eps.append(bn.eps)
    return params, tuple(scales), tuple(running), tuple(momentum), tuple(eps)


def gine_kan_stack(x, edge_attr, g: "GraphIndex", convs, bns):
    """``for conv, bn in zip(convs, bns): x = bn(conv(x, g, edge_attr))`` as ONE tape node (``_GineKanStackFn``), or ``None`` when the
    stack is outside what the node covers (the caller then runs the loop): GINE convolutions around KAN chains of identical
    hidden -> ... -> hidden widths (<= 64: one pack launch for the stack), one uniform grid and precision for all of them, a
    split-like mode, training-mode affine BatchNorm1d modules, fp32 CUDA rows, at most 16 KANLinears in all, not under torch.compile."""
    plan = _gine_stack_plan(x, convs, bns)
    if plan is None:
        return None
    first, nl, mode = plan
    params, scales, running, momentum, eps = _gine_stack_args(convs, bns)
    return _GineKanStackFn.apply(x, edge_attr, g, scales, first._knots(), first.grid_size, first.spline_order, mode, running,
                                 momentum, eps, len(convs), nl, *params)


_GINE_LAYER_ABI = True      # False: the per-operation composition (module attributes for the A/B tests, not environment switches)


def gine_kan_layer(x, edge_attr, g: "GraphIndex", self_scale: float, net, batch_norm=None):
    """``net(aggregate_gine(x, edge_attr))`` -- and ``bn(.)`` of it when ``batch_norm`` (a training-mode ``kagnn_amd.BatchNorm1d``
    with affine parameters) is given -- as one tape node (``_GineKanLayerFn``), or ``None`` when the chain is outside what the
    node covers (the caller then composes the operations): a KAN chain of uniform grids, one precision, split-like mode, <= 8
    layers, fp32 CUDA rows, not under torch.compile."""
    if not _GINE_LAYER_ABI or torch.compiler.is_compiling() or not x.is_cuda or x.dtype != torch.float32 or x.size(0) < 2:
        return None
    layers = list(getattr(net, "layers", []))
    if not (1 <= len(layers) <= 8) or any(type(l).__name__ != "KANLinear" for l in layers):
        return None
    first = layers[0]
    mode = first.precision if first.precision is not None else default_precision()
    if not split_like(mode) or first.grid_size + first.spline_order > 16:
        return None
    if any(l.precision != first.precision or l.grid_size != first.grid_size or l.spline_order != first.spline_order
           or not l.enable_standalone_scale_spline for l in layers):
        return None
    #  the library evaluates the whole chain on ONE knot vector: layers with uniform but different grids (another
    # grid_range, update_grid with grid_eps = 1) must not be folded onto the first layer's -- the guards of ops.gin_kan_layer
    knots = [l._knots() for l in layers]
    if any(k.dim() != 1 or k.numel() != knots[0].numel() for k in knots) or not _same_knots(layers, knots):
        return None
    if max(max(l.in_features, l.out_features) for l in layers) > 7680:
        return None
    params = []
    for l in layers:
        params += [l.base_weight, l.spline_weight, l.spline_scaler]
    bn = (None, None, None, None, 0.0, 0.0)
    if batch_norm is not None:
        factor, use_running = batch_norm.step()
        bn = (batch_norm.weight, batch_norm.bias, batch_norm.running_mean if use_running else None,
              batch_norm.running_var if use_running else None, factor, batch_norm.eps)
    return _GineKanLayerFn.apply(x, edge_attr, g, float(self_scale), knots[0], first.grid_size, first.spline_order, mode, *bn, *params)



def _embedding_sum_fwd_raw(x, tables):
    """-> (sum_c tables[c][x[:, c]], the contiguous index matrix, the tables' shapes)"""
    _need_cuda(x, *tables)
    if x.dtype != torch.int64 or x.dim() != 2 or x.size(1) != len(tables):
        raise ValueError("x must be an int64 [N, columns] matrix with one table per column")
    x = x.contiguous()
    n, cols, f = x.size(0), x.size(1), tables[0].size(1)
    out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    tabs = [t.contiguous() for t in tables]
    for c, t in enumerate(tabs):
        if t.dtype != torch.float32 or t.size(1) != f:
            raise ValueError("embedding tables must be fp32 [V, F] with one F")
        _call("kagnn_embedding_fwd", x.data_ptr() + 8 * c, cols, n, _ptr(t), t.size(0), f, _ptr(out), f, int(c > 0), _stream())
    return out, x, [tuple(t.shape) for t in tabs]


def _embedding_sum_bwd_raw(x, g, shapes, wanted=None):
    """table gradients of ``_embedding_sum_fwd_raw`` (``None`` where ``wanted[c]`` is false)"""
    g = _rows(g)
    n, cols = x.shape
    grads = []
    for c, (v, f) in enumerate(shapes):
        if wanted is not None and not wanted[c]:
            grads.append(None)
            continue
        gt = torch.empty((v, f), dtype=torch.float32, device=g.device)
        ws = _ws(_sizes("kagnn_embedding_bwd_workspace_bytes", n, v, f), g.device)
        _call("kagnn_embedding_bwd", x.data_ptr() + 8 * c, cols, n, _ptr(g), _ld(g), v, f, _ptr(gt), _ptr(ws), ws.numel(), _stream())
        grads.append(gt)
    return grads


class _EmbeddingSumFn(Function):
    """``sum_c tables[c][x[:, c]]`` (the OGB-style Atom / BondEncoder of the graph-level models, reference
    ``graph_regression/models.py:244-281``): one launch per feature column each way (``kagnn_embedding_fwd / _bwd``) instead of a
    gather + add per column forward and aten's sort-based ``embedding_dense_backward`` (~12 launches per table) backward."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x, *tables):
        out, xc, shapes = _embedding_sum_fwd_raw(x, tables)
        ctx.save_for_backward(xc)
        ctx.shapes = shapes
        return out

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return (None, *_embedding_sum_bwd_raw(x, g, ctx.shapes, ctx.needs_input_grad[1:]))


def embedding_sum(x: torch.Tensor, tables) -> torch.Tensor:
    return _EmbeddingSumFn.apply(x, *tables)


# ======================================================================== the whole regression model as one tape node
class _ModelPlan:
    """the non-tensor arguments of ``_KaginModelFn`` (one object instead of ~20 positional Python values)"""
    __slots__ = ("n_atom", "n_bond", "n_stack", "n_readout", "scales", "running", "momentum", "eps", "nconv", "nl", "knots", "G", "K", "mode",
                 "ro_knots", "ro_G", "ro_K", "ro_modes")


class _KaginModelFn(Function):
    """``KAGIN.forward`` of the graph-regression models (reference ``graph_regression/models.py:107-119``: Atom / BondEncoder ->
    ``n_layers x {GINEConv(KAN) -> BatchNorm1d}`` -> ``global_add_pool`` -> KAN read-out) as ONE tape node.  No new kernels and no new
    library entry points: the node runs the calls of its five constituents (``_EmbeddingSumFn`` x 2, ``_GineKanStackFn``,
    ``_SegmentPoolFn``, ``_KANLinearFn`` per read-out layer) back to back -- same kernels, same order per tensor, same bits.  What
    it removes is Python: on a 256-molecule mini-batch the device needs ~0.8 ms per training step and the host ~1.2 ms, a third of
    it ``Function.apply`` / ``nn.Module.__call__`` / autograd-engine overhead of those nodes (``profiles/r05_experiments.md`` 9)."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x_int, e_int, g, seg, plan, *params):
        a, b, c = plan.n_atom, plan.n_atom + plan.n_bond, plan.n_atom + plan.n_bond + plan.n_stack
        x0, xi, ashapes = _embedding_sum_fwd_raw(x_int, params[:a])
        ea, ei, bshapes = _embedding_sum_fwd_raw(e_int, params[a:b])
        h, st = _gine_stack_fwd_raw(x0, ea, g, plan.scales, plan.knots, plan.G, plan.K, plan.mode, plan.running, plan.momentum, plan.eps,
                                    plan.nconv, plan.nl, params[b:c])
        pooled = _segment_pool_raw(h, seg, False)
        ro = params[c:]
        layers = [(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]) for i in range(plan.n_readout)]
        packs = None
        if plan.n_readout > 1 and len(set(plan.ro_modes)) == 1 and split_like(plan.ro_modes[0]):
            packs = kan_pack_chain(layers, plan.ro_G, plan.ro_K, plan.ro_modes[0])         # one pack launch (None: each layer packs itself)
        acts, pds, kept = [pooled], [], []
        for i, (bw, sw, sc) in enumerate(layers):
            bw_c, sw_c, sc_c = bw.contiguous(), sw.contiguous(), None if sc is None else sc.contiguous()
            y, pd = _kan_fwd_raw(acts[-1], bw_c, sw_c, sc_c, plan.ro_knots[i], plan.ro_G, plan.ro_K, plan.ro_modes[i],
                                 None if packs is None else packs[i], _weights_key(bw, sw, sc) if packs is not None else None)
            acts.append(y); pds.append(pd); kept.append((sw_c, sc_c))
        #  everything the backward reads goes through save_for_backward (autograd's in-place checks on weights and
        # activations apply, a retained graph can run twice) and the OUTPUT is not among it: acts[-1] in the state made an
        # output -> grad_fn -> state -> output cycle that kept a step's whole activation set alive until the cyclic collector ran
        flat_kept = []
        for sw_c, sc_c in kept:
            flat_kept += [sw_c, sc_c]
        tensors = [xi, ei, st.xg, st.ea, st.acts_all, st.h_all, st.stats, st.packs, st.knots, *st.bnw, *st.sws, *st.scs,
                   *acts[:-1], *pds, *flat_kept]
        ctx.save_for_backward(*tensors)
        ctx.meta = (plan, g, seg, ashapes, bshapes, h.size(0), len(st.bnw), len(st.sws),
                    (st.g, st.self_scales, st.G, st.K, st.mode, st.nconv, st.nl, st.H, st.fb, st.db))
        return acts[-1]

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gout):
        plan, g, seg, ashapes, bshapes, n, nbn, nsw, stmeta = ctx.meta
        t = ctx.saved_tensors
        st = _StackState()
        st.g, st.self_scales, st.G, st.K, st.mode, st.nconv, st.nl, st.H, st.fb, st.db = stmeta
        xi, ei, st.xg, st.ea, st.acts_all, st.h_all, st.stats, st.packs, st.knots = t[:9]
        o = 9
        st.bnw = t[o:o + nbn]; o += nbn
        st.sws = t[o:o + nsw]; o += nsw
        st.scs = t[o:o + nsw]; o += nsw
        nr = plan.n_readout
        acts = t[o:o + nr]; o += nr
        pds = t[o:o + nr]; o += nr
        kept = [(t[o + 2 * i], t[o + 2 * i + 1]) for i in range(nr)]
        gy = _rows(gout)
        ro_grads = [None] * (3 * plan.n_readout)
        for i in reversed(range(plan.n_readout)):
            x_i, (sw_c, sc_c) = acts[i], kept[i]
            fin, fout, G, K, mode = x_i.size(1), sw_c.size(0), plan.ro_G, plan.ro_K, plan.ro_modes[i]
            gx = _kan_bwd_input_raw(x_i, gy, plan.ro_knots[i], pds[i], fin, fout, G, K, mode)
            ro_grads[3 * i:3 * i + 3] = _kan_bwd_weight_raw(x_i, gy, plan.ro_knots[i], sw_c, sc_c, fin, fout, G, K, mode, True)
            gy = gx
        gh = _segment_broadcast_raw(gy, seg, n, False)
        gx0, gea, sgrads = _gine_stack_bwd_raw(gh, st, True)
        agrads = _embedding_sum_bwd_raw(xi, gx0, ashapes)
        bgrads = _embedding_sum_bwd_raw(ei, gea, bshapes)
        return (None, None, None, None, None, *agrads, *bgrads, *sgrads, *ro_grads)


# ======================================================================== ... and as ONE library call each way (round 6)
_MODEL_SIZES: dict = {}


def _model_struct(plan, params, cache_owner):
    """a ``kagnn_kagin_model_t`` with everything that does not change from step to step filled in -- parameter pointers, table
    shapes, layer configuration -- copied from a per-model template as long as the parameters still live where they did"""
    from . import _lib
    ptrs = [0 if t is None else t.data_ptr() for t in params]
    key = (tuple(ptrs), plan.n_atom, plan.n_bond, plan.nconv, plan.nl, plan.G, plan.K, plan.mode, plan.ro_G, plan.ro_K, tuple(plan.ro_modes),
           plan.scales, plan.momentum, plan.eps, tuple(0 if r[0] is None else r[0].data_ptr() for r in plan.running), plan.knots.data_ptr(),
           tuple(k.data_ptr() for k in plan.ro_knots))
    hit = cache_owner.__dict__.get("_kagnn_model_call_template")
    if hit is None or hit[0] != key:
        m = _lib.KaginModel()
        a, b, c = plan.n_atom, plan.n_atom + plan.n_bond, plan.n_atom + plan.n_bond + plan.n_stack
        H = params[0].size(1)
        m.hidden, m.num_atom_tables, m.num_bond_tables = H, plan.n_atom, plan.n_bond
        for t in range(plan.n_atom):
            m.atom_table[t], m.atom_rows[t] = ptrs[t], params[t].size(0)
        for t in range(plan.n_bond):
            m.bond_table[t], m.bond_rows[t] = ptrs[a + t], params[a + t].size(0)
        m.num_convs, m.num_layers, m.grid_size, m.spline_order, m.mode = plan.nconv, plan.nl, plan.G, plan.K, plan.mode
        per = 2 + 3 * plan.nl
        for i in range(plan.nconv):
            m.bn_weight[i], m.bn_bias[i] = ptrs[b + i * per], ptrs[b + i * per + 1]
            rm, rv = plan.running[i]
            m.running_mean[i] = None if rm is None else rm.data_ptr()
            m.running_var[i] = None if rv is None else rv.data_ptr()
            m.self_scale[i], m.momentum[i], m.eps[i] = float(plan.scales[i]), float(plan.momentum[i]), float(plan.eps[i])
            for l in range(plan.nl):
                k = i * plan.nl + l
                m.base_weight[k], m.spline_weight[k], m.spline_scaler[k] = ptrs[b + i * per + 2 + 3 * l: b + i * per + 5 + 3 * l]
        m.knots = plan.knots.data_ptr()
        m.num_readout, m.readout_grid_size, m.readout_spline_order = plan.n_readout, plan.ro_G, plan.ro_K
        m.readout_widths[0] = H
        for i in range(plan.n_readout):
            bw, sw, sc = params[c + 3 * i: c + 3 * i + 3]
            m.readout_widths[i + 1] = sw.size(0)
            m.readout_modes[i] = plan.ro_modes[i]
            m.readout_knots[i] = plan.ro_knots[i].data_ptr()
            m.readout_base_weight[i], m.readout_spline_weight[i] = bw.data_ptr(), sw.data_ptr()
            m.readout_spline_scaler[i] = None if sc is None else sc.data_ptr()
        hit = (key, bytes(m))                     # (plain bytes: the module stays deep-copyable / picklable; a stale copy fails the key)
        cache_owner.__dict__["_kagnn_model_call_template"] = hit
    return _lib.KaginModel.from_buffer_copy(hit[1])


class _KaginModelCallFn(Function):
    """``_KaginModelFn`` with its library calls folded into ONE per direction (``kagnn_kagin_model_fwd / _bwd``, include/kagnn_hip.h):
    the same entry points in the same order with the same arguments, sequenced inside the library -- same bits --, the activations
    the backward needs in ONE ``saved`` buffer, all parameter gradients in ONE flat buffer handed back as views.  What is left on the
    host per direction: two or three allocations, ~25 field assignments, one ctypes call."""

    @staticmethod
    @_on_operand_device
    def forward(ctx, x_int, e_int, g, seg, plan, owner, *params):
        from . import _lib
        params = tuple(None if t is None else t.contiguous() for t in params)
        xi, ei = x_int.contiguous(), e_int.contiguous()
        m = _model_struct(plan, params, owner)
        n, e, nb = xi.size(0), ei.size(0), seg.numel() - 1
        flags = edges = None
        if isinstance(g, torch.Tensor):                # the raw edge_index: the library builds this batch's CSR inside the call
            if g.size(1) != e:
                raise ValueError("edge_attr must have one row per edge of edge_index")
            edges = g if g.is_contiguous() else g.contiguous()
            flags = torch.empty(2, dtype=torch.int32, device=xi.device)
            m.edge_src, m.edge_dst, m.csr_flags = edges.data_ptr(), edges.data_ptr() + 8 * e, flags.data_ptr()
            _validate_pending(xi.device)
        else:
            if n != g.num_nodes or e != g.num_edges:
                raise ValueError("x must have one row per node and edge_attr one per edge of the graph")
            m.rowptr, m.col, m.perm = g.rowptr.data_ptr(), g.col.data_ptr(), g.perm.data_ptr()
            m.rowptr_t, m.col_t, m.perm_t = g.rowptr_t.data_ptr(), g.col_t.data_ptr(), g.perm_t.data_ptr()
        m.num_nodes, m.num_edges, m.num_graphs, m.x_stride, m.e_stride = n, e, nb, xi.size(1), ei.size(1)
        m.x_index, m.e_index = xi.data_ptr(), ei.data_ptr()
        m.seg_ptr = seg.data_ptr()
        skey = (n, e, nb, int(m.hidden), plan.n_atom, plan.n_bond, plan.nconv, plan.nl, plan.G, plan.K, plan.mode, plan.ro_G, plan.ro_K,
                tuple(plan.ro_modes), tuple(int(m.readout_widths[i]) for i in range(plan.n_readout + 1)),
                tuple(int(m.atom_rows[t]) for t in range(plan.n_atom)), tuple(int(m.bond_rows[t]) for t in range(plan.n_bond)),
                tuple(params[-3 * plan.n_readout + 3 * i + 2] is not None for i in range(plan.n_readout)), edges is not None)
        sizes = _MODEL_SIZES.get(skey)
        if sizes is None:
            outs = [ctypes.c_size_t(0) for _ in range(4)]
            _call("kagnn_kagin_model_sizes", ctypes.byref(m), *[ctypes.byref(o) for o in outs])
            if len(_MODEL_SIZES) > 512:
                _MODEL_SIZES.clear()
            sizes = _MODEL_SIZES[skey] = tuple(o.value for o in outs)
        dev = xi.device
        saved = torch.empty(sizes[0], dtype=torch.uint8, device=dev)
        ws = torch.empty(sizes[1], dtype=torch.uint8, device=dev)
        out = torch.empty((nb, int(m.readout_widths[plan.n_readout])), dtype=torch.float32, device=dev)
        m.saved, m.saved_bytes, m.workspace, m.workspace_bytes, m.out = saved.data_ptr(), sizes[0], ws.data_ptr(), sizes[1], out.data_ptr()
        _call("kagnn_kagin_model_fwd", ctypes.byref(m), _stream())
        if flags is not None:
            _defer_flag_check(flags)                   # (ids are clamped on the device; the flags are read without blocking later)
            ctx.save_for_backward(saved, xi, ei, seg, plan.knots, *plan.ro_knots, *params)
        else:
            ctx.save_for_backward(saved, xi, ei, seg, g.rowptr_t, g.col_t, g.perm_t, plan.knots, *plan.ro_knots, *params)
        ctx.model, ctx.sizes, ctx.graph = m, sizes, g
        ctx.shapes = [None if t is None else tuple(t.shape) for t in params]
        return out

    @staticmethod
    @once_differentiable
    @_on_operand_device
    def backward(ctx, gout):
        m, sizes = ctx.model, ctx.sizes
        _ = ctx.saved_tensors                       # (autograd's in-place checks on everything the library is about to read)
        gy = _rows(gout)
        dev = gy.device
        ws = torch.empty(sizes[2], dtype=torch.uint8, device=dev)
        flat = torch.empty(sizes[3], dtype=torch.float32, device=dev)
        m.workspace, m.workspace_bytes, m.g_out, m.ld_g_out, m.grads = ws.data_ptr(), sizes[2], gy.data_ptr(), _ld(gy), flat.data_ptr()
        _call("kagnn_kagin_model_bwd", ctypes.byref(m), _stream())
        grads, off = [], 0
        for shp in ctx.shapes:
            if shp is None:
                grads.append(None)
                continue
            cnt = 1
            for d in shp:
                cnt *= d
            grads.append(flat[off:off + cnt].view(shp))
            off += cnt
        return (None, None, None, None, None, None, *grads)


_GINE_MODEL_CALL = True      # False: _KaginModelFn's per-operation library calls (bit-identical; module attribute for the A/B test)
import os as _os
# True (KAGNN_MODEL_CSR=1): the library builds the batch's CSR inside kagnn_kagin_model_fwd, on a stream of its own beside the encoders and
# the weight packs (identical arrays).  Measured, NOT a gain: the two cross-stream waits cost the device about what the overlap hides
# (same-box A/B 0.71-0.77 ms per step without, 0.77-0.82 with; profiles/r06_experiments.md 2) -- opt-in.
_GINE_MODEL_CSR = _os.environ.get("KAGNN_MODEL_CSR", "0") == "1"


def ops_small_csr() -> bool:
    from . import ops
    return ops._SMALL_CSR


def ops_has_prefetched() -> bool:
    from . import ops
    return bool(ops._prefetched)


def _lib_csr_small_ok(e: int, n: int) -> bool:
    from . import _lib
    return bool(_lib.load().kagnn_csr_small_ok(e, n))

_GINE_MODEL_NODE = True      # False: the model runs as its five kinds of tape nodes (bit-identical; module attribute for the A/B test)


def _hook_free(mods) -> bool:
    mod = torch.nn.modules.module
    if (mod._global_forward_hooks or mod._global_forward_pre_hooks or mod._global_backward_hooks
            or getattr(mod, "_global_backward_pre_hooks", None)):
        return False
    for m in mods:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return False
    return True


def kagin_regression_forward(model, data):
    """``KAGINRegression.forward(data)`` as ONE tape node (``_KaginModelFn``), or ``None`` when the model or the batch is outside
    what the node covers (the caller then runs its modules one by one): training mode, no dropout, embedding-table encoders on
    int64 features, a GINE stack that ``_GineKanStackFn`` takes, a read-out ``KAN`` of uniform-grid ``KANLinear`` layers with at
    most 16 coefficients, no hooks anywhere, not under torch.compile."""
    if not _GINE_MODEL_NODE or not model.training or model.dropout.p > 0.0 or torch.compiler.is_compiling():
        return None
    x, e = data.x, data.edge_attr
    if e.dim() == 1:
        e = e.unsqueeze(1)
    atom, bond = model.atom_encoder, model.bond_encoder
    fast = getattr(atom, "_fast_tables", None), getattr(bond, "_fast_tables", None)
    if fast[0] is None or fast[1] is None:
        return None
    atabs, btabs = fast[0](x), fast[1](e)
    if atabs is None or btabs is None:
        return None
    convs, bns, ro = list(model.conv), list(model.bn), list(getattr(model.kan, "layers", []))
    if not ro or any(type(l).__name__ != "KANLinear" for l in ro) or type(model.kan).__name__ != "KAN":
        return None
    if any(type(c).__name__ != "GINEKANLayer" for c in convs) or any(type(b).__name__ != "BatchNorm1d" or not hasattr(b, "step") for b in bns):
        return None
    H = atabs[0].size(1)
    if btabs[0].size(1) != H or x.size(0) < 2:
        return None
    mods = [model, atom, bond, model.kan, *convs, *bns, *ro]
    for cv in convs:
        mods.append(cv.nn); mods += list(getattr(cv.nn, "layers", []))
    if not _hook_free(mods):
        return None
    plan_s = _gine_stack_plan(_ShapeOnly(x.size(0), H, x.device), convs, bns)
    if plan_s is None:
        return None
    first, nl, mode = plan_s
    G_r, K_r = ro[0].grid_size, ro[0].spline_order
    ro_knots, ro_modes = [], []
    for l in ro:
        k = l._knots()
        if k.dim() != 1 or l.grid_size != G_r or l.spline_order != K_r or G_r + K_r > 16:
            return None
        m = l.precision if l.precision is not None else default_precision()
        if split_like(m) and max(l.in_features, l.out_features) > 7680:
            m = PREC_FP32
        ro_knots.append(k); ro_modes.append(int(m))
    if ro[0].in_features != H:
        return None
    from .graph_models import _segment_ptr
    ei_raw = data.edge_index
    in_call = (_GINE_MODEL_CALL and _GINE_MODEL_CSR and H <= 64 and len(atabs) <= 16 and len(btabs) <= 16 and len(ro) <= 8
               and torch.is_tensor(ei_raw) and ei_raw.is_cuda and ei_raw.dtype == torch.int64 and ei_raw.dim() == 2 and ei_raw.size(0) == 2
               and ops_small_csr() and _lib_csr_small_ok(int(ei_raw.size(1)), int(x.size(0))) and not ops_has_prefetched())
    # (in_call: the library builds this batch's CSR itself, on its own stream beside the encoders and the weight packs)
    g = ei_raw if in_call else graph_index(ei_raw, x.size(0), cache=False)
    seg = _segment_ptr(data)
    sparams, scales, running, momentum, eps = _gine_stack_args(convs, bns)
    plan = _ModelPlan()
    plan.n_atom, plan.n_bond, plan.n_stack, plan.n_readout = len(atabs), len(btabs), len(sparams), len(ro)
    plan.scales, plan.running, plan.momentum, plan.eps, plan.nconv, plan.nl = scales, running, momentum, eps, len(convs), nl
    plan.knots, plan.G, plan.K, plan.mode = first._knots(), first.grid_size, first.spline_order, mode
    plan.ro_knots, plan.ro_G, plan.ro_K, plan.ro_modes = ro_knots, G_r, K_r, ro_modes
    rparams = []
    for l in ro:
        rparams += [l.base_weight, l.spline_weight, l.spline_scaler if l.enable_standalone_scale_spline else None]
    if _GINE_MODEL_CALL and H <= 64 and len(atabs) <= 16 and len(btabs) <= 16 and len(ro) <= 8:
        return _KaginModelCallFn.apply(x, e, g, seg, plan, model, *atabs, *btabs, *sparams, *rparams)
    return _KaginModelFn.apply(x, e, g, seg, plan, *atabs, *btabs, *sparams, *rparams)


class _ShapeOnly:
    """what ``_gine_stack_plan`` asks of its input before the input exists (the atom encoder's output: fp32 [N, H] on the device)"""
    __slots__ = ("_n", "_h", "device")
    is_cuda, dtype = True, torch.float32

    def __init__(self, n, h, device):
        self._n, self._h, self.device = n, h, device

    def size(self, d):
        return (self._n, self._h)[d]

