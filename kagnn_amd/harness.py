"""Timing harness: the working counterpart of the reference's ``node_classification_clean/time_model.py``.

``time_model`` reproduces the reference loop :35-48 -- Adam(lr=1e-3), 20 x {zero_grad, forward, softmax,
CrossEntropyLoss on the masked nodes, backward, step} -- including its softmax-before-CE quirk, and fixes
what makes the original unusable as a benchmark: it warms up, synchronises the device
around the timed region, and forwards ``grid_size`` to the model.  Returns seconds per epoch.

``graphed=True`` captures one whole epoch (forward, loss, backward, Adam) into a HIP graph after the
warm-up and replays it: full-batch node classification has static shapes, and on Cora-sized graphs the
epoch is launch-bound (~150 kernels of a few microseconds each), so replaying removes the host from the
loop.  The arithmetic is the same kernels in the same order; the masked rows are selected through a
precomputed index instead of boolean indexing (which would synchronise inside the capture).
"""
from __future__ import annotations

import time

import numpy as np
import torch


def _time_model_graphed(model, x, edge_index, y, mask, nb_epochs: int, warmup: int):
    from . import ops
    try:
        optimizer = torch.optim.Adam(model.parameters(), lr=0.001, capturable=True, fused=True)
    except (TypeError, RuntimeError):
        optimizer = torch.optim.Adam(model.parameters(), lr=0.001, capturable=True)
    if mask.dtype != torch.bool:
        mask = torch.zeros(x.size(0), dtype=torch.bool, device=x.device).index_fill_(0, mask, True)
    if isinstance(edge_index, ops.GraphIndex) or (isinstance(edge_index, torch.Tensor) and edge_index.is_sparse):
        graph_index = edge_index           # sparse adjacency (gcn timing branch): the convs index it, cached
    else:
        graph_index = ops.graph_index(edge_index, x.size(0))

    def epoch():
        optimizer.zero_grad(set_to_none=True)
        loss = ops.softmax_cross_entropy(model(x, graph_index), y, mask, pre_softmax=True)
        loss.backward()
        optimizer.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                    # warm-up off the default stream, as graph capture requires
        for _ in range(max(warmup, 3)):
            epoch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = epoch()
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nb_epochs):
        graph.replay()
        losses.append(static_loss.detach().clone())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nb_epochs
    return float(np.round(dt, 6)), [float(l) for l in losses]


def time_model(model, x, edge_index, y, mask, nb_epochs: int = 20, warmup: int = 2, graphed: bool = False):
    if graphed:
        return _time_model_graphed(model, x, edge_index, y, mask, nb_epochs, warmup)
    from . import ops
    # (the reference's optimiser, time_model.py:36; on the device its update runs as ONE fused launch over all parameter tensors
    # instead of ~10 multi-tensor launches: same update rule)
    try:
        import os
        optimizer = torch.optim.Adam(model.parameters(), lr=0.001, fused=bool(x.is_cuda) and os.environ.get("KAGNN_FUSED_ADAM", "1") != "0")
    except (TypeError, RuntimeError):
        optimizer = torch.optim.Adam(model.parameters(), lr=0.001)
    losses = []

    def epoch():
        optimizer.zero_grad()
        out = model(x, edge_index)
        # the reference applies softmax before CrossEntropyLoss (:43-44): softmax, masked gather, log_softmax and the
        # mean are one kernel each way (kagnn_softmax_xent_*)
        loss = ops.softmax_cross_entropy(out, y, mask, pre_softmax=True)
        loss.backward()
        optimizer.step()
        return loss

    # (round 6) the backward on the calling thread: on small graphs an epoch is ~1 ms of ~100 short kernels and handing its tape nodes
    # to autograd's per-device worker thread costs more than running them (see train_graph_batches; KAGNN_CFG4_MT=1 for the A/B)
    import os
    mt_was = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(os.environ.get("KAGNN_CFG4_MT", "0") == "1")
    try:
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
    finally:
        torch.autograd.set_multithreading_enabled(mt_was)
    return float(np.round(dt, 6)), [float(l.detach()) for l in losses]


class Adam:
    """``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (no amsgrad) for fp32 device parameters as ONE library call per
    step (``kagnn_adam_step``: one launch per 64 tensors) -- the optimiser of the reference's graph-regression scripts
    (``graph_regression/optuna_zinc.py:49,62``).  Same update rule in fp32; what it removes is torch.optim's per-step Python (state
    dictionaries, tensor grouping, step counters kept as tensors: 0.2-0.3 ms of host time per step, fused or not -- a quarter of a
    256-molecule mini-batch's step).  ``step()`` / ``zero_grad()`` / ``state_dict()``-free by design: a training-loop helper, not a
    torch.optim subclass (schedulers and checkpoints want the real one: pass it to ``train_graph_batches(optimizer=...)``).
    As in torch.optim.Adam the step count of the bias correction is PER PARAMETER TENSOR and advances only when that tensor has a
    gradient (a global counter gave a tensor that skips steps another correction than the reference's optimiser);
    ``amsgrad`` / ``maximize`` are not implemented and rejected."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False,
                 maximize: bool = False):
        import ctypes
        if amsgrad or maximize:
            raise NotImplementedError("kagnn_amd.harness.Adam implements torch.optim.Adam's default rule only (no amsgrad / maximize); "
                                      "pass a torch.optim.Adam to train_graph_batches(optimizer=...) for those")
        self.params = [p for p in params if p.requires_grad]
        if not self.params or any((not p.is_cuda) or p.dtype != torch.float32 or not p.is_contiguous() for p in self.params):
            raise TypeError("kagnn_amd.harness.Adam takes contiguous fp32 parameters on the GPU (there is no CPU path)")
        if len({p.device for p in self.params}) != 1:
            raise ValueError("kagnn_amd.harness.Adam: all parameters on one device")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.steps = 0                                               # steps taken by the optimiser
        self._tensor_steps = None                                    # per-tensor counts; None while every tensor has taken every step
        total = sum(p.numel() for p in self.params)
        self._m = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        self._v = torch.zeros_like(self._m)
        off, moff = 0, []
        for p in self.params:
            moff.append(off); off += p.numel()
        n = len(self.params)
        self._VP, self._I64 = ctypes.c_void_p * n, ctypes.c_int64 * n
        self._m_ptr = [self._m.data_ptr() + 4 * o for o in moff]
        self._v_ptr = [self._v.data_ptr() + 4 * o for o in moff]
        self._numel = [p.numel() for p in self.params]
        self._p_ptr = [p.data_ptr() for p in self.params]            # (parameters are updated in place: their storage stays)
        self._p_tab, self._m_tab, self._v_tab = self._VP(*self._p_ptr), self._VP(*self._m_ptr), self._VP(*self._v_ptr)
        self._n_tab = self._I64(*self._numel)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        from . import ops
        self.steps += 1
        f32 = torch.float32
        ptrs = [p.data_ptr() for p in self.params]
        if ptrs != self._p_ptr:                                       # a parameter's storage was replaced (p.data = ...): follow it
            self._p_ptr, self._p_tab = ptrs, self._VP(*ptrs)
        gs = [p.grad for p in self.params]
        if not all(g is not None and g.dtype is f32 and g.is_contiguous() for g in gs):     # rare: a subset, or gradients to convert
            # (`None in gs` would compare every TENSOR with None through torch's dispatcher: 10 us each)
            keep = [k for k, g in enumerate(gs) if g is not None]
            if not keep:
                self.steps -= 1                                       # (no gradient anywhere: not a step, as in torch.optim.Adam)
                return
            if len(keep) < len(gs) and self._tensor_steps is None:    # first time a tensor sits a step out: counts diverge from here
                self._tensor_steps = [self.steps - 1] * len(gs)
            gs = {k: (gs[k] if gs[k].dtype is f32 and gs[k].is_contiguous() else gs[k].to(f32).contiguous()) for k in keep}
            groups = {}
            if self._tensor_steps is None:
                groups[self.steps] = keep
            else:
                for k in keep:
                    self._tensor_steps[k] += 1
                    groups.setdefault(self._tensor_steps[k], []).append(k)
            for t, ks in groups.items():                              # one library call per distinct step count (normally one)
                VP, I64 = _ctypes_arrays(len(ks))
                with ops._device_of(self.params[0]):
                    ops._call("kagnn_adam_step", len(ks), VP(*[self._p_ptr[k] for k in ks]), VP(*[gs[k].data_ptr() for k in ks]),
                              VP(*[self._m_ptr[k] for k in ks]), VP(*[self._v_ptr[k] for k in ks]), I64(*[self._numel[k] for k in ks]),
                              self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, t, ops._stream())
            return
        if self._tensor_steps is not None:                            # all tensors present, but their counts differ
            groups = {}
            for k in range(len(gs)):
                self._tensor_steps[k] += 1
                groups.setdefault(self._tensor_steps[k], []).append(k)
            if len(groups) > 1:
                for t, ks in groups.items():
                    VP, I64 = _ctypes_arrays(len(ks))
                    with ops._device_of(self.params[0]):
                        ops._call("kagnn_adam_step", len(ks), VP(*[self._p_ptr[k] for k in ks]), VP(*[gs[k].data_ptr() for k in ks]),
                                  VP(*[self._m_ptr[k] for k in ks]), VP(*[self._v_ptr[k] for k in ks]), I64(*[self._numel[k] for k in ks]),
                                  self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, t, ops._stream())
                return
            step_t = next(iter(groups))
        else:
            step_t = self.steps
        tables = (self._p_tab, self._VP(*[g.data_ptr() for g in gs]), self._m_tab, self._v_tab, self._n_tab)
        with ops._device_of(self.params[0]):
            ops._call("kagnn_adam_step", len(gs), *tables, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, step_t,
                      ops._stream())


def _ctypes_arrays(n: int):
    import ctypes
    return ctypes.c_void_p * n, ctypes.c_int64 * n


class GradientReplicas:
    """Data-parallel replicas of a mini-batch training loop ("replicas only ... a gradient all-reduce of <= a few
    MB" -- BASELINE config 4: every rank trains the same model on its own mini-batches; the reference loop is
    ``graph_regression/optuna_zinc.py:56-66`` on one device).  ONE flat all-reduce per step, between ``backward()`` and
    ``optimizer.step()``: the ranks' gradients are concatenated into one buffer, summed (RCCL), and ``p.grad`` of every parameter
    becomes a view of the result.  The 1/P of the mean over the global batch is folded into the upstream gradient of the loss
    (``scale()``), so the all-reduce is a plain sum and no extra pass touches the gradients.  Parameters are broadcast from
    rank 0 at construction; equal updates keep them equal (the optimiser state is replicated, not communicated)."""

    def __init__(self, params, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.params = [p for p in params if p.requires_grad]
        for p in self.params:
            dist.broadcast(p.data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._flat = None
        self._scale = None

    def scale(self, like: torch.Tensor) -> torch.Tensor:
        """the upstream gradient ``1 / world`` of this rank's loss (``loss.backward(replicas.scale(loss))``)"""
        if self._scale is None or self._scale.device != like.device or self._scale.dtype != like.dtype:
            self._scale = torch.full((), 1.0 / self.world, dtype=like.dtype, device=like.device)
        return self._scale

    def sync(self) -> None:
        """sum the gradients of this step over the ranks: one ``torch.cat``, one all-reduce, views back"""
        import torch.distributed as dist
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in live])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in live:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n


def train_graph_batches(model, batches, nb_epochs: int = 1, warmup: int = 0, lr: float = 1e-3, optimizer=None, group=None,
                        loss_fn=None):
    """The mini-batch training loop of the reference's graph-regression scripts (``graph_regression/optuna_zinc.py:56-66``: Adam,
    L1 loss, ``{zero_grad, loss(model(data).squeeze(), data.y), backward, step}`` per batch) over ``batches`` -- objects with
    ``x, edge_index, edge_attr, batch, y`` (``num_graphs`` / ``ptr`` when the loader supplies them) already on the device.
    Returns ``(seconds per step, mean training loss per epoch)``.  Differences from the script, none of them in the mathematics:
    * the optimiser is ``kagnn_amd.harness.Adam`` unless one is passed (torch.optim.Adam's rule in fp32 as one library call;
      torch's own kernels round differently -- its fused one mixes double arithmetic -- so trajectories agree to rounding, not
      to the bit; with ``optimizer=torch.optim.Adam(..., fused=True)`` the loop IS the script's, bit for bit);
    * the loss is ``ops.l1_loss`` (the same mean absolute error, the same gradient bits; its forward sum runs in another order);
    * the running loss is accumulated ON THE DEVICE and read once per epoch -- the script's ``loss.item()`` per batch drains the
      stream every step, which on a step of ~1 ms of device work is the difference between the host running ahead of the GPU
      and waiting for it.
    ``group`` (a torch.distributed process group, or ``True`` for the default one): data-parallel REPLICAS -- ``batches`` are this
    rank's own; every step ends in ONE flat gradient all-reduce (``GradientReplicas``), i.e. the step of the script on the
    P-times larger batch that the ranks' batches form together (except that BatchNorm statistics stay per replica, as in
    torch's DistributedDataParallel).  The returned loss is this rank's."""
    replicas = None
    if group is not None:
        replicas = GradientReplicas(model.parameters(), None if group is True else group)
    if optimizer is None:
        optimizer = Adam(model.parameters(), lr=lr)            # torch.optim.Adam's rule, one library call per step
    from . import ops as ops_mod
    if loss_fn is None:
        loss_fn = ops_mod.l1_loss      # = torch.nn.L1Loss() (mean |p - t|), one launch each way instead of six
    model.train()
    on_gpu = any(p.is_cuda for p in model.parameters())

    import os
    # (opt-in: on the round-6 boxes the step is HOST-bound, and the prefetch's stream switch + events cost the host more than the
    # 55 us of device time they hide -- same-box A/B in profiles/r06_experiments.md; a device-bound loop can turn it on)
    prefetch = on_gpu and os.environ.get("KAGNN_PREFETCH_CSR", "0") == "1"

    root = []

    def one(like):
        if not root or root[0].device != like.device or root[0].dtype != like.dtype or root[0].shape != like.shape:
            root[:] = [torch.ones(like.shape, dtype=like.dtype, device=like.device)]
        return root[0]

    def lookahead(it):
        it = iter(it)
        try:
            cur = next(it)
        except StopIteration:
            return
        for nxt in it:
            yield cur, nxt
            cur = nxt
        yield cur, None

    def epoch():
        losses, weights = [], []
        for data, upcoming in lookahead(batches):
            if prefetch and upcoming is not None and hasattr(upcoming, "edge_index") and torch.is_tensor(getattr(upcoming, "x", None)):
                # the NEXT batch's CSR is built on a side stream beside this step's kernels (a loader worker's job; ops.prefetch_graph_index)
                ops_mod.prefetch_graph_index(upcoming.edge_index, upcoming.x.size(0))
            optimizer.zero_grad(set_to_none=True)
            loss = loss_fn(model(data).squeeze(), data.y.squeeze())
            if replicas is None:
                loss.backward(one(loss))           # (the implicit root gradient would be a torch.ones_like: one fill launch per step)
            else:
                loss.backward(replicas.scale(loss))
                replicas.sync()
            optimizer.step()
            losses.append(loss.detach())           # (kept on the device: no kernel and no read-back per step)
            weights.append(float(int(getattr(data, "num_graphs", 0) or data.y.size(0))))
        if on_gpu:
            ops_mod.flush_graph_checks()       # the epoch's deferred node-id range checks (incl. the LAST batch's): one wait per epoch
        if not losses:
            return torch.zeros(())
        # the epoch's mean training loss, weighted by graphs per batch: three small launches per EPOCH, still no read-back
        # (the caller converts after the final synchronisation)
        w = torch.tensor(weights, dtype=losses[0].dtype).to(losses[0].device, non_blocking=True)
        return (torch.stack(losses) * w).sum() / max(sum(weights), 1.0)

    # The backward of a ~1 ms step is ~80 kernel launches from ~10 library calls: handing every tape node to autograd's per-device
    # worker thread costs more than running it (host issue time of the ZINC-shaped step on the round-6 boxes: 1.41-1.50 ms per
    # step with the engine's threads, 0.85-0.90 ms on the calling thread -- tools/host_profile_cfg4.py, profiles/r06_experiments.md).
    # One device, one stream: nothing runs concurrently in that backward anyway.
    mt_was = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(os.environ.get("KAGNN_CFG4_MT", "0") == "1")     # (=1: the engine's worker threads, for A/B)
    try:
        for _ in range(warmup):
            epoch()
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        means = [epoch() for _ in range(nb_epochs)]
        if on_gpu:
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(1, nb_epochs * len(batches))
    finally:
        torch.autograd.set_multithreading_enabled(mt_was)
    return float(dt), [float(m) for m in means]


def count_params(model) -> int:
    return int(sum(p.numel() for p in model.parameters()))


def make_model(params: dict):
    """``utils.make_model`` of the reference (``node_classification_clean/utils.py:88-123``) for the KAN / FastKAN
    architectures: the same ``params`` dictionary builds the same model classes.  The MLP baselines
    (``architecture == 'mlp'``) are stock torch_geometric models and outside this package."""
    from .models import GFASTKAN_Nodes, GKAN_Nodes
    common = dict(conv_type=params["conv_type"], mp_layers=params["mp_layers"], num_features=params["num_features"],
                  hidden_channels=params["hidden_channels"], num_classes=params["num_classes"], skip=params["skip"],
                  hidden_layers=params["hidden_layers"], dropout=params["dropout"], grid_size=params["grid_size"],
                  heads=params.get("heads", 4))
    if params["architecture"] == "kan":
        return GKAN_Nodes(spline_order=params["spline_order"], **common)
    if params["architecture"] == "fastkan":
        return GFASTKAN_Nodes(**common)
    raise ValueError("kagnn_amd.harness.make_model builds the 'kan' and 'fastkan' architectures")
