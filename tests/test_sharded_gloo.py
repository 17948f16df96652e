"""world_size-2 / -4 / -8 gloo tests (CPU) of the feature-sharded layer's communication logic.

The local compute is INJECTED here (oracle ops, differentiable through stock autograd) -- the product
default is the HIP library; what is under test is the sharding / all-reduce / all-gather algebra:
the sharded forward and every gradient must equal the unsharded oracle."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """oracle-backed stand-ins for kagnn_amd.ops.{aggregate_sum,kan_linear} (test infrastructure)."""

    @staticmethod
    def aggregate_sum(x, graph, self_scale=1.0):
        from oracle import kan_oracle as orc
        return orc.sum_aggregate(x, graph) + self_scale * x          # `graph` is the raw edge_index here

    @staticmethod
    def kan_linear(x, bw, sw, sc, knots, G, K, mode=None):
        from oracle import kan_oracle as orc
        return orc.kan_linear_forward(x, bw, sw, sc, knots.expand(x.size(1), -1), K)

    # ---- the local halves of the feature-sharded FastKAN layer (kagnn_amd.ops.fastkan_row_moments ... _shard_bwd_finish),
    # restated with torch ops in fp64: what is under test is the exchange algebra of kagnn_amd.sharded around them
    BatchNorm1d = torch.nn.BatchNorm1d

    @staticmethod
    def fastkan_row_moments(x):
        xd = x.double()
        mean = xd.mean(1)
        return torch.stack([mean, ((xd - mean[:, None]) ** 2).sum(1)], 1).to(x.dtype)

    @staticmethod
    def fastkan_merge_moments(gathered, width, eps):
        g = gathered.double()
        mean, m2 = g[0, :, 0].clone(), g[0, :, 1].clone()
        for p in range(1, g.size(0)):                       # Chan's update, rank order
            na, nb = float(p * width), float(width)
            d = g[p, :, 0] - mean
            mean = mean + d * (nb / (na + nb))
            m2 = m2 + g[p, :, 1] + d * d * (na * nb / (na + nb))
        return torch.stack([mean, torch.rsqrt(m2 / (g.size(0) * width) + eps)], 1).to(gathered.dtype)

    @staticmethod
    def _fk_partial(z, x, sw, bw, bb, centers, den):
        from oracle import kan_oracle as orc
        phi = orc.rbf_bases(z, centers.double(), den)
        y = torch.nn.functional.linear(phi.view(z.size(0), -1), sw.double())
        if bw is not None:
            y = y + torch.nn.functional.linear(torch.nn.functional.silu(x), bw.double(), None if bb is None else bb.double())
        return y

    @staticmethod
    def fastkan_shard_fwd(x, stats, ln_w, ln_b, sw, bw, bb, centers, den, mode=None):
        xd = x.double()
        with torch.no_grad():
            z = xd if ln_w is None else (xd - stats[:, :1].double()) * stats[:, 1:].double() * ln_w.double() + ln_b.double()
            return OracleOps._fk_partial(z, xd, sw, bw, bb, centers, den).to(x.dtype)

    @staticmethod
    def fastkan_shard_bwd(x, gy, stats, ln_w, ln_b, sw, bw, centers, den, mode=None, want_bias=False):
        xd = x.detach().double()
        zhat = None if ln_w is None else (xd - stats[:, :1].double()) * stats[:, 1:].double()
        with torch.enable_grad():
            xb = xd.clone().requires_grad_(True)                     # the base branch's (and, without LayerNorm, the RBF's) input
            z = xb if ln_w is None else (zhat * ln_w.detach().double() + ln_b.detach().double()).requires_grad_(True)
            swd = sw.detach().double().requires_grad_(True)
            bwd = None if bw is None else bw.detach().double().requires_grad_(True)
            bbd = torch.zeros(sw.size(0), dtype=torch.float64, requires_grad=True) if (want_bias and bw is not None) else None
            y = OracleOps._fk_partial(z, xb, swd, bwd, bbd, centers, den)
            y.backward(gy.double())
        st = dict(gx=xb.grad, gz=None if ln_w is None else z.grad, zhat=zhat, rstd=None if ln_w is None else stats[:, 1:].double(),
                  lw=ln_w, dtype=x.dtype)
        sums = None
        if ln_w is not None:
            h = z.grad * ln_w.detach().double()
            sums = torch.stack([h.sum(1), (h * zhat).sum(1)], 1).to(x.dtype)
        f = x.dtype
        return st, sums, swd.grad.to(f), None if bwd is None else bwd.grad.to(f), None if bbd is None else bbd.grad.to(f)

    @staticmethod
    def fastkan_shard_bwd_finish(st, sums, width_total):
        f = st["dtype"]
        if st["lw"] is None:
            return st["gx"].to(f), None, None
        h = st["gz"] * st["lw"].detach().double()
        s = sums.double() / width_total
        gx = st["gx"] + st["rstd"] * (h - s[:, :1] - st["zhat"] * s[:, 1:])
        return gx.to(f), (st["gz"] * st["zhat"]).sum(0).to(f), st["gz"].sum(0).to(f)


def _worker(rank, world, port, out_dir, f=8, hid=12, grid=5, full=True):
    """``f`` / ``hid`` / ``grid``: layer widths and grid size -- world 8 at 64 / 64 / 5 and 128 / 128 / 8 gives every rank the 8 and 16
    columns of the headline's and config 3's P = 8 shards (the narrow instantiations' shapes; SURVEY 8(e), BASELINE config 3);
    ``full``: also the regression sections that do not depend on the world size (run at world 2 only)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kagnn_amd
        from kagnn_amd.sharded import ShardedGIKANLayer
        from oracle import kan_oracle as orc
        n, e = 400, 3000
        ei = orc.powerlaw_graph(n, e, seed=11)
        gen = torch.Generator().manual_seed(11)
        x = torch.randn(n, f, generator=gen) * 0.3
        gy = torch.randn(n, f, generator=gen)
        torch.manual_seed(5)                                  # same full module on every rank
        conv = kagnn_amd.GIKANLayer(f, f, grid_size=grid, spline_order=3, hidden_dim=hid, nb_layers=2)
        layers = [{k: v.detach().clone() for k, v in l.state_dict().items()} for l in conv.nn.layers]
        y_ref, gx_ref, g_ref = orc.kan_gin_layer_fwd_bwd(x, ei, layers, 3, gy)

        assert f % world == 0 and hid % world == 0
        w = f // world
        sl = slice(rank * w, (rank + 1) * w)
        tol = 2e-5
        for chunks in (1, 3, 7):                              # 3 / 7: overlapped row chunks (uneven: 400 = 134+134+132, 58*6+52)
            sconv = ShardedGIKANLayer(conv, None, local_ops=OracleOps, chunks=chunks)
            xs = sconv.shard_columns(x).requires_grad_(True)
            y = sconv(xs, ei)
            y.backward(sconv.shard_columns(gy))
            assert y.shape == (n, w) and xs.grad.shape == (n, w)
            assert torch.allclose(y, y_ref[:, sl], atol=tol, rtol=tol)
            assert torch.allclose(xs.grad, gx_ref[:, sl], atol=tol, rtol=tol)
            for li, layer in enumerate(sconv.layers):
                assert layer.hi - layer.lo == (f if li == 0 else hid) // world       # this rank's slice of the input features
                isl = slice(layer.lo, layer.hi)
                assert torch.allclose(layer.base_weight.grad, g_ref[li]["base_weight"][:, isl], atol=tol, rtol=tol)
                assert torch.allclose(layer.spline_weight.grad, g_ref[li]["spline_weight"][:, isl], atol=tol, rtol=tol)
                assert torch.allclose(layer.spline_scaler.grad, g_ref[li]["spline_scaler"][:, isl], atol=tol, rtol=tol)
        # ---- the transposed variant: column-sharded aggregation, all-to-all, row-sharded KAN chain, all-to-all
        from kagnn_amd.sharded import TransposedShardedGIKANLayer
        for n2, sync in ((400, True), (401, False), (401, "flat")):   # per-parameter / explicit flat / queued flat all-reduce
            ei2 = orc.powerlaw_graph(n2, e, seed=12)
            x2 = torch.randn(n2, f, generator=gen) * 0.3
            gy2 = torch.randn(n2, f, generator=gen)
            y_ref, gx_ref, g_ref = orc.kan_gin_layer_fwd_bwd(x2, ei2, layers, 3, gy2)
            tconv = TransposedShardedGIKANLayer(conv, None, local_ops=OracleOps, sync_in_backward=sync)
            xs = tconv.shard_columns(x2).requires_grad_(True)
            y = tconv(xs, ei2)
            y.backward(tconv.shard_columns(gy2))
            if sync is False:
                tconv.sync_gradients()                        # the explicit single flat all-reduce
            assert torch.allclose(y, y_ref[:, sl], atol=tol, rtol=tol)
            assert torch.allclose(xs.grad, gx_ref[:, sl], atol=tol, rtol=tol)
            for li, layer in enumerate(tconv.layers):         # replicated parameters: full gradients on every rank
                for name in ("base_weight", "spline_weight", "spline_scaler"):
                    assert torch.allclose(getattr(layer, name).grad, g_ref[li][name], atol=tol, rtol=tol), (n2, li, name)
        if not full:
            open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
            return
        # ---- "flat" with gradient accumulation (two backward passes, no zero_grad) and the module used twice in one
        # forward: only what a pass adds may be summed over the ranks (the queued callback used to all-reduce
        # the accumulated .grad again: P*S1 + S2)
        tconv = TransposedShardedGIKANLayer(conv, None, local_ops=OracleOps, sync_in_backward="flat")
        xs = tconv.shard_columns(x2).requires_grad_(True)
        for _ in range(2):
            tconv(xs, ei2).backward(tconv.shard_columns(gy2))
        for li, layer in enumerate(tconv.layers):
            for name in ("base_weight", "spline_weight", "spline_scaler"):
                assert torch.allclose(getattr(layer, name).grad, 2.0 * g_ref[li][name], atol=2 * tol, rtol=tol), ("accumulate", li, name)
        tconv.zero_grad()
        (tconv(xs, ei2) * tconv.shard_columns(gy2)).sum().add((tconv(xs, ei2) * tconv.shard_columns(gy2)).sum()).backward()
        for li, layer in enumerate(tconv.layers):
            for name in ("base_weight", "spline_weight", "spline_scaler"):
                assert torch.allclose(getattr(layer, name).grad, 2.0 * g_ref[li][name], atol=2 * tol, rtol=tol), ("used twice", li, name)
        # ----  a backward pass that RAISES after the flat-sync latch was armed: the engine drops the queued
        # callback, and the latch must not survive into the next pass (every later backward used to skip the all-reduce)
        tconv = TransposedShardedGIKANLayer(conv, None, local_ops=OracleOps, sync_in_backward="flat")
        xs = tconv.shard_columns(x2).requires_grad_(True)

        def boom(_g):
            raise RuntimeError("boom")
        hook = xs.register_hook(boom)                        # fires at the very end of the pass, on both ranks alike
        with pytest.raises(RuntimeError, match="boom"):
            tconv(xs, ei2).backward(tconv.shard_columns(gy2))
        hook.remove()
        assert tconv._flat_pending is not None               # (the failed pass left its latch behind ...)
        tconv.zero_grad()
        xs.grad = None
        tconv(xs, ei2).backward(tconv.shard_columns(gy2))   # (... and the next pass must re-arm and synchronise)
        assert tconv._flat_pending is None
        for li, layer in enumerate(tconv.layers):
            for name in ("base_weight", "spline_weight", "spline_scaler"):
                assert torch.allclose(getattr(layer, name).grad, g_ref[li][name], atol=tol, rtol=tol), ("after a failed pass", li, name)
        # ---- comm="p2p" states its preconditions at construction, not in the middle of a forward
        with pytest.raises(ValueError, match="p2p"):
            ShardedGIKANLayer(conv, None, local_ops=OracleOps, comm="p2p")
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _worker_e2(rank, world, port, out_dir, f=16, hid=24, classes=10):
    """SURVEY 8(e)'s other exchanges: the feature-sharded FastKAN-GIN layer (LayerNorm statistics: 2 floats
    per row each way), and whole node models on column shards (shard-local BatchNorm1d, input-sharded skip read-out closed by
    one all-reduce of the [N, classes] partial sums) -- against the unsharded oracle, every gradient."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kagnn_amd
        import helpers
        from kagnn_amd.sharded import ShardedGIFASTKANLayer, ShardedNodeModel
        from oracle import kan_oracle as orc
        n, e = 300, 2500
        ei = orc.powerlaw_graph(n, e, seed=21)
        gen = torch.Generator().manual_seed(21)
        x = torch.randn(n, f, generator=gen) * 0.3 + 0.7          # (a non-zero row mean: the statistics merge has something to do)
        gy = torch.randn(n, f, generator=gen)
        tol = 2e-5
        w = f // world
        sl = slice(rank * w, (rank + 1) * w)

        def close(a, b, what, t=tol):
            err = float((a.double() - b.double()).abs().max()) / max(1e-30, float(b.abs().max()))
            assert err <= t, (what, err)

        # ---- (a) ShardedGIFASTKANLayer
        torch.manual_seed(6)
        conv = kagnn_amd.GIFASTKANLayer(f, f, grid_size=5, hidden_dim=hid, nb_layers=2)
        with torch.no_grad():                                     # (default init: LayerNorm weight 1 / bias 0, base bias ~0: make them matter)
            for l in conv.nn.layers:
                l.layernorm.weight.uniform_(0.5, 1.5); l.layernorm.bias.uniform_(-0.3, 0.3); l.base_linear.bias.uniform_(-0.5, 0.5)
        layers = [{k: v.detach().clone().double() for k, v in l.state_dict().items()} for l in conv.nn.layers]
        leaves = [{k: (v.requires_grad_(True) if k != "rbf.grid" else v) for k, v in p.items()} for p in layers]
        xr = x.double().requires_grad_(True)
        y_ref = orc.gin_conv(xr, ei, lambda h: orc.fastkan_forward(h, leaves))
        y_ref.backward(gy.double())
        for chunks in (1, 3):
            sconv = ShardedGIFASTKANLayer(conv, None, local_ops=OracleOps, chunks=chunks)
            xs = sconv.shard_columns(x).requires_grad_(True)
            y = sconv(xs, ei)
            y.backward(sconv.shard_columns(gy))
            close(y, y_ref[:, sl], "fastkan y")
            close(xs.grad, xr.grad[:, sl], "fastkan gx")
            for li, (layer, p) in enumerate(zip(sconv.layers, leaves)):
                cols = layer.columns
                ng = layer.centers.numel()
                fin, fout = p["base_linear.weight"].size(1), p["base_linear.weight"].size(0)
                close(layer.ln_weight.grad, p["layernorm.weight"].grad[cols], f"L{li} g_ln_weight")
                close(layer.ln_bias.grad, p["layernorm.bias"].grad[cols], f"L{li} g_ln_bias")
                close(layer.spline_weight.grad, p["spline_linear.weight"].grad.view(fout, fin, ng)[:, cols].reshape(fout, -1), f"L{li} g_spline")
                close(layer.base_weight.grad, p["base_linear.weight"].grad[:, cols], f"L{li} g_base_weight")
                assert (layer.base_bias is not None) == (rank == 0)
                if rank == 0:
                    close(layer.base_bias.grad, p["base_linear.bias"].grad, f"L{li} g_base_bias")

        # ---- (b) whole node models on column shards: GKAN_Nodes and GFASTKAN_Nodes, gin, skip read-out
        labels = torch.randint(0, classes, (n,), generator=gen)
        for arch in ("kan", "fastkan"):
            torch.manual_seed(7)
            if arch == "kan":
                model = kagnn_amd.GKAN_Nodes("gin", 2, f, hid, classes, grid_size=4, spline_order=3, hidden_layers=2)
            else:
                model = kagnn_amd.GFASTKAN_Nodes("gin", 2, f, hid, classes, grid_size=4, hidden_layers=2)
            with torch.no_grad():
                for bn in model.bns:
                    bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
            state = {k: v.detach().clone() for k, v in model.state_dict().items()}
            # the loss every rank evaluates on the full logits (harness loss: softmax -> CrossEntropy, time_model.py:43-44)
            logits_ref = None

            def loss_grad(lg):
                lg = lg.detach().double().requires_grad_(True)
                orc.harness_loss(lg, labels).backward()
                return lg.grad
            xr = x.double()
            out_ref, gx_ref, g_ref = None, None, None
            lg0 = orc.node_model_forward(xr, ei, {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}, arch, "gin", 2)
            gout = loss_grad(lg0)
            out_ref, gx_ref, g_ref = helpers.oracle_node_model_fwd_bwd(x, ei, state, gout, arch, "gin", 2)
            sm = ShardedNodeModel(model, None, local_ops=OracleOps, chunks=2)
            sm.train()
            xs = sm.shard_columns(x).requires_grad_(True)
            logits = sm(xs, ei)
            assert logits.shape == (n, classes)
            close(logits, out_ref, f"{arch} logits")
            logits.backward(loss_grad(logits).to(logits.dtype))                   # every rank: its own loss on the full logits
            close(xs.grad, gx_ref[:, sl], f"{arch} gx", 5e-5)
            # parameter gradients: every sharded tensor against its slice of the unsharded model's
            checked = 0
            for ci, sc in enumerate(sm.convs):
                for li, layer in enumerate(sc.layers):
                    pre = f"convs.{ci}.nn.layers.{li}."
                    cols = layer.columns
                    if arch == "kan":
                        for name in ("base_weight", "spline_weight", "spline_scaler"):
                            close(getattr(layer, name).grad, g_ref[pre + name][:, cols], pre + name, 5e-5)
                            checked += 1
                    else:
                        ng = layer.centers.numel()
                        gsw = g_ref[pre + "spline_linear.weight"]
                        fout = gsw.size(0)
                        close(layer.spline_weight.grad, gsw.view(fout, -1, ng)[:, cols].reshape(fout, -1), pre + "spline", 5e-5)
                        close(layer.base_weight.grad, g_ref[pre + "base_linear.weight"][:, cols], pre + "base_weight", 5e-5)
                        close(layer.ln_weight.grad, g_ref[pre + "layernorm.weight"][cols], pre + "ln_w", 5e-5)
                        close(layer.ln_bias.grad, g_ref[pre + "layernorm.bias"][cols], pre + "ln_b", 5e-5)
                        checked += 4
            hw = hid // world
            for bi, bn in enumerate(sm.bns):
                close(bn.weight.grad, g_ref[f"bns.{bi}.weight"][rank * hw:(rank + 1) * hw], f"bns.{bi}.weight", 5e-5)
                close(bn.bias.grad, g_ref[f"bns.{bi}.bias"][rank * hw:(rank + 1) * hw], f"bns.{bi}.bias", 5e-5)
            cols = sm.lay_out.columns
            assert cols.numel() == (f + 2 * hid) // world
            if arch == "kan":
                for name in ("base_weight", "spline_weight", "spline_scaler"):
                    close(getattr(sm.lay_out, name).grad, g_ref["lay_out." + name][:, cols], "lay_out." + name, 5e-5)
            else:
                ng = sm.lay_out.centers.numel()
                gsw = g_ref["lay_out.spline_linear.weight"]
                close(sm.lay_out.spline_weight.grad, gsw.view(classes, -1, ng)[:, cols].reshape(classes, -1), "lay_out.spline", 5e-5)
                close(sm.lay_out.ln_weight.grad, g_ref["lay_out.layernorm.weight"][cols], "lay_out.ln_w", 5e-5)
                if rank == 0:
                    close(sm.lay_out.base_bias.grad, g_ref["lay_out.base_linear.bias"], "lay_out.base_bias", 5e-5)
            assert checked >= 6
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_layer_world2_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.parametrize("world,f,hid,grid", [(4, 64, 64, 5), (8, 64, 64, 5), (4, 128, 128, 8), (8, 128, 128, 8), (8, 64, 128, 8)],
                         ids=["P4-F64", "P8-F64", "P4-F128-G8", "P8-F128-G8", "P8-F64-H128-G8"])
def test_sharded_layer_world4_and_world8_gloo(tmp_path, world, f, hid, grid):
    """the SYSTEM at P = 4 and P = 8 (VERDICT r04 missing 2: only P = 2 had ever run): the rank-major [P][n][out/P] staging, the
    reduce-scatter / all-gather algebra, uneven row chunks (400 rows in 3 and 7 chunks) and the transposed variant's all-to-all with
    401 rows over P ranks -- at the widths the P = 8 shards have (8 columns per rank at F = 64, 16 at F = 128 / grid 8)."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), f, hid, grid, False), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("world,f,hid", [(2, 16, 24), (4, 16, 24), (8, 64, 64)], ids=["P2", "P4", "P8-F64"])
def test_sharded_fastkan_layer_and_node_models_gloo(tmp_path, world, f, hid):
    """SURVEY 8(e) beyond the KAN-GIN conv (VERDICT r05 row e2 a + b): feature-sharded FastKAN-GIN layer and the sharded
    GKAN_Nodes / GFASTKAN_Nodes step at world 2 / 4 / 8 against the unsharded oracle (reference ``fastkan.py:76-85``,
    ``models.py:195-203,246-257``)."""
    mp.spawn(_worker_e2, args=(world, _free_port(), str(tmp_path), f, hid), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _small_graph_batches(seed, nb, B=6, n_atom=21, n_bond=4):
    """ZINC-shaped mini-batches (categorical atom / bond features, a few small graphs each), as types.SimpleNamespace"""
    from types import SimpleNamespace
    out = []
    for k in range(nb):
        g = torch.Generator().manual_seed(seed + k)
        sizes = torch.randint(5, 12, (B,), generator=g)
        n = int(sizes.sum()); off = torch.cumsum(sizes, 0) - sizes
        src, dst, batch = [], [], []
        for b in range(B):
            m = int(sizes[b]); eb = 2 * m + 2
            src.append(torch.randint(0, m, (eb,), generator=g) + off[b]); dst.append(torch.randint(0, m, (eb,), generator=g) + off[b])
            batch.append(torch.full((m,), b))
        e = sum(len(t) for t in src)
        x = torch.randint(0, n_atom, (n, 1), generator=g)
        out.append(SimpleNamespace(x=x, edge_index=torch.stack([torch.cat(src), torch.cat(dst)]), edge_attr=torch.randint(0, n_bond, (e,), generator=g),
                                   batch=torch.cat(batch), num_graphs=B, y=x.float().mean() + torch.randn(B, generator=g) * 0.1))
    return out


class _OracleKAGIN(torch.nn.Module):
    """test stand-in for kagnn_amd.KAGINRegression on CPU: the same parameters (a reference-keyed state dict), forward through
    ``oracle.graph_regression_forward`` + stock autograd.  What is under test is harness.train_graph_batches(group=)."""

    def __init__(self, state, gnn_layers):
        super().__init__()
        self.names = [k for k, v in state.items() if v.is_floating_point() and not k.endswith(("grid", "running_mean", "running_var", "eps"))]
        self.params = torch.nn.ParameterList([torch.nn.Parameter(state[k].clone()) for k in self.names])
        self.fixed = {k: v.clone() for k, v in state.items() if k not in self.names}
        self.gnn_layers = gnn_layers

    def forward(self, d):
        from oracle import kan_oracle as orc
        st = dict(self.fixed)
        st.update({k: p for k, p in zip(self.names, self.params)})
        ea = d.edge_attr if d.edge_attr.dim() == 2 else d.edge_attr.view(-1, 1)
        return orc.graph_regression_forward(d.x, d.edge_index, ea, d.batch, d.num_graphs, st, "kan", self.gnn_layers)


def _worker_replicas(rank, world, port, out_dir):
    """SURVEY 8(e) "replicas only" (BASELINE config 4): every rank its own mini-batches, ONE flat gradient all-reduce per step
    (VERDICT r05 row e2 c; reference loop graph_regression/optuna_zinc.py:56-66) -- against the same loop written out on one
    process with the gradients of all ranks' batches averaged by hand."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kagnn_amd
        from kagnn_amd.harness import train_graph_batches
        H, L, steps = 8, 2, 3
        torch.manual_seed(4)
        proto = kagnn_amd.KAGINRegression(1, 1, L, H, 2, 4, 3, 1, 0.0, True)
        proto.atom_encoder = kagnn_amd.graph_models.AtomEncoder(H, [21])
        proto.bond_encoder.bond_embedding_list = torch.nn.ModuleList([torch.nn.Embedding(4, H)])
        state = {k: v.detach().clone() for k, v in proto.state_dict().items()}
        all_batches = [_small_graph_batches(100 * r, steps) for r in range(world)]
        # this rank starts from DIFFERENT parameters: the constructor's broadcast from rank 0 has to repair that
        mine = _OracleKAGIN(state, L)
        with torch.no_grad():
            for p_ in mine.parameters():
                p_.add_(0.01 * rank)
        opt = torch.optim.Adam(mine.parameters(), lr=5e-3)
        _, means = train_graph_batches(mine, all_batches[rank], nb_epochs=2, lr=5e-3, optimizer=opt, group=True, loss_fn=torch.nn.L1Loss())
        # the same loop by hand on one process: per step the mean of the ranks' gradients
        ref = _OracleKAGIN(state, L)
        ropt = torch.optim.Adam(ref.parameters(), lr=5e-3)
        ref.train()
        for _ in range(2):
            for k in range(steps):
                grads = None
                for r in range(world):
                    ropt.zero_grad(set_to_none=True)
                    d = all_batches[r][k]
                    torch.nn.L1Loss()(ref(d).squeeze(), d.y.squeeze()).backward()
                    g = [p.grad.clone() for p in ref.parameters()]
                    grads = g if grads is None else [a + b for a, b in zip(grads, g)]
                for p, g in zip(ref.parameters(), grads):
                    p.grad = g / world
                ropt.step()
        for (name, a), b in zip(zip(mine.names, mine.params), ref.params):
            err = float((a - b).abs().max()) / max(1e-12, float(b.abs().max()))
            # (six Adam steps: the all-reduce adds g_r / P in another association than the hand average, and Adam divides by sqrt(v) --
            # on elements with tiny gradients a last-bit difference becomes a visible fraction of an lr-sized update; seen up to
            # 2.6e-5 over runs whose lstsq-made initial weights differ in the last bit.  A dropped or doubled all-reduce is O(1).)
            assert err < 5e-4, (name, err)
        # replicas stay bit-identical: same reduced gradients, same optimiser state
        flat = torch.cat([p.detach().reshape(-1) for p in mine.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], t) for t in gathered[1:])
        assert all(m == m for m in means)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_graph_level_replicas_one_flat_gradient_all_reduce_gloo(tmp_path, world):
    mp.spawn(_worker_replicas, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


# ---------------------------------------------------------------------------------------------------------
# the same two sharded layers with the PRODUCT's local ops (HIP kernels): two ranks share cuda:0, collectives
# over gloo -- checks the sharding algebra end to end against the unsharded layer on the same device
def _gpu_worker(rank, world, port, out_dir, backend="gloo", f=16, big=True):
    """backend "gloo": both ranks on cuda:0 (what a 1-GPU box can run); "nccl": one device per rank over RCCL -- the real thing,
    including the library's own RCCL entry points (comm="rccl_c") and the barrier's RCCL branch; needs >= 2 GPUs"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kagnn_amd
        from kagnn_amd import ops
        from kagnn_amd.sharded import ShardedGIKANLayer, TransposedShardedGIKANLayer
        from oracle import kan_oracle as orc
        n, e = 3001, 30000
        ei = orc.powerlaw_graph(n, e, seed=3).to(dev)
        gen = torch.Generator().manual_seed(3)
        x = (torch.randn(n, f, generator=gen) * 0.3).to(dev)
        gy = torch.randn(n, f, generator=gen).to(dev)
        torch.manual_seed(9)
        conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2).to(dev)
        graph = ops.GraphIndex(ei, n)
        xr = x.clone().requires_grad_(True)
        y_ref = conv(xr, graph)
        y_ref.backward(gy)
        w = f // world
        sl = slice(rank * w, (rank + 1) * w)

        def close(a, b, what):
            err = float((a - b).abs().max()) / max(1e-300, float(b.abs().max()))
            assert err <= 1e-4, (what, err)

        # comm="p2p": hipIpc peer-mapped exchange buffers + kagnn_p2p_reduce_scatter / _all_gather (two processes, one GPU)
        combos = [(ShardedGIKANLayer, {}), (ShardedGIKANLayer, {"chunks": 3}), (ShardedGIKANLayer, {"comm": "p2p"}),
                  (ShardedGIKANLayer, {"comm": "p2p", "chunks": 3}), (TransposedShardedGIKANLayer, {}), (TransposedShardedGIKANLayer, {"comm": "p2p"})]
        if backend == "nccl":              # (RCCL refuses two ranks on one device: only with a GPU per rank)
            combos += [(ShardedGIKANLayer, {"comm": "rccl_c"}), (ShardedGIKANLayer, {"comm": "rccl_c", "chunks": 3})]
        for cls, kw in combos:
            sconv = cls(conv, None, **kw).to(dev)
            xs = sconv.shard_columns(x).requires_grad_(True)
            y = sconv(xs, graph)
            y.backward(sconv.shard_columns(gy))
            if kw.get("comm") == "p2p":               # a second step through the same exchange buffers (reuse across steps)
                xs.grad = None
                sconv.zero_grad()
                y = sconv(xs, graph)
                y.backward(sconv.shard_columns(gy))
            close(y, y_ref[:, sl], cls.__name__ + ".y")
            close(xs.grad, xr.grad[:, sl], cls.__name__ + ".gx")
            for li, (layer, full) in enumerate(zip(sconv.layers, conv.nn.layers)):
                isl = slice(layer.lo, layer.hi) if cls is ShardedGIKANLayer else slice(None)
                for name in ("base_weight", "spline_weight", "spline_scaler"):
                    close(getattr(layer, name).grad, getattr(full, name).grad[:, isl], f"{cls.__name__}.{li}.{name}")
        # ---- (ADVICE r03 / VERDICT r03 weak 8a) ONE exchange per step and no backward in between: a single-KANLinear chain
        # under no_grad, called repeatedly -- the p2p exchange alternates between two peer-mapped buffers, so a fast rank
        # never overwrites what a slow peer is still pulling; every call must give the unsharded layer's rows
        torch.manual_seed(10)
        conv1 = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=1).to(dev)
        s1 = ShardedGIKANLayer(conv1, None, comm="p2p", chunks=2).to(dev)
        with torch.no_grad():
            for it in range(6):
                x_it = x * (1.0 + 0.2 * it)
                close(s1(s1.shard_columns(x_it), graph), conv1(x_it, graph)[:, sl], f"p2p forward-only call {it}")
        assert s1._xch[0].fwd_uses == 6 and s1._xch[0].bwd_uses == 0
        # ---- the row-CHUNKED exchanges on the HIP kernels (>= 262 144 rows => 4 chunks by default): the headline width
        # (64, grid 5) and config 3's (128, grid 8: two-window kernels), RCCL-free collectives (gloo) and p2p pulls
        if not big:
            open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
            return
        nb, eb = 262144 + 77, 1_500_000
        eib = orc.powerlaw_graph(nb, eb, seed=4).to(dev)
        gb = ops.GraphIndex(eib, nb)
        for fb, grid in ((64, 5), (128, 8)):
            genb = torch.Generator().manual_seed(fb)
            xb = (torch.randn(nb, fb, generator=genb) * 0.3).to(dev)
            gyb = torch.randn(nb, fb, generator=genb).to(dev)
            torch.manual_seed(fb)
            convb = kagnn_amd.GIKANLayer(fb, fb, grid_size=grid, spline_order=3, hidden_dim=fb, nb_layers=2).to(dev)
            xrb = xb.clone().requires_grad_(True)
            yb = convb(xrb, gb)
            yb.backward(gyb)
            wb = fb // world
            slb = slice(rank * wb, (rank + 1) * wb)
            for kw in ({}, {"comm": "p2p"}) + (({"comm": "rccl_c"},) if backend == "nccl" else ()):
                sb = ShardedGIKANLayer(convb, None, **kw).to(dev)
                for step in range(2):                  # twice: both buffers of the p2p exchange
                    xs = sb.shard_columns(xb).requires_grad_(True)
                    sb.zero_grad()
                    y = sb(xs, gb)
                    y.backward(sb.shard_columns(gyb))
                tag = f"chunked F={fb} grid={grid} {kw.get('comm', 'collectives')}"
                close(y, yb[:, slb], tag + ".y")
                close(xs.grad, xrb.grad[:, slb], tag + ".gx")
                for li, (layer, full) in enumerate(zip(sb.layers, convb.nn.layers)):
                    for name in ("base_weight", "spline_weight", "spline_scaler"):
                        close(getattr(layer, name).grad, getattr(full, name).grad[:, layer.lo:layer.hi], f"{tag}.{li}.{name}")
                del sb
            del convb, xb, gyb, xrb, yb
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _gpu_worker_e2(rank, world, port, out_dir, f=32, hid=32, classes=10):
    """the exchanges of _worker_e2 / _worker_replicas with the PRODUCT's local ops: `world` processes sharing cuda:0 (collectives
    over gloo), HIP kernels as local compute, against the unsharded modules on the same device"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kagnn_amd
        from kagnn_amd import ops
        from kagnn_amd.sharded import ShardedGIFASTKANLayer, ShardedNodeModel
        from oracle import kan_oracle as orc
        n, e = 3001, 30000
        ei = orc.powerlaw_graph(n, e, seed=5).to(dev)
        gen = torch.Generator().manual_seed(5)
        x = (torch.randn(n, f, generator=gen) * 0.3 + 0.5).to(dev)
        gy = torch.randn(n, f, generator=gen).to(dev)
        graph = ops.GraphIndex(ei, n)
        w = f // world
        sl = slice(rank * w, (rank + 1) * w)

        def close(a, b, what, tol=1e-4):
            err = float((a - b).abs().max()) / max(1e-300, float(b.abs().max()))
            assert err <= tol, (what, err)

        # ---- (a) the feature-sharded FastKAN-GIN layer, both precision modes, 1 and 3 row chunks
        for mode in (ops.PREC_SPLIT, ops.PREC_FP32):
            torch.manual_seed(11)
            conv = kagnn_amd.GIFASTKANLayer(f, f, grid_size=5, hidden_dim=hid, nb_layers=2)
            with torch.no_grad():
                for l in conv.nn.layers:
                    l.layernorm.weight.uniform_(0.5, 1.5); l.layernorm.bias.uniform_(-0.3, 0.3); l.base_linear.bias.uniform_(-0.5, 0.5)
                    l.precision = mode
            conv = conv.to(dev)
            xr = x.clone().requires_grad_(True)
            y_ref = conv(xr, graph)
            y_ref.backward(gy)
            for chunks in (1, 3):
                sconv = ShardedGIFASTKANLayer(conv, None, chunks=chunks).to(dev)
                for step in range(2):                      # twice: the side stream / workspace reuse across steps
                    xs = sconv.shard_columns(x).requires_grad_(True)
                    sconv.zero_grad()
                    y = sconv(xs, graph)
                    y.backward(sconv.shard_columns(gy))
                tag = f"fastkan mode {mode} chunks {chunks}"
                close(y, y_ref[:, sl], tag + " y")
                close(xs.grad, xr.grad[:, sl], tag + " gx")
                for li, (layer, full) in enumerate(zip(sconv.layers, conv.nn.layers)):
                    cols, ng = layer.columns, layer.centers.numel()
                    close(layer.ln_weight.grad, full.layernorm.weight.grad[cols], f"{tag} L{li} ln_w")
                    close(layer.ln_bias.grad, full.layernorm.bias.grad[cols], f"{tag} L{li} ln_b")
                    fo = full.output_dim
                    close(layer.spline_weight.grad, full.spline_linear.weight.grad.view(fo, -1, ng)[:, cols].reshape(fo, -1), f"{tag} L{li} spline")
                    close(layer.base_weight.grad, full.base_linear.weight.grad[:, cols], f"{tag} L{li} base_w")
                    if rank == 0:
                        close(layer.base_bias.grad, full.base_linear.bias.grad, f"{tag} L{li} base_b")

        # ---- (b) whole node models on column shards against the unsharded models (fused epilogue paths and all)
        labels = torch.randint(0, classes, (n,), generator=gen).to(dev)
        for arch in ("kan", "fastkan"):
            torch.manual_seed(12)
            if arch == "kan":
                model = kagnn_amd.GKAN_Nodes("gin", 2, f, hid, classes, grid_size=4, spline_order=3, hidden_layers=2)
            else:
                model = kagnn_amd.GFASTKAN_Nodes("gin", 2, f, hid, classes, grid_size=4, hidden_layers=2)
            with torch.no_grad():
                for bn in model.bns:
                    bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
            model = model.to(dev).train()
            sm = ShardedNodeModel(model, None, chunks=2).to(dev).train()       # (built BEFORE the reference step: same running statistics)
            xr = x.clone().requires_grad_(True)
            out = model(xr, graph)
            ops.softmax_cross_entropy(out, labels, None, pre_softmax=True).backward()
            xs = sm.shard_columns(x).requires_grad_(True)
            logits = sm(xs, graph)
            close(logits, out, f"{arch} logits")
            ops.softmax_cross_entropy(logits, labels, None, pre_softmax=True).backward()   # every rank: the loss on the full logits
            close(xs.grad, xr.grad[:, sl], f"{arch} gx", 2e-4)
            for ci, (sc, fc) in enumerate(zip(sm.convs, model.convs)):
                for li, (layer, full) in enumerate(zip(sc.layers, fc.nn.layers)):
                    cols = layer.columns
                    if arch == "kan":
                        for name in ("base_weight", "spline_weight", "spline_scaler"):
                            close(getattr(layer, name).grad, getattr(full, name).grad[:, cols], f"{arch} convs.{ci}.{li}.{name}", 2e-4)
                    else:
                        ng, fo = layer.centers.numel(), full.output_dim
                        close(layer.spline_weight.grad, full.spline_linear.weight.grad.view(fo, -1, ng)[:, cols].reshape(fo, -1), f"{arch} convs.{ci}.{li}.spline", 2e-4)
                        close(layer.ln_weight.grad, full.layernorm.weight.grad[cols], f"{arch} convs.{ci}.{li}.ln_w", 2e-4)
            hw = hid // world
            for bi, (sb, fb) in enumerate(zip(sm.bns, model.bns)):
                close(sb.weight.grad, fb.weight.grad[rank * hw:(rank + 1) * hw], f"{arch} bns.{bi}.weight", 2e-4)
                close(sb.bias.grad, fb.bias.grad[rank * hw:(rank + 1) * hw], f"{arch} bns.{bi}.bias", 2e-4)
                close(sb.running_mean, fb.running_mean[rank * hw:(rank + 1) * hw], f"{arch} bns.{bi}.running_mean")
                close(sb.running_var, fb.running_var[rank * hw:(rank + 1) * hw], f"{arch} bns.{bi}.running_var")
            cols = sm.lay_out.columns
            if arch == "kan":
                for name in ("base_weight", "spline_weight", "spline_scaler"):
                    close(getattr(sm.lay_out, name).grad, getattr(model.lay_out, name).grad[:, cols], f"{arch} lay_out.{name}", 2e-4)
            else:
                ng = sm.lay_out.centers.numel()
                close(sm.lay_out.spline_weight.grad, model.lay_out.spline_linear.weight.grad.view(classes, -1, ng)[:, cols].reshape(classes, -1),
                      f"{arch} lay_out.spline", 2e-4)
                close(sm.lay_out.ln_weight.grad, model.lay_out.layernorm.weight.grad[cols], f"{arch} lay_out.ln_w", 2e-4)

        # ---- (c) BASELINE config 4 as data-parallel replicas: every rank its own ZINC-shaped mini-batches, ONE flat gradient
        # all-reduce per step, harness.Adam -- against the same steps on one process with the ranks' gradients averaged by hand
        from kagnn_amd.harness import Adam, train_graph_batches
        H, L, steps = 32, 2, 3

        def make():
            torch.manual_seed(13)
            m = kagnn_amd.KAGINRegression(1, 1, L, H, 2, 4, 3, 1, 0.0, True)
            m.atom_encoder = kagnn_amd.graph_models.AtomEncoder(H, [21])
            m.bond_encoder.bond_embedding_list = torch.nn.ModuleList([torch.nn.Embedding(4, H)])
            return m.to(dev)
        all_batches = []
        for r in range(world):
            bs = _small_graph_batches(200 * r + 7, steps, B=24)
            for b in bs:
                for k in ("x", "edge_index", "edge_attr", "batch", "y"):
                    setattr(b, k, getattr(b, k).to(dev))
            all_batches.append(bs)
        mine = make()
        with torch.no_grad():
            for p_ in mine.parameters():
                p_.add_(0.01 * rank)                       # (the constructor's broadcast from rank 0 has to repair this)
        _, means = train_graph_batches(mine, all_batches[rank], nb_epochs=2, lr=5e-3, group=True)
        ref = make()
        # (rank 0's construction is THE model: the spline-weight init is a CPU lstsq whose last bit is not reproducible)
        for p_ in ref.parameters():
            dist.broadcast(p_.data, src=0)
        ropt = Adam(ref.parameters(), lr=5e-3)
        ref.train()
        for _ in range(2):
            for k in range(steps):
                grads = None
                for r in range(world):
                    ropt.zero_grad()
                    d = all_batches[r][k]
                    ops.l1_loss(ref(d).squeeze(), d.y.squeeze()).backward()
                    g = [p_.grad.clone() for p_ in ref.parameters()]
                    grads = g if grads is None else [a + b for a, b in zip(grads, g)]
                for p_, g in zip(ref.parameters(), grads):
                    p_.grad = g / world
                ropt.step()
        for (name, a), b in zip(mine.named_parameters(), ref.parameters()):
            # (the all-reduce sums g_r / P in another association than the hand average, and Adam divides by sqrt(v): on elements whose
            # gradient is tiny a rounding-level difference becomes a visible fraction of an lr-sized update; first-step gradients are
            # compared to the bit by the CPU test)
            close(a, b, f"replicas {name}", 3e-3)
        flat = torch.cat([p_.detach().reshape(-1) for p_ in mine.parameters()]).cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], t) for t in gathered[1:]), "replicas diverged"
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _gpu_worker_e2_big(rank, world, port, out_dir):
    """the round-6 sharded paths at REAL sizes, two processes on cuda:0: the FastKAN-GIN layer at 262 221 rows (the default 4 row
    chunks: reduce-scatter / all-gather chunked and overlapped, LayerNorm exchange over all rows at once) and BASELINE config 2's
    model shape -- GKAN_Nodes(gin, 3 layers, 128 features, hidden 64, 40 classes) on an arxiv-sized graph -- as a ShardedNodeModel
    against the unsharded model's fused default path"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kagnn_amd
        from kagnn_amd import ops
        from kagnn_amd.sharded import ShardedGIFASTKANLayer, ShardedNodeModel
        from oracle import kan_oracle as orc

        def close(a, b, what, tol=1e-4):
            err = float((a - b).abs().max()) / max(1e-300, float(b.abs().max()))
            assert err <= tol, (what, err)

        # ---- FastKAN-GIN layer, 4 row chunks by default
        n, e, f = 262144 + 77, 1_500_000, 64
        ei = orc.powerlaw_graph(n, e, seed=6).to(dev)
        graph = ops.GraphIndex(ei, n)
        gen = torch.Generator().manual_seed(6)
        x = (torch.randn(n, f, generator=gen) * 0.3 + 0.2).to(dev)
        gy = torch.randn(n, f, generator=gen).to(dev)
        torch.manual_seed(21)
        conv = kagnn_amd.GIFASTKANLayer(f, f, grid_size=8, hidden_dim=f, nb_layers=2).to(dev)
        xr = x.clone().requires_grad_(True)
        y_ref = conv(xr, graph)
        y_ref.backward(gy)
        w = f // world
        sl = slice(rank * w, (rank + 1) * w)
        sconv = ShardedGIFASTKANLayer(conv, None).to(dev)
        xs = sconv.shard_columns(x).requires_grad_(True)
        y = sconv(xs, graph)
        y.backward(sconv.shard_columns(gy))
        close(y, y_ref[:, sl], "big fastkan y")
        close(xs.grad, xr.grad[:, sl], "big fastkan gx")
        for li, (layer, full) in enumerate(zip(sconv.layers, conv.nn.layers)):
            cols, ng, fo = layer.columns, layer.centers.numel(), full.output_dim
            close(layer.spline_weight.grad, full.spline_linear.weight.grad.view(fo, -1, ng)[:, cols].reshape(fo, -1), f"big fastkan L{li} spline")
            close(layer.ln_weight.grad, full.layernorm.weight.grad[cols], f"big fastkan L{li} ln_w")
            close(layer.ln_bias.grad, full.layernorm.bias.grad[cols], f"big fastkan L{li} ln_b")
        del conv, sconv, xr, xs, y, y_ref, x, gy, graph, ei
        torch.cuda.empty_cache()

        # ---- config 2's model shape on column shards
        n, e, fin, hid, classes = 169_343, 1_166_243, 128, 64, 40
        ei = orc.powerlaw_graph(n, e, seed=7).to(dev)
        graph = ops.GraphIndex(ei, n)
        gen = torch.Generator().manual_seed(7)
        x = (torch.randn(n, fin, generator=gen) * 0.4).to(dev)
        labels = torch.randint(0, classes, (n,), generator=gen).to(dev)
        torch.manual_seed(22)
        model = kagnn_amd.GKAN_Nodes("gin", 3, fin, hid, classes, grid_size=5, spline_order=3, hidden_layers=2).to(dev).train()
        sm = ShardedNodeModel(model, None).to(dev).train()
        xr = x.clone().requires_grad_(True)
        out = model(xr, graph)
        ops.softmax_cross_entropy(out, labels, None, pre_softmax=True).backward()
        xs = sm.shard_columns(x).requires_grad_(True)
        logits = sm(xs, graph)
        close(logits, out, "arxiv-shaped logits", 2e-4)
        ops.softmax_cross_entropy(logits, labels, None, pre_softmax=True).backward()
        wi = fin // world
        close(xs.grad, xr.grad[:, rank * wi:(rank + 1) * wi], "arxiv-shaped gx", 1e-3)      # (three norms deep, a mean-type loss: tiny gradients)
        for ci, (sc, fc) in enumerate(zip(sm.convs, model.convs)):
            for li, (layer, full) in enumerate(zip(sc.layers, fc.nn.layers)):
                for name in ("base_weight", "spline_weight", "spline_scaler"):
                    close(getattr(layer, name).grad, getattr(full, name).grad[:, layer.columns], f"arxiv-shaped convs.{ci}.{li}.{name}", 1e-3)
        for name in ("base_weight", "spline_weight", "spline_scaler"):
            close(getattr(sm.lay_out, name).grad, getattr(model.lay_out, name).grad[:, sm.lay_out.columns], f"arxiv-shaped lay_out.{name}", 1e-3)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_fastkan_layer_and_arxiv_shaped_node_model_two_ranks_one_gpu_real_sizes(tmp_path):
    mp.spawn(_gpu_worker_e2_big, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_fastkan_node_models_and_replicas_ranks_on_one_gpu(tmp_path, world):
    """VERDICT r05 row e2 on the HIP kernels: feature-sharded FastKAN-GIN layer (kagnn_fastkan_row_moments / _merge_moments /
    _shard_fwd / _shard_bwd / _shard_bwd_finish), sharded GKAN_Nodes / GFASTKAN_Nodes steps, and config 4's replicas with one flat
    gradient all-reduce -- 2 and 4 processes sharing cuda:0"""
    mp.spawn(_gpu_worker_e2, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.gpu
def test_sharded_layers_two_ranks_one_gpu(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.gpu
def test_sharded_layers_four_ranks_one_gpu(tmp_path):
    """P = 4 as a SYSTEM on the HIP kernels (VERDICT r04 missing 2): four PROCESSES sharing cuda:0, 32 features = 8 columns per
    rank (the narrow instantiations of the P = 8 headline shard), collectives over gloo and -- comm="p2p" -- kagnn_p2p_reduce_scatter /
    _all_gather pulling from three other processes' hipIpc-mapped buffers, rank-ordered sums, both p2p schemes, chunked and not,
    the two-buffer reuse across steps and forward-only calls."""
    mp.spawn(_gpu_worker, args=(4, _free_port(), str(tmp_path), "gloo", 32, False), nprocs=4, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(4))


@pytest.mark.gpu
def test_sharded_layers_four_ranks_one_gpu_chunked_headline_widths(tmp_path):
    """the same four processes on the row-CHUNKED exchanges at the headline's and config 3's widths (64 / grid 5 -> 16 columns per
    rank; 128 / grid 8 -> 32 columns per rank, two-window kernels): 262 221 rows => 4 chunks, gloo collectives and p2p pulls"""
    mp.spawn(_gpu_worker, args=(4, _free_port(), str(tmp_path), "gloo", 16, True), nprocs=4, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(4))


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one rank per device over RCCL / xGMI")
def test_sharded_layers_two_ranks_two_gpus_rccl(tmp_path):
    """the same checks with one DEVICE per rank: collectives over RCCL, peer pulls over xGMI, the barrier's RCCL branch, and the
    library's own RCCL entry points (comm="rccl_c") -- skipped on the 1-GPU boxes every round of this build has had"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


# ---------------------------------------------------------------------------------------------------------
# bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), both ranks on cuda:0 with
# the collectives on gloo (KAGNN_BENCH_BACKEND=gloo: numbers meaningless, control flow real): the combination loop, the
# reporter process that owns THE line, and the watchdog -- none of the N > 1 transports has ever run on two devices, so
# the line must survive one that hangs and one that takes a rank down (KAGNN_BENCH_FAULT)
def _run_bench_two_ranks(extra_env, timeout=420, nproc=2, workload=None):
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, KAGNN_BENCH_BACKEND="gloo", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
           "--nodes", "20000", "--edges", "200000"] + (["--workload", workload] if workload else [])
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    return r.returncode, json.loads(lines[0]), r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_prints_one_line_even_when_a_transport_hangs_or_kills_a_rank():
    rc, line, err = _run_bench_two_ranks({})
    assert rc == 0, err[-3000:]
    combos = {(c["scheme"], c["comm"]): c for c in line["multi_gpu_probe"]["combinations"]}
    assert set(combos) == {("feature", "rccl"), ("transposed", "rccl"), ("feature", "p2p"), ("transposed", "p2p"), ("feature", "rccl_c")}
    for k in (("feature", "rccl"), ("transposed", "rccl"), ("feature", "p2p"), ("transposed", "p2p")):
        assert "error" not in combos[k] and combos[k]["ms_per_step"] > 0, combos[k]
    # two ranks on one device: RCCL itself refuses (the library entry points need one GPU per rank) -- listed, not fatal
    assert "error" in combos[("feature", "rccl_c")], combos[("feature", "rccl_c")]
    assert "interim" not in line and line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["north_star_scheme"]["scheme"] == "feature" and "per_rank" in line and line["roofline"]["frac"] > 0
    sel = line["multi_gpu_probe"]["selected"]
    assert (sel["scheme"], sel["comm"]) in combos and "error" not in combos[(sel["scheme"], sel["comm"])]
    # `value` is north_star's scheme (it used to be the fastest combination, i.e. silently another
    # partitioning); the fastest of the rest sits beside it under its own key
    assert line["complete"] is True and line["value_scheme_is_north_star"] is True
    assert (sel["scheme"], sel["comm"]) == ("feature", "rccl")            # (rccl_c cannot run with two ranks on one device)
    fa = line["fastest_alternative"]
    assert (fa["scheme"], fa["comm"]) in {("transposed", "rccl"), ("feature", "p2p"), ("transposed", "p2p")} and fa["ms_per_step"] > 0
    assert fa["ms_per_step"] == min(combos[k]["ms_per_step"] for k in combos if k[1] != "rccl_c" and k != ("feature", "rccl"))

    # a transport that hangs on one rank: the watchdog ends the run, the line is the interim one of what had finished
    rc, line, err = _run_bench_two_ranks({"KAGNN_BENCH_FAULT": "feature/p2p:hang", "KAGNN_BENCH_PHASE_TIMEOUT": "25"})
    assert "interim" in line and line["value"] > 0 and "overran its deadline" in err
    assert rc != 0 and line["complete"] is False         # (a hung transport must be visible to the launcher)
    assert line["value_scheme_is_north_star"] is True and line["fastest_alternative"]["scheme"] == "transposed"
    done = [(c["scheme"], c["comm"]) for c in line["multi_gpu_probe"]["combinations"]]
    assert done == [("feature", "rccl"), ("transposed", "rccl")]
    assert line["multi_gpu_probe"]["not_finished"] == ["feature/p2p", "transposed/p2p", "feature/rccl_c"]

    # a transport that takes rank 0 itself down (SIGKILL): the reporter still prints what rank 0 had handed it
    rc, line, err = _run_bench_two_ranks({"KAGNN_BENCH_FAULT": "transposed/p2p:kill:0", "KAGNN_BENCH_PHASE_TIMEOUT": "25"})
    assert "interim" in line and len(line["multi_gpu_probe"]["combinations"]) == 3 and line["ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_four_ranks_one_gpu_runs_every_combination():
    """bench.py --gpus 4 as the driver launches it (four ranks on cuda:0, collectives on gloo): the probe, the selection of north_star's
    scheme and the per-rank gather at P = 4 -- 16 columns per rank at the headline width"""
    rc, line, err = _run_bench_two_ranks({}, nproc=4)
    assert rc == 0, err[-3000:]
    combos = {(c["scheme"], c["comm"]): c for c in line["multi_gpu_probe"]["combinations"]}
    for k in (("feature", "rccl"), ("transposed", "rccl"), ("feature", "p2p"), ("transposed", "p2p")):
        assert "error" not in combos[k] and combos[k]["ms_per_step"] > 0, combos[k]
    assert line["n_gpus"] == 4 and line["complete"] is True and line["value_scheme_is_north_star"] is True and len(line["per_rank"]) == 4


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["fastkan", "model"])
def test_bench_two_ranks_fastkan_and_model_workloads(workload):
    """`bench.py --gpus N --workload fastkan|model` produce a complete line under the watchdog (two ranks on
    cuda:0, collectives on gloo): the feature-sharded FastKAN-GIN layer and the whole GKAN_Nodes training step on column shards"""
    rc, line, err = _run_bench_two_ranks({}, workload=workload)
    assert rc == 0, err[-3000:]
    assert line["complete"] is True and line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["config"]["workload"].startswith(workload) and line["value_scheme_is_north_star"] is True
    combos = line["multi_gpu_probe"]["combinations"]
    assert len(combos) == 1 and "error" not in combos[0], combos
    assert line["roofline"]["frac"] > 0 and len(line["per_rank"]) == 2
    if workload == "model":
        assert abs(line["value"] - 3 * 200000 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
