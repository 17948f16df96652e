"""Full-size (BASELINE.json metric: N=1M, E=10M, F=64) checks of the HIP path through
size-independent properties plus oracle comparisons on row samples, and the feature-sharded layer on
one GPU."""
import pytest
import torch

import kagnn_amd
from kagnn_amd import ops
from kagnn_amd.sharded import ShardedKANLinear
from oracle import kan_oracle as orc
from helpers import assert_close, oracle_kan_linear_fwd_bwd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, E, F = 1_000_000, 10_000_000, 64


@pytest.fixture(scope="module")
def big():
    ei = orc.powerlaw_graph(N, E, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, F, generator=g) * 0.25
    return ei, x, ops.GraphIndex(ei.to(DEV), N)


def test_fullsize_csr_properties(big):
    ei, _, gi = big
    rp = gi.rowptr.long()
    assert int(rp[0]) == 0 and int(rp[-1]) == E and bool((rp[1:] >= rp[:-1]).all())
    assert torch.equal(rp[1:] - rp[:-1], torch.bincount(ei[1].to(DEV), minlength=N))      # histogram
    perm = gi.perm.long()
    assert torch.equal(torch.sort(perm).values, torch.arange(E, device=DEV))                # a permutation
    dst_sorted = ei[1].to(DEV)[perm]
    assert bool((dst_sorted[1:] >= dst_sorted[:-1]).all())                                  # sortedness
    same = dst_sorted[1:] == dst_sorted[:-1]
    assert bool((perm[1:][same] > perm[:-1][same]).all())                                   # stability
    assert torch.equal(gi.col.long(), ei[0].to(DEV)[perm])
    assert gi.num_hub_seg > 0                                                               # top hub ~1e4 in-edges


def test_fullsize_aggregation_checksum_and_samples(big):
    ei, x, gi = big
    xd = x.to(DEV)
    out = ops.aggregate_sum(xd, gi, self_scale=1.0)
    # checksum of checksums: column sums of the output == column sums of all messages + self term (fp64)
    want = xd.double().sum(0) + xd.double().index_select(0, ei[0].to(DEV)).sum(0)
    assert_close(out.double().sum(0), want, 1e-6, what="column checksum")   # fp32 row rounding, random walk over 1M rows
    # sampled destination rows (incl. the biggest hub and an isolated node) against the CPU oracle
    deg = torch.bincount(ei[1], minlength=N)
    rows = torch.cat([torch.randint(0, N, (3000,), generator=torch.Generator().manual_seed(1)),
                      deg.argmax().view(1), (deg == 0).nonzero()[:3].view(-1)])
    mask = torch.zeros(N, dtype=torch.bool); mask[rows] = True
    keep = mask[ei[1]]
    sub = orc.sum_aggregate(x.double(), ei[:, keep], N) + x.double()
    assert_close(out.cpu()[rows], sub[rows], what="sampled rows")
    # transpose consistency: <A x, y> == <x, A^T y>
    y = torch.randn(N, F, generator=torch.Generator().manual_seed(2)).to(DEV)
    aty = ops._aggregate_raw(y, gi, True, 1.0, None, None, None, None, False)
    lhs = float((out.double() * y.double()).sum()); rhs = float((xd.double() * aty.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))


@pytest.mark.parametrize("mode", [ops.PREC_SPLIT, ops.PREC_FP32], ids=["split", "fp32"])
def test_fullsize_kanlinear_samples_and_linearity(big, mode):
    _, x, _ = big
    gen = torch.Generator().manual_seed(3)
    p = orc.init_kan_linear(F, F, 5, 3, gen)
    layer = kagnn_amd.KANLinear(F, F, grid_size=5, spline_order=3)
    layer.load_state_dict(p)
    layer = layer.to(DEV)
    layer.precision = mode
    xs = (x * 3.0).to(DEV).requires_grad_(True)          # std 0.75: most values inside the spline support
    rows = torch.randint(0, N, (4096,), generator=gen)
    gy = torch.zeros(N, F)
    gy[rows] = torch.randn(rows.numel(), F, generator=gen)
    y = layer(xs)
    y.backward(gy.to(DEV))
    ur = torch.unique(rows)
    y64, gx64, g64 = oracle_kan_linear_fwd_bwd((x * 3.0)[ur], gy[ur], p, 3)
    assert_close(y.detach().cpu()[ur], y64, what="y rows")
    assert_close(xs.grad.cpu()[ur], gx64, what="gx rows")
    for k in ("base_weight", "spline_weight", "spline_scaler"):
        assert_close(getattr(layer, k).grad, g64[k], what="g_" + k)    # gy is zero outside the sample
    off = torch.ones(N, dtype=torch.bool); off[ur] = False
    assert float(xs.grad[off.to(DEV)].abs().max()) == 0.0             # no gradient leaks to other rows
    # linearity of the weight gradient in gy / additivity over a row partition (full size)
    g2 = torch.randn(N, F, generator=gen).to(DEV)
    half = N // 2 + 17
    def wgrad(gmat):
        for q in layer.parameters():
            q.grad = None
        layer(xs.detach()).backward(gmat)
        return layer.spline_weight.grad.clone(), layer.base_weight.grad.clone()
    full_s, full_b = wgrad(g2)
    ga = g2.clone(); ga[half:] = 0
    gb = g2.clone(); gb[:half] = 0
    a_s, a_b = wgrad(ga)
    b_s, b_b = wgrad(gb)
    assert_close(a_s + b_s, full_s, 2e-5, what="dW additivity")
    assert_close(a_b + b_b, full_b, 2e-5, what="dWb additivity")


def _oracle_chain_rows(h0_rows, gy_rows, layers):
    """fp64 oracle of the KAN chain on a set of rows: y, d/dh0 and the parameter gradients those rows contribute"""
    h = h0_rows.double().requires_grad_(True)
    ps = [{k: (v.double().requires_grad_(True) if k != "grid" else v.double()) for k, v in p.items()} for p in layers]
    t = h
    for p in ps:
        t = orc.kan_linear_forward(t, p["base_weight"], p["spline_weight"], p["spline_scaler"], p["grid"], 3)
    t.backward(gy_rows.double())
    return t.detach(), h.grad, [{k: p[k].grad for k in ("base_weight", "spline_weight", "spline_scaler")} for p in ps]


def test_fullsize_one_call_gin_layer_backward_vs_oracle(big):
    """THE TIMED PATH's backward at the metric's size: GIKANLayer(64 -> 64, grid 5, 2 KANLinears) on the
    1M / 10M graph through ``kagnn_gin_kan_layer_fwd / _bwd`` (one library call each way -- what bench.py times).
    (A) upstream gradient non-zero on a 4 096-row sample: y on the sample, EVERY parameter gradient (exact: the other rows
        contribute zero) and the whole input gradient (rows reached from the sample against the fp64 oracle, bit-zero elsewhere);
        the oracle's chain is fed the device's own h0 rows, its transposed aggregation runs over the edges into the sample.
    (B) dense upstream gradient: gx on sampled SOURCE rows (needs d/dh0 at every out-neighbour: the oracle chain runs on
        those rows), and additivity of the parameter gradients over a row partition of gy through the same entry points."""
    ei, x, gi = big
    gen = torch.Generator().manual_seed(41)
    torch.manual_seed(41)
    conv = kagnn_amd.GIKANLayer(F, F, grid_size=5, spline_order=3, hidden_dim=F, nb_layers=2)
    layers = [{k: v.detach().clone() for k, v in l.state_dict().items()} for l in conv.nn.layers]
    conv = conv.to(DEV)
    xd = x.to(DEV)
    h0 = ops.aggregate_sum(xd, gi, self_scale=1.0)            # (checked against the oracle in the aggregation test above)
    src, dst = ei[0], ei[1]
    deg = torch.bincount(dst, minlength=N)

    def run(gy_dev):
        conv.zero_grad()
        xr = xd.detach().requires_grad_(True)
        timer = ops.EntryPointTimer()
        ops.set_timer(timer)
        try:
            y = conv(xr, gi)
            y.backward(gy_dev)
        finally:
            ops.set_timer(None)
        names = {r[0] for r in timer.records}
        # the one-call path ran (the binding always goes through ..._bwd_add: kagnn_gin_kan_layer_bwd is that with no addend)
        assert "kagnn_gin_kan_layer_fwd" in names and ({"kagnn_gin_kan_layer_bwd", "kagnn_gin_kan_layer_bwd_add"} & names), names
        assert not ({"kagnn_kan_linear_fwd", "kagnn_kan_linear_bwd_input", "kagnn_kan_linear_bwd_weight", "kagnn_aggregate_sum"} & names), names
        return y.detach(), xr.grad, [{k: getattr(l, k).grad.clone() for k in ("base_weight", "spline_weight", "spline_scaler")}
                                     for l in conv.nn.layers]

    # ---- (A) sparse upstream gradient
    rows = torch.unique(torch.cat([torch.randint(0, N, (4096,), generator=gen), deg.argmax().view(1),
                                   (deg == 0).nonzero()[:3].view(-1)]))
    gy = torch.zeros(N, F)
    gy[rows] = torch.randn(rows.numel(), F, generator=gen)
    y, gx, g = run(gy.to(DEV))
    y64, gh0, g64 = _oracle_chain_rows(h0[rows.to(DEV)].cpu(), gy[rows], layers)
    assert_close(y.cpu()[rows], y64, what="fullsize layer y rows")
    for li in range(2):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            assert_close(g[li][k], g64[li][k], what=f"fullsize layer L{li}.{k}")
    slot = torch.full((N,), -1, dtype=torch.long)
    slot[rows] = torch.arange(rows.numel())
    into = slot[dst] >= 0                                         # edges whose destination carries a gradient
    want = torch.zeros(N, F, dtype=torch.float64)
    want[rows] = gh0                                              # self term, (1 + eps) = 1
    want.index_add_(0, src[into], gh0[slot[dst[into]]])
    touched = torch.zeros(N, dtype=torch.bool)
    touched[rows] = True
    touched[src[into]] = True
    gxc = gx.cpu()
    assert_close(gxc[touched], want[touched], what="fullsize layer gx (rows reached from the sample)")
    assert float(gxc[~touched].abs().max()) == 0.0               # nothing leaks to unreachable rows
    del want, gxc

    # ---- (B) dense upstream gradient
    gyd = torch.randn(N, F, generator=gen)
    y, gx, g_full = run(gyd.to(DEV))
    js = torch.unique(torch.cat([torch.randint(0, N, (400,), generator=gen), deg.argmax().view(1)]))
    jmask = torch.zeros(N, dtype=torch.bool); jmask[js] = True
    out_e = jmask[src]                                            # edges leaving the sampled sources
    need = torch.unique(torch.cat([js, dst[out_e]]))
    slot = torch.full((N,), -1, dtype=torch.long)
    slot[need] = torch.arange(need.numel())
    _, gh0, _ = _oracle_chain_rows(h0[need.to(DEV)].cpu(), gyd[need], layers)
    want = torch.zeros(N, F, dtype=torch.float64)
    want[js] = gh0[slot[js]]
    want.index_add_(0, src[out_e], gh0[slot[dst[out_e]]])
    assert_close(gx.cpu()[js], want[js], what="fullsize layer gx on sampled source rows (dense gy)")
    half = N // 2 + 29
    ga = gyd.clone(); ga[half:] = 0
    gb = gyd.clone(); gb[:half] = 0
    _, gxa, g_a = run(ga.to(DEV))
    _, gxb, g_b = run(gb.to(DEV))
    for li in range(2):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            assert_close(g_a[li][k] + g_b[li][k], g_full[li][k], 2e-5, what=f"fullsize layer L{li}.{k} additivity")
    assert_close(gxa + gxb, gx, 2e-5, what="fullsize layer gx linearity in gy", elementwise=False)


def test_two_feature_shards_sum_to_the_full_layer():
    """the per-rank pieces of kagnn_amd.sharded on ONE GPU: partial sums over input-feature shards
    (strided column views, narrow in_features) add up to the unsharded layer, fwd and bwd."""
    n = 5000
    gen = torch.Generator().manual_seed(4)
    full = kagnn_amd.KANLinear(64, 64, grid_size=5, spline_order=3).to(DEV)
    x = (torch.randn(n, 64, generator=gen) * 0.6).to(DEV)
    gy = torch.randn(n, 64, generator=gen).to(DEV)
    xr = x.clone().requires_grad_(True)
    yf = full(xr)
    yf.backward(gy)
    for world in (2, 8):
        total = torch.zeros_like(yf)
        gx = torch.zeros_like(x)
        for r in range(world):
            sh = ShardedKANLinear(full, r, world).to(DEV)
            xs = x[:, sh.lo:sh.hi].detach().requires_grad_(True)        # strided view, ld = 64
            part = sh(xs, ops)
            part.backward(gy)
            total += part.detach()
            gx[:, sh.lo:sh.hi] = xs.grad
            assert_close(sh.base_weight.grad, full.base_weight.grad[:, sh.lo:sh.hi], what="g_base shard")
            assert_close(sh.spline_weight.grad, full.spline_weight.grad[:, sh.lo:sh.hi], what="g_spline shard")
            assert_close(sh.spline_scaler.grad, full.spline_scaler.grad[:, sh.lo:sh.hi], what="g_scaler shard")
        assert_close(total, yf.detach(), what=f"sum of {world} partials")
        assert_close(gx, xr.grad, what=f"gx from {world} shards")


def test_sharded_layer_world1_nccl():
    """ShardedGIKANLayer end to end through RCCL with a 1-rank group (the 8-GPU run is the driver's)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        from kagnn_amd.sharded import ShardedGIKANLayer
        n, e, f = 20000, 200000, 64
        ei = orc.powerlaw_graph(n, e, seed=5)
        gen = torch.Generator().manual_seed(5)
        x = torch.randn(n, f, generator=gen) * 0.25
        gy = torch.randn(n, f, generator=gen)
        conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2)
        layers = [{k: v.detach().clone() for k, v in l.state_dict().items()} for l in conv.nn.layers]
        y_ref, gx_ref, _ = orc.kan_gin_layer_fwd_bwd(x, ei, layers, 3, gy)
        graph = ops.GraphIndex(ei.to(DEV), n)
        for chunks in (1, 3):                  # 3: row-chunked collectives launched under the side stream, gathers ahead
            s = ShardedGIKANLayer(conv, None, chunks=chunks).to(DEV)
            xs = x.to(DEV).requires_grad_(True)
            y = s(xs, graph)
            y.backward(gy.to(DEV))
            assert_close(y, y_ref, what=f"sharded y (chunks={chunks})")
            assert_close(xs.grad, gx_ref, what=f"sharded gx (chunks={chunks})")
    finally:
        if created:
            dist.destroy_process_group()


def test_rccl_c_entry_points_world1_vs_oracle_and_plain_ops():
    """libkagnn_rccl.so (include/kagnn_rccl.h): kagnn_sharded_kan_linear_fwd / _bwd on an ncclComm_t of one rank -- the
    staging kernels, ncclReduceScatter / ncclAllGather on the side stream, the row chunks, ONE weight-gradient pass.
    With one rank the exchange is the identity, so every output must equal the plain KANLinear ops BIT FOR BIT (rows are
    independent in y and gx; the weight gradient is the same single launch); the whole sharded layer (comm="rccl_c") is
    then checked against the fp64 oracle.  More than one rank needs more than one GPU (RCCL refuses two ranks on a device)."""
    import os
    import torch.distributed as dist
    from kagnn_amd import rccl
    comm = rccl.Communicator(rccl.Communicator.unique_id(), 1, 0, torch.device(DEV))
    try:
        for n, fin, fout, grid, chunks in ((5000, 64, 64, 5, 1), (5000, 64, 64, 5, 3), (4097, 24, 40, 5, 2),
                                           (300_000, 64, 64, 5, 4), (70_001, 128, 128, 8, 3), (7, 16, 16, 3, 4)):
            torch.manual_seed(n + fin)
            layer = kagnn_amd.KANLinear(fin, fout, grid_size=grid, spline_order=3).to(DEV)
            gen = torch.Generator(device=DEV).manual_seed(n)
            x = (torch.randn(n, fin, device=DEV, generator=gen) * 0.4)
            gy = torch.randn(n, fout, device=DEV, generator=gen)
            xa = x.clone().requires_grad_(True)
            ya = layer(xa)
            ya.backward(gy)
            want = {k: p.grad.clone() for k, p in layer.named_parameters()}
            layer.zero_grad()
            xb = x.clone().requires_grad_(True)
            yb = rccl.sharded_kan_linear(xb, layer.base_weight, layer.spline_weight, layer.spline_scaler, layer._knots(),
                                         grid, 3, None, comm, row_chunks=chunks)
            yb.backward(gy)
            torch.cuda.synchronize()
            tag = f"n={n} {fin}->{fout} grid {grid} chunks {chunks}"
            if fin <= 64:
                assert torch.equal(ya, yb), tag
                assert torch.equal(xa.grad, xb.grad), tag
                for k, p in layer.named_parameters():
                    assert torch.equal(want[k], p.grad), (tag, k)
            else:       # wider inputs: the forward splits its feature loop by ROW COUNT (few rows -> more splits), so a row chunk
                        # may sum in another order than the whole matrix -- same values to rounding
                assert_close(yb, ya.detach(), tol=2e-6, what=tag + " y")
                assert_close(xb.grad, xa.grad, tol=2e-6, what=tag + " gx")
                for k, p in layer.named_parameters():
                    assert_close(p.grad, want[k], tol=2e-6, what=tag + " " + k)
        # the sharded layer on that transport, against the oracle
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        created = not dist.is_initialized()
        if created:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        try:
            from kagnn_amd.sharded import ShardedGIKANLayer
            n, e, f = 20000, 200000, 64
            ei = orc.powerlaw_graph(n, e, seed=6)
            gen = torch.Generator().manual_seed(6)
            x = torch.randn(n, f, generator=gen) * 0.25
            gy = torch.randn(n, f, generator=gen)
            conv = kagnn_amd.GIKANLayer(f, f, grid_size=5, spline_order=3, hidden_dim=f, nb_layers=2)
            layers = [{k: v.detach().clone() for k, v in l.state_dict().items()} for l in conv.nn.layers]
            y_ref, gx_ref, g_ref = orc.kan_gin_layer_fwd_bwd(x, ei, layers, 3, gy)
            graph = ops.GraphIndex(ei.to(DEV), n)
            for chunks in (1, 3):
                s = ShardedGIKANLayer(conv, None, chunks=chunks, comm="rccl_c").to(DEV)
                for _ in range(2):                       # twice: the communicator and its side stream are reused
                    s.zero_grad()
                    xs = x.to(DEV).requires_grad_(True)
                    y = s(xs, graph)
                    y.backward(gy.to(DEV))
                assert_close(y, y_ref, what=f"rccl_c sharded y (chunks={chunks})")
                assert_close(xs.grad, gx_ref, what=f"rccl_c sharded gx (chunks={chunks})")
                for li, layer in enumerate(s.layers):
                    for k in ("base_weight", "spline_weight", "spline_scaler"):
                        assert_close(getattr(layer, k).grad, g_ref[li][k], what=f"rccl_c L{li}.{k} (chunks={chunks})")
        finally:
            if created:
                dist.destroy_process_group()
    finally:
        comm.close()


def test_activations_beyond_4gib_stay_on_the_split_kernels():
    """x [20M, 64] fp32 is 5.1 GB: byte offsets no longer fit 32 bits.  The split kernels re-open their buffer windows
    per workgroup tile, so the layer must (a) not fall back to the fp32 kernels and (b) treat the far rows exactly
    like the near ones.  Rows are independent in the forward and the input gradient (compare slices against the same
    kernels on a small tensor); the weight gradient is a sum over rows (compare with the sum over two halves)."""
    n = 20_000_000
    torch.manual_seed(11)
    layer = kagnn_amd.KANLinear(64, 64, grid_size=5, spline_order=3).to(DEV)
    layer.precision = ops.PREC_SPLIT
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, 64, device=DEV, generator=gen).mul_(0.6).requires_grad_(True)
    gy = torch.randn(n, 64, device=DEV, generator=gen)
    assert x.numel() * 4 > (1 << 32) and ops._fits32(x.detach(), 64)      # no fp32 fallback for size
    timer = ops.EntryPointTimer()
    ops.set_timer(timer)
    y = layer(x)
    y.backward(gy)
    ops.set_timer(None)
    torch.cuda.synchronize()
    assert {"kagnn_kan_linear_fwd", "kagnn_kan_linear_bwd_input", "kagnn_kan_linear_bwd_weight"} <= set(timer.summary())
    full = {k: p.grad.clone() for k, p in layer.named_parameters()}
    # slices across the 4 GiB boundary (row 16,777,216 is byte 2^32) and at both ends
    for lo in (0, (1 << 24) - 700, n - 1500):
        hi = lo + 1500
        xs = x.detach()[lo:hi].clone().requires_grad_(True)
        layer.zero_grad()
        ys = layer(xs)
        ys.backward(gy[lo:hi].clone())
        assert torch.equal(ys, y.detach()[lo:hi]), lo           # same kernels, same per-row arithmetic
        assert_close(x.grad[lo:hi], xs.grad, tol=1e-6, what=f"gx rows {lo}..")   # per-row scale exponents are row-local
    del y
    parts = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in full.items()}
    half = n // 2
    for lo, hi in ((0, half), (half, n)):
        xs = x.detach()[lo:hi]
        layer.zero_grad()
        layer(xs.requires_grad_(False)).backward(gy[lo:hi])
        for k, p in layer.named_parameters():
            parts[k] += p.grad.double()
    for k in full:
        assert_close(full[k], parts[k], tol=1e-4, what=f"{k} gradient = sum over halves")


def test_aggregation_and_batchnorm_beyond_4gib():
    """the same 5.1 GB activation through the neighbour aggregation (both directions) and BatchNorm: column
    checksums in fp64 and sampled rows against torch on the device"""
    n, e = 20_000_000, 5_000_000
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(n, 64, device=DEV, generator=gen)
    src = torch.randint(0, n, (e,), device=DEV, generator=gen)
    dst = torch.randint(0, n, (e,), device=DEV, generator=gen)
    dst[: e // 2] = dst[: e // 2] // 2 + n // 2             # half of the messages land beyond byte 2^32
    gi = ops.GraphIndex(torch.stack([src, dst]), n)
    for transposed, (a, b) in ((False, (src, dst)), (True, (dst, src))):
        out = ops._aggregate_raw(x, gi, transposed, 1.0, None, None, None, None, False)
        want = x.double().sum(0) + x.index_select(0, a).double().sum(0)
        assert_close(out.double().sum(0), want, 1e-6, what=f"column checksum (transposed={transposed})")
        rows = b[torch.randint(0, e, (2000,), device=DEV, generator=gen)]
        rows = torch.cat([rows, torch.tensor([0, n - 1, (1 << 24) - 1, 1 << 24], device=DEV)])
        ref = x[rows].double()
        hit = torch.isin(b, rows)
        ref_full = torch.zeros(n, 1, device=DEV, dtype=torch.float64)      # row -> slot map without an [n, 64] fp64 buffer
        slot = torch.full((n,), -1, device=DEV, dtype=torch.long)
        uniq = torch.unique(rows)
        slot[uniq] = torch.arange(uniq.numel(), device=DEV)
        acc = torch.zeros(uniq.numel(), 64, device=DEV, dtype=torch.float64)
        acc.index_add_(0, slot[b[hit]], x[a[hit]].double())
        assert_close(out[rows].double(), ref + acc[slot[rows]], what=f"sampled rows (transposed={transposed})")
        del out, ref_full
    w, bb = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    xr = x.requires_grad_(True)
    y = ops.batch_norm(xr, w, bb, rm, rv, True, 0.1, 1e-5)
    mean, var = x.detach().double().mean(0), x.detach().double().var(0, unbiased=False)
    for lo in (0, (1 << 24) - 300, n - 1000):
        want = (x.detach()[lo:lo + 1000].double() - mean) / torch.sqrt(var + 1e-5) * w.double() + bb.double()
        assert_close(y[lo:lo + 1000], want, what=f"batchnorm rows {lo}..")
    gyv = torch.ones(1, 64, device=DEV).expand(n, 64)
    y.backward(gyv)                                                         # d/dx of sum(y) is 0 for batch statistics
    assert float(xr.grad.abs().max()) < 1e-3


def test_fullsize_layer_memory_contract_saves_only_the_layer_inputs(big):
    """"saved-for-backward = layer inputs only" (the reference's autograd keeps ~45 GB at this size).
    The KAN-GIN layer at 1M x 64 holds, between forward and backward, h0 (the aggregate = the first KANLinear's input), h1 (the
    second one's input) and the output y -- 3 x N x F x 4 bytes -- plus the weight packs (< 1 MB); no bases, no messages, no
    workspace survives the calls; after the backward only the gradients remain.  (VERDICT r05 missing 6 / next 5.)"""
    _, x, gi = big
    torch.manual_seed(0)
    conv = kagnn_amd.GIKANLayer(F, F, grid_size=5, spline_order=3, hidden_dim=F, nb_layers=2).to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    gy = torch.randn(N, F, generator=torch.Generator().manual_seed(1)).to(DEV)
    for _ in range(2):                                   # warm the allocator / one-time caches (packs, size queries)
        conv(xd, gi).backward(gy)
    xd.grad = None
    conv.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    rows = N * F * 4
    y = conv(xd, gi)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - base
    fwd_peak = torch.cuda.max_memory_allocated() - base
    assert held <= 3 * rows + (2 << 20), f"forward keeps {held / rows:.2f} x N*F*4 (h0, h1, y = 3 allowed)"
    assert fwd_peak <= 4 * rows + (8 << 20), f"forward peak {fwd_peak / rows:.2f} x N*F*4"
    y.backward(gy)
    del y
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated() - base
    peak = torch.cuda.max_memory_allocated() - base
    params = sum(p.numel() * 4 for p in conv.parameters())
    assert after <= rows + 2 * params + (2 << 20), f"after the backward {after / rows:.2f} x N*F*4 remain (x.grad = 1 allowed)"
    # the backward's transients: two gradient matrices in flight + the weight-gradient slabs
    assert peak <= 6 * rows + (64 << 20), f"step peak {peak / rows:.2f} x N*F*4"
    print(f"memory contract at 1M x 64: held {held / 1e9:.3f} GB between fwd and bwd, step peak {peak / 1e9:.3f} GB over the resident "
          f"{base / 1e9:.3f} GB (reference CPU path: 45.1 GB RSS)")
