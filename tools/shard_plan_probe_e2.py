"""GPU box (ONE MI355X): per-rank device time of the round-6 sharded paths (kagnn_amd/sharded.py: ShardedGIFASTKANLayer, ShardedNodeModel)
at P = 1, 2, 4, 8 -- every kernel a rank runs, on its column slice, timed with the library's stage timer -- plus the wire terms from the byte
counts at `--link-gbs` GB/s per xGMI link and direction (7 links per GPU, used at once by a direct exchange).  N = 1M rows / E = 10M edges.
The per-rank projection DESIGN.md section 6 quotes for SURVEY 8(e)'s other exchanges.  Writes gpurun_out/shard_plan_e2.json.

    FastKAN-GIN layer (F = 64, 8 grids): aggregation of [N, F/P] both ways; per FastKANLayer: local row moments, merge of the gathered
        [P, N, 2], forward on the slice (partial sums [N, F]), backward halves on the gathered gradient, LayerNorm finish
    GKAN_Nodes step (3 x KAN-GIN conv hidden 64 grid 5, BatchNorm, skip read-out 256 -> 40): the three sharded convolutions' local kernels,
        BatchNorm1d fwd + bwd on [N, 64/P], the read-out KANLinear on the rank's 256/P input columns (all 40 outputs), loss on the full logits
"""
import json, os, sys
import torch
sys.path.insert(0, os.getcwd())
import kagnn_amd
from kagnn_amd import ops

LINK = float(next((a.split("=")[1] for a in sys.argv if a.startswith("--link-gbs=")), "50"))
dev = "cuda"
n, e = 1_000_000, 10_000_000


def powerlaw_graph(num_nodes, num_edges, seed=0):
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(num_nodes, generator=g)
    u = torch.rand(num_edges, generator=g, dtype=torch.float64)
    dst = perm[torch.floor(num_nodes * u * u).long().clamp_(max=num_nodes - 1)]
    src = torch.randint(0, num_nodes, (num_edges,), generator=g)
    return torch.stack([src, dst])


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    with ops.LibraryStageTimer(None):
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    return sum(v["total_ms"] for v in ops.LibraryStageTimer.collect().values()) / reps


def timed_events(fn, reps=6):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


graph = ops.GraphIndex(powerlaw_graph(n, e).to(dev), n)
def wire(nbytes, P):
    """ms for a rank to move `nbytes` to / from its P - 1 peers, one xGMI link per peer (direct exchange), all links at once"""
    return 0.0 if P == 1 else nbytes / (P - 1) / (LINK * 1e9) * 1e3
out = {"link_GBs_per_direction": LINK, "N": n, "E": e, "fastkan_layer": [], "gkan_nodes_step": []}

# ------------------------------------------------------------------ FastKAN-GIN layer, F = 64, 8 grids
F, NG = 64, 8
torch.manual_seed(0)
full = kagnn_amd.GIFASTKANLayer(F, F, grid_size=NG, hidden_dim=F, nb_layers=2).to(dev)
xf = (torch.randn(n, F, device=dev) * 0.25)
gyf = torch.randn(n, F, device=dev)
one = timed_events(lambda: full(xf.requires_grad_(True), graph).backward(gyf))
for P in (1, 2, 4, 8):
    w = F // P
    xs, gs = xf[:, :w].contiguous(), gyf[:, :w].contiguous()
    t_agg = 2 * timed(lambda: ops.aggregate_sum(xs, graph, self_scale=1.0))
    t_kan = t_ln = 0.0
    for l in full.nn.layers:
        lw, lb = l.layernorm.weight[:w].contiguous(), l.layernorm.bias[:w].contiguous()
        sw = l.spline_linear.weight.view(F, F, NG)[:, :w].reshape(F, w * NG).contiguous()
        bw = l.base_linear.weight[:, :w].contiguous()
        mom = ops.fastkan_row_moments(xs)
        gathered = mom.unsqueeze(0).repeat(P, 1, 1).contiguous()
        stats = ops.fastkan_merge_moments(gathered, w, 1e-5)
        t_ln += timed(lambda: ops.fastkan_row_moments(xs)) + timed(lambda: ops.fastkan_merge_moments(gathered, w, 1e-5))
        t_kan += timed(lambda: ops.fastkan_shard_fwd(xs, stats, lw, lb, sw, bw, l.base_linear.bias, l.rbf.grid, l.rbf.denominator))

        def bwd():
            st, sums, *_ = ops.fastkan_shard_bwd(xs, gyf, stats, lw, lb, sw, bw, l.rbf.grid, l.rbf.denominator, want_bias=True)
            ops.fastkan_shard_bwd_finish(st, sums, F)
        t_kan += timed(bwd)
    frac = (P - 1) / P
    exch = 4 * n * F * 4 * frac                               # 2 layers x (reduce-scatter fwd + all-gather bwd) of [N, F] partial sums
    ln = 2 * (8 * n * (P - 1) + 2 * 8 * n * frac)             # per layer: all-gather of P x [N, 2] fwd, all-reduce of [N, 2] bwd
    row = {"P": P, "aggregation_ms": t_agg, "fastkan_layers_on_slice_ms": t_kan, "layernorm_exchange_kernels_ms": t_ln,
           "compute_ms": t_agg + t_kan + t_ln, "wire_partial_sums_ms": wire(exch, P), "wire_layernorm_ms": wire(ln, P)}
    row["no_overlap_ms"] = row["compute_ms"] + row["wire_partial_sums_ms"] + row["wire_layernorm_ms"]
    row["full_overlap_ms"] = max(row["compute_ms"], row["wire_partial_sums_ms"]) + row["wire_layernorm_ms"]
    out["fastkan_layer"].append(row)
    print("fastkan", row, flush=True)
out["fastkan_layer_one_gpu_ms"] = one

# ------------------------------------------------------------------ GKAN_Nodes training step, hidden 64, grid 5, 40 classes
H, G, C = 64, 5, 40
torch.manual_seed(0)
model = kagnn_amd.GKAN_Nodes("gin", 3, H, H, C, skip=True, grid_size=G, spline_order=3, hidden_layers=2).to(dev).train()
labels = torch.randint(0, C, (n,), device=dev)
xm = torch.randn(n, H, device=dev) * 0.25
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)


def full_step():
    opt.zero_grad(set_to_none=True)
    ops.softmax_cross_entropy(model(xm, graph), labels, None, pre_softmax=True).backward()
    opt.step()
one_m = timed_events(full_step)
for P in (1, 2, 4, 8):
    w = H // P
    xs = xm[:, :w].contiguous()
    gfull = torch.randn(n, H, device=dev)
    conv = model.convs[0]
    t_conv = 0.0
    t_agg = 2 * timed(lambda: ops.aggregate_sum(xs, graph, self_scale=1.0))
    for l in conv.nn.layers:
        bwt, swt, sct = l.base_weight[:, :w].contiguous(), l.spline_weight[:, :w].contiguous(), l.spline_scaler[:, :w].contiguous()
        knots = l.grid[0].contiguous()

        def kl():
            xr = xs.detach().requires_grad_(True)
            bw_, sw_, sc_ = bwt.detach().requires_grad_(True), swt.detach().requires_grad_(True), sct.detach().requires_grad_(True)
            ops.kan_linear(xr, bw_, sw_, sc_, knots, G, 3).backward(gfull)
        t_conv += timed(kl)
    bn = kagnn_amd.BatchNorm1d(w).to(dev).train()
    hs = torch.randn(n, w, device=dev)

    def bnf():
        hr = hs.detach().requires_grad_(True)
        bn(hr).backward(hs)
    t_bn = timed(bnf)
    # read-out: this rank's 4 x (64 / P) input columns of [x | h1 | h2 | h3], all 40 outputs
    ro = model.lay_out
    cols = torch.cat([torch.arange(k * H, k * H + w) for k in range(4)]).to(dev)
    rb, rs, rc = ro.base_weight[:, cols].contiguous(), ro.spline_weight[:, cols].contiguous(), ro.spline_scaler[:, cols].contiguous()
    hin = torch.randn(n, 4 * w, device=dev) * 0.5
    glog = torch.randn(n, C, device=dev)

    def readout():
        hr = hin.detach().requires_grad_(True)
        b_, s_, c_ = rb.detach().requires_grad_(True), rs.detach().requires_grad_(True), rc.detach().requires_grad_(True)
        ops.kan_linear(hr, b_, s_, c_, ro.grid[0].contiguous(), G, 3).backward(glog)
    t_ro = timed(readout)
    logits = torch.randn(n, C, device=dev)

    def loss():
        lg = logits.detach().requires_grad_(True)
        ops.softmax_cross_entropy(lg, labels, None, pre_softmax=True).backward()
    t_loss = timed_events(loss)
    frac = (P - 1) / P
    exch = 3 * 4 * n * H * 4 * frac + 2 * n * C * 4 * frac    # 3 convs x 4 exchanges of [N, 64] + the read-out's all-reduce of [N, 40]
    row = {"P": P, "three_convs_ms": 3 * (t_agg + t_conv), "three_norms_ms": 3 * t_bn, "read_out_ms": t_ro, "loss_ms": t_loss,
           "compute_ms": 3 * (t_agg + t_conv + t_bn) + t_ro + t_loss, "wire_ms": wire(exch, P)}
    row["no_overlap_ms"] = row["compute_ms"] + row["wire_ms"]
    row["full_overlap_ms"] = max(row["compute_ms"], row["wire_ms"])
    out["gkan_nodes_step"].append(row)
    print("model", row, flush=True)
out["gkan_nodes_step_one_gpu_ms"] = one_m

os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/shard_plan_e2.json", "w"), indent=1)
print(f"\nFastKAN-GIN layer (F = 64, 8 grids): one GPU {one:.2f} ms per layer fwd+bwd; {LINK:.0f} GB/s per link and direction")
print("| P | aggregation | FastKAN layers on the slice | LayerNorm exchange kernels | per-rank compute | wire: partial sums / LayerNorm | total, no ... full overlap | vs one GPU |")
print("|---|---|---|---|---|---|---|---|")
for r in out["fastkan_layer"]:
    print(f"| {r['P']} | {r['aggregation_ms']:.2f} | {r['fastkan_layers_on_slice_ms']:.2f} | {r['layernorm_exchange_kernels_ms']:.2f} | {r['compute_ms']:.2f} | "
          f"{r['wire_partial_sums_ms']:.2f} / {r['wire_layernorm_ms']:.2f} | {r['no_overlap_ms']:.2f} ... {r['full_overlap_ms']:.2f} | "
          f"{one / r['no_overlap_ms']:.2f}-{one / r['full_overlap_ms']:.2f}x |")
print(f"\nGKAN_Nodes training step (3 conv layers, hidden 64, grid 5, 40 classes): one GPU {one_m:.2f} ms per step (fused default path)")
print("| P | three convolutions | three norms | read-out | loss | per-rank compute | wire | total, no ... full overlap | vs one GPU |")
print("|---|---|---|---|---|---|---|---|---|")
for r in out["gkan_nodes_step"]:
    print(f"| {r['P']} | {r['three_convs_ms']:.2f} | {r['three_norms_ms']:.2f} | {r['read_out_ms']:.2f} | {r['loss_ms']:.2f} | {r['compute_ms']:.2f} | {r['wire_ms']:.2f} | "
          f"{r['no_overlap_ms']:.2f} ... {r['full_overlap_ms']:.2f} | {one_m / r['no_overlap_ms']:.2f}-{one_m / r['full_overlap_ms']:.2f}x |")
