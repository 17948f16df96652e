"""GPU box (ONE MI355X): per-rank kernel times of the two sharded layers of kagnn_amd/sharded.py at P = 1, 2, 4, 8 ranks, for the
headline layer (F = 64, grid 5) and BASELINE config 3's (F = 128, grid 8), N = 1M rows / E = 10M edges -- the compute column of
DESIGN.md section 6's table, regenerated from HEAD.  The wire terms are modelled from the byte counts
(`--link-gbs`, GB/s per xGMI link and direction; 7 links per GPU, all used at once by the direct exchanges).  Writes one JSON
document (gpurun_out/shard_plan.json) and prints the markdown table.

    feature-sharded  (north_star's scheme): per rank  aggregation of [N, F/P] both ways + per KANLinear fwd / dX / dW at in = F/P
    transposed                            : per rank  aggregation of [N, F/P] both ways + the whole chain on N/P rows, full width
"""
import json, os, sys, time
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


def timed(fn, reps=10):
    """device time per call: the library's own stage timer (HIP events around every kernel stage) -- wall clock would fold the
    caching allocator's occasional hipMalloc / hipFree stalls of this many-shapes script into single configurations"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with ops.LibraryStageTimer(None):
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    return sum(v["total_ms"] for v in ops.LibraryStageTimer.collect().values()) / reps


def timed_torch(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


graph = ops.GraphIndex(powerlaw_graph(n, e).to(dev), n)
out = {"link_GBs_per_direction": LINK, "links_per_gpu": 7, "N": n, "E": e, "workloads": {}}
for name, F, grid in (("headline", 64, 5), ("config3", 128, 8)):
    rows = []
    full = kagnn_amd.GIKANLayer(F, F, grid_size=grid, spline_order=3, hidden_dim=F, nb_layers=2).to(dev)
    xf = (torch.randn(n, F, device=dev) * 0.25).requires_grad_(True)
    gyf = torch.randn(n, F, device=dev)

    def one_gpu():
        xf.grad = None
        full.zero_grad()
        full(xf, graph).backward(gyf)
    t1 = timed(one_gpu)
    for P in (1, 2, 4, 8):
        w = F // P
        xs = torch.randn(n, w, device=dev)
        agg = timed(lambda: ops._aggregate_raw(xs, graph, False, 1.0, None, None, None, None, False)) + \
            timed(lambda: ops._aggregate_raw(xs, graph, True, 1.0, None, None, None, None, False))
        # feature-sharded: a KANLinear on the rank's input slice, full output width (partial sums)
        lay = kagnn_amd.KANLinear(w, F, grid_size=grid, spline_order=3).to(dev)
        h = (torch.randn(n, w, device=dev) * 0.3).requires_grad_(True)
        gy = torch.randn(n, F, device=dev)

        def kan_fb():
            lay.zero_grad(); h.grad = None
            lay(h).backward(gy)
        kan_feat = timed(kan_fb)
        # the RCCL form's rank-major staging passes (one per KANLinear and direction)
        part = torch.randn(n, F, device=dev)
        stage = timed_torch(lambda: part.view(n, P, w).permute(1, 0, 2).contiguous()) if P > 1 else 0.0
        # transposed: the whole chain on N/P rows at full width
        rows_p = -(-n // P)
        chain = kagnn_amd.KAN([F, F, F], grid_size=grid, spline_order=3).to(dev)
        hr = (torch.randn(rows_p, F, device=dev) * 0.3).requires_grad_(True)
        gr = torch.randn(rows_p, F, device=dev)

        def chain_fb():
            chain.zero_grad(); hr.grad = None
            chain(hr).backward(gr)
        kan_rows = timed(chain_fb)
        # wire bytes per rank and step
        rs_bytes = n * F * 4 * (P - 1) / P                      # one reduce-scatter / all-gather of [N, F] partial sums
        a2a_bytes = 2 * n * F * 4 * (P - 1) / P / P            # the two all-to-alls of one direction: 2 x N*F*4*(P-1)/P^2
        links = min(P - 1, 7)
        wire_feat = 4 * rs_bytes / (links * LINK * 1e9) * 1e3 if P > 1 else 0.0          # 2 KANLinear x (fwd RS + bwd AG), all links at once
        wire_tr = 2 * a2a_bytes / (links * LINK * 1e9) * 1e3 if P > 1 else 0.0           # fwd pair + bwd pair
        comp_feat = agg + 2 * kan_feat
        comp_tr = agg + kan_rows
        rows.append({"P": P, "aggregation_ms": agg, "kanlinear_in_slice_fwd_bwd_ms": kan_feat, "rccl_staging_pass_ms": stage,
                     "chain_on_row_shard_ms": kan_rows,
                     "feature_sharded": {"compute_ms": comp_feat, "wire_ms": wire_feat, "no_overlap_ms": comp_feat + wire_feat,
                                         "full_overlap_ms": max(comp_feat, wire_feat), "rccl_staging_ms": 4 * stage},
                     "transposed": {"compute_ms": comp_tr, "wire_ms": wire_tr, "no_overlap_ms": comp_tr + wire_tr}})
        print(name, rows[-1], flush=True)
        del lay, h, gy, chain, hr, gr, part, xs
        torch.cuda.empty_cache()
    out["workloads"][name] = {"F": F, "grid": grid, "one_gpu_layer_ms": t1, "ranks": rows}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/shard_plan.json", "w"), indent=1)
for name, wl in out["workloads"].items():
    print(f"\n{name}: one GPU {wl['one_gpu_layer_ms']:.2f} ms per layer fwd+bwd; {LINK:.0f} GB/s per link and direction")
    print("| P | per-rank compute (feature-sharded) | + wire, no overlap | full overlap | speed-up range | per-rank compute (transposed) | + wire | speed-up |")
    print("|---|---|---|---|---|---|---|---|")
    for r in wl["ranks"]:
        f, t = r["feature_sharded"], r["transposed"]
        print(f"| {r['P']} | {f['compute_ms']:.2f} | {f['no_overlap_ms']:.2f} | {f['full_overlap_ms']:.2f} | "
              f"{wl['one_gpu_layer_ms'] / f['no_overlap_ms']:.2f}-{wl['one_gpu_layer_ms'] / f['full_overlap_ms']:.2f}x | "
              f"{t['compute_ms']:.2f} | {t['no_overlap_ms']:.2f} | {wl['one_gpu_layer_ms'] / t['no_overlap_ms']:.2f}x |")
