#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: edges/sec of ONE KAN-GIN conv layer
(sum-aggregate + KAN([64,64,64]), grid 5, order 3) forward AND backward on the synthetic
power-law graph of SURVEY.md 8(d) (1M nodes / 10M edges, fp32), plus the roofline of the
dominant kernel and the CPU baseline (oracle, "port") on a bounded sample.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: the layer is sharded over the ranks (aggregation on feature-column shards, KAN chain on row
shards, RCCL all-to-all in between; KAGNN_SHARDING=feature shards the spline coefficient tensor instead --
see kagnn_amd/sharded.py and DESIGN.md) -- total work is fixed => "scaling": "strong".
One JSON line on stdout (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3
F16_MFMA_PEAK_TF = 2500.0


def layer_bytes(n, e, f):
    """Algorithmic bytes of one KAN-GIN layer fwd+bwd, SURVEY.md 8(d): E(8F+8) + N(52F+8)."""
    return e * (8 * f + 8) + n * (52 * f + 8)


def agg_bytes(n, e, f):
    """one aggregation launch: col idx 4E + rowptr 4N + gathered rows 4F*E + self rows 4F*N + write 4F*N."""
    return e * (4 * f + 4) + n * (8 * f + 4)


def kan_flops(n, fin, fout, c):
    return 2.0 * n * fin * (c + 1) * fout


def powerlaw_graph(num_nodes, num_edges, seed=0):
    """The synthetic graph recipe of SURVEY.md 8(d): dst = perm[floor(N u^2)] (in-degree of rank r ~ r^-1/2), src
    uniform, duplicates and self loops kept, unsorted.  (Same recipe as oracle/kan_oracle.py:powerlaw_graph, which the
    tests use; restated here so that the measured path imports nothing from oracle/.)"""
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(num_nodes, generator=g)
    u = torch.rand(num_edges, generator=g, dtype=torch.float64)
    dst = perm[torch.floor(num_nodes * u * u).long().clamp_(max=num_nodes - 1)]
    src = torch.randint(0, num_nodes, (num_edges,), generator=g)
    return torch.stack([src, dst])


def cpu_baseline(n_sample, e_sample, f, grid, order, seed=0):
    """The reference's algorithm (oracle/kan_oracle.py: dense bases, F.linear, index_select +
    scatter_add_, stock autograd) on the host cores, bounded sample of the same workload.  torch's CPU
    elementwise kernels stop scaling (and then regress badly) beyond a few dozen threads, so a few thread
    counts are tried and the best one is reported -- `cores` is the thread count actually used."""
    from oracle import kan_oracle as orc
    ei = orc.powerlaw_graph(n_sample, e_sample, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n_sample, f, generator=g) * 0.25
    layers = [orc.init_kan_linear(f, f, grid, order, g) for _ in range(2)]
    host = os.cpu_count() or 1
    best, best_threads, tried = float("inf"), 1, []
    for th in sorted({min(8, host), min(32, host)}):
        torch.set_num_threads(th)
        if not tried:
            orc.kan_gin_layer_fwd_bwd(x, ei, layers, order)          # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        orc.kan_gin_layer_fwd_bwd(x, ei, layers, order)
        dt = time.perf_counter() - t0
        tried.append((th, round(dt, 2)))
        if dt < best:
            best, best_threads = dt, th
    return {"value": e_sample / best, "unit": "edges/s", "cores": best_threads, "kind": "port",
            "sample": f"same recipe at N={n_sample}, E={e_sample}, F={f}, grid={grid}, order={order}; 1 warm-up, then one "
                      f"timed fwd+bwd per thread count {tried} (threads, s) on a {host}-thread host; best reported"}


def secondary_figures(dev, conv, graph, x, n, e, f, grid, order):
    """SURVEY.md 8(d)'s side figures, measured after (never inside) the timed region: what a device-to-device copy
    reaches on this box next to the 8 TB/s the roofline is priced against, the share of the first KANLinear's inputs
    that falls inside the spline support, and the full 3-layer GKAN_Nodes training step quoted as 3E / t_step."""
    import kagnn_amd
    from kagnn_amd import harness, ops
    out = {}
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)       # 1 GiB, well past the 256 MB of MALL
    b = torch.empty_like(a)
    b.copy_(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        b.copy_(a)
    ev1.record()
    torch.cuda.synchronize()
    out["hbm_copy_GBs"] = 10 * 2.0 * a.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del a, b
    with torch.no_grad():
        h0 = ops.aggregate_sum(x.detach(), graph, self_scale=1.0)
        knots = conv.nn.layers[0].grid[0]
        out["in_support_frac"] = float(((h0 >= knots[0]) & (h0 < knots[-1])).float().mean())
        del h0
    torch.manual_seed(0)
    classes = 40
    model = kagnn_amd.GKAN_Nodes("gin", 3, f, f, classes, skip=True, grid_size=grid, spline_order=order,
                                 hidden_layers=2).to(dev)
    y = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(2)).to(dev)
    mask = torch.ones(n, dtype=torch.bool, device=dev)
    t_step, _ = harness.time_model(model, x.detach(), graph, y, mask, nb_epochs=5, warmup=2)
    out["model_step"] = {"what": f"GKAN_Nodes(gin, 3 conv layers, hidden {f}, {classes} classes, skip, BatchNorm) training step "
                                 "(forward, softmax + cross-entropy, backward, Adam), time_model.py:35-48",
                         "ms_per_step": t_step * 1e3, "edges_per_s": 3 * e / t_step}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--grid", type=int, default=5)
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("KAGNN_PRECISION", "split"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="nodes in the CPU-baseline sample")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (copy bandwidth, full model step)")
    args = ap.parse_args()
    os.environ["KAGNN_PRECISION"] = args.precision

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # debugging aid for boxes with fewer GPUs than ranks: KAGNN_BENCH_BACKEND=gloo puts every rank on cuda:0 and
    # moves the collectives to gloo (numbers measured that way are meaningless; the launch contract uses RCCL)
    backend = os.environ.get("KAGNN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import kagnn_amd
    from kagnn_amd import ops

    n, e, f = args.nodes, args.edges, args.hidden
    ei = powerlaw_graph(n, e, seed=0).to(dev)
    gen = torch.Generator().manual_seed(0)
    x_full = torch.randn(n, f, generator=gen) * 0.25
    gy_full = torch.Generator().manual_seed(1)
    gy_full = torch.randn(n, f, generator=gy_full)
    torch.manual_seed(0)
    conv = kagnn_amd.GIKANLayer(f, f, grid_size=args.grid, spline_order=args.order, hidden_dim=f, nb_layers=2)

    graph = ops.GraphIndex(ei, n)
    if world == 1:
        conv = conv.to(dev)
        x = x_full.to(dev).requires_grad_(True)
        gy = gy_full.to(dev)

        def step():
            x.grad = None
            for p in conv.parameters():
                p.grad = None
            y = conv(x, graph)
            y.backward(gy)
    else:
        # default: aggregation on column shards, KAN chain on row shards, two all-to-alls per direction
        # (kagnn_amd/sharded.py); KAGNN_SHARDING=feature selects the reduce-scatter / all-gather variant that
        # shards the spline coefficient tensor itself (8x more wire traffic at this width)
        from kagnn_amd.sharded import ShardedGIKANLayer, TransposedShardedGIKANLayer
        sharding = os.environ.get("KAGNN_SHARDING", "transposed")
        cls = ShardedGIKANLayer if sharding == "feature" else TransposedShardedGIKANLayer
        sconv = (cls(conv, dist.group.WORLD, sync_in_backward=False) if cls is TransposedShardedGIKANLayer
                 else cls(conv, dist.group.WORLD)).to(dev)
        x = sconv.shard_columns(x_full.to(dev)).requires_grad_(True)
        gy = sconv.shard_columns(gy_full.to(dev))

        def step():
            x.grad = None
            for p in sconv.parameters():
                p.grad = None
            y = sconv(x, graph)
            y.backward(gy)
            if hasattr(sconv, "sync_gradients"):
                sconv.sync_gradients()             # one flat all-reduce for all weight gradients

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # per-entry-point breakdown: three extra untimed steps with HIP events around every library call.  In the
    # timed region only the dominant entry point keeps its events (the roofline figure is measured there, live);
    # timing all ~25 calls of a step adds ~0.07 ms of event records to a 2.7 ms step.
    PROFILE_STEPS = 3
    warm_timer = ops.EntryPointTimer()
    ops.set_timer(warm_timer)
    sync()
    for _ in range(PROFILE_STEPS):
        step()
    sync()
    ops.set_timer(None)
    warm = warm_timer.summary()
    only = max(warm, key=lambda k: warm[k]["total_ms"]) if warm else None
    timer = ops.EntryPointTimer(only=only)
    ops.set_timer(timer)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / args.steps * 1e3
    value = e / (dt / args.steps)

    if rank == 0:
        prof = timer.summary()
        c = args.grid + args.order
        fl = f // world if world > 1 else f
        # dominant entry point of the layer (largest device time per step); its launches were timed in the timed region
        per_step = {k: v["total_ms"] / PROFILE_STEPS for k, v in warm.items()}
        dom = only if only in prof else max(per_step, key=per_step.get)
        avg_ms = prof[dom]["avg_ms"] if dom in prof else warm[dom]["avg_ms"]
        if dom == "kagnn_aggregate_sum":
            b = agg_bytes(n, e, fl)
            roof = {"kernel": dom, "bound": "hbm", "achieved": b / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "algorithmic_bytes_per_launch": b}
        else:
            fl_ops = kan_flops(n, fl, f, c)
            peak = FP32_MFMA_PEAK_TF if args.precision in ("fp32", "exact", "0") else F16_MFMA_PEAK_TF
            eff = fl_ops if peak == FP32_MFMA_PEAK_TF else 3.0 * fl_ops     # 3 MFMA products per fp32 product
            roof = {"kernel": dom, "bound": "mfma", "achieved": eff / (avg_ms * 1e-3) / 1e12, "peak": peak,
                    "unit": "TFLOP/s", "algorithmic_flops_per_launch": fl_ops}
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["avg_launch_ms"] = avg_ms
        roof["traffic"] = None
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr):
            try:
                roof["traffic"] = json.load(open(tr)).get(dom)
            except Exception:
                pass
        layer_gbs = layer_bytes(n, e, f) / (ms * 1e-3) / 1e9
        out = {
            "metric": "edges/sec KAN-GIN fwd+bwd, hidden=64 grid=5, 1M-node synthetic; HBM % peak",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision in ("fp32", "exact", "0") else "f32 (fp16 hi/lo split operands, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"KAN-GIN conv layer fwd+bwd (aggregate + KAN([{f},{f},{f}]) grid={args.grid} order={args.order}), "
                                   f"power-law graph N={n} E={e} seed 0 (SURVEY 8(d))",
                       "nodes": n, "edges": e, "hidden": f, "grid_size": args.grid, "spline_order": args.order,
                       "precision": args.precision,
                       "parallelism": "single GPU" if world == 1 else (
                           f"feature-sharded x{world} (RCCL reduce-scatter/all-gather)" if os.environ.get("KAGNN_SHARDING") == "feature"
                           else f"column-sharded aggregation + row-sharded KAN chain x{world} (RCCL all-to-all, weight-gradient all-reduce)")},
            "layer_algorithmic_bytes": layer_bytes(n, e, f),
            "layer_hbm_GBs": layer_gbs, "layer_hbm_frac": layer_gbs / HBM_PEAK_GBS,
            "roofline": roof,
            "entry_points_ms_per_step": per_step, "entry_points_measured_in": "3 extra untimed steps after the warm-up (HIP events around every call)",
        }
        if not args.no_extras and world == 1:
            out["secondary"] = secondary_figures(dev, conv, graph, x, n, e, f, args.grid, args.order)
        if not args.no_cpu_baseline and world == 1:
            ns = min(args.cpu_sample, n)
            out["cpu_baseline"] = cpu_baseline(ns, ns * (e // n if n else 10), f, args.grid, args.order)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
