#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: edges/sec of ONE KAN-GIN conv layer
(sum-aggregate + KAN([F,F,F])) forward AND backward on the synthetic power-law graph of
SURVEY.md 8(d) (1M nodes / 10M edges, fp32 I/O).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|config3|fastkan|model]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One JSON line on stdout (rank 0).  Besides the contract's fields it carries
  layer_hbm_frac        the headline roofline figure: B_layer / t_step / 8 TB/s
  roofline              the dominant entry point (largest device time per step), HIP events in the timed region
  roofline_kernels      the four hot kernels (aggregation, KAN forward, dX, dW), each against BOTH its HBM bytes and
                        its matrix-core flops -- the limiting ones are the KAN kernels, not the aggregation
  fp32_mode_ms_per_step the same step on the exact-fp32 MFMA kernels (KAGNN_PRECISION=fp32)
  cpu_baseline          the oracle (the reference's algorithm, "port") on the host cores, bounded sample
The timed path is the product's default: ONE library call per convolution each way (kagnn_gin_kan_layer_fwd / _bwd); the
dominant kernel is timed live inside it by the library's stage timer (kagnn_stage_timer_*: HIP events on the launch stream).
N > 1 (one rank per GPU): a probe runs the combinations {feature-sharded (north_star's scheme), column/row transposed} x
{RCCL through torch.distributed, direct peer-to-peer kernels} + the feature-sharded layer on the library's own RCCL entry
points (include/kagnn_rccl.h), each under the full contract (W warm-up + K timed steps, max over ranks), plain RCCL first;
`value` is NORTH_STAR'S SCHEME -- the faster of feature/rccl and feature/rccl_c (spline coefficients sharded by input feature,
RCCL exchange; named in config.parallelism), re-timed with the stage timer on -- so the driver's scaling curve never changes
partitioning silently ("value_scheme_is_north_star": true; false only when neither of the two ran, in which case the fastest of the
rest stands in and says so).  The fastest combination outside that set sits beside it under "fastest_alternative"; all are listed
under "multi_gpu_probe".  Total work is fixed => "scaling": "strong".  "complete" is false on an interim line (see below).
The line survives a transport that hangs or kills a rank (none of the N > 1 paths has ever run on more than one device):
rank 0's line travels through a forked reporter process that prints the LAST line it was handed when rank 0 ends -- the
complete one, or the interim one written after the last combination that finished -- and every phase runs under a
watchdog (KAGNN_BENCH_PHASE_TIMEOUT seconds, default 90; 300 for the first, which includes RCCL's start-up).
`--workload fastkan` / `--workload model` at N > 1 (round 6): the feature-sharded FastKAN-GIN layer (LayerNorm exchange of 2 floats
per row each way) and the whole GKAN_Nodes training step on column shards (kagnn_amd.sharded.ShardedGIFASTKANLayer /
ShardedNodeModel), north_star's scheme over RCCL only; same contract, reporter and watchdog.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import resource
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3
F16_MFMA_PEAK_TF = 2500.0

WORKLOADS = {
    # name: (hidden, grid) -- N / E / order come from the flags (defaults 1M / 10M / 3)
    "headline": (64, 5),       # BASELINE.json metric: hidden=64 grid=5
    "config3": (128, 8),       # BASELINE.json configs[2]: hidden=128 grid=8 (the 8-GPU config)
    "fastkan": (64, 8),        # the RBF-basis twin of the headline layer (BASELINE.json configs[4]'s kernel path): FastKAN-GIN,
                               # hidden 64, 8 grids -- BASELINE.md section 2 holds the reference CPU time of this very layer
    "model": (64, 5),          # SURVEY 8(d)'s secondary figure as a workload of its own: the full GKAN_Nodes(gin, 3 conv layers, hidden
                               # 64, 40 classes, skip, BatchNorm) TRAINING step (forward, softmax + CE, backward, Adam), value = 3E / t;
                               # N > 1: kagnn_amd.sharded.ShardedNodeModel (column shards, shard-local BatchNorm, sharded read-out)
}


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


# ---------------------------------------------------------------------------------------------- CPU baseline
def _physical_cores():
    try:
        seen = set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((pid, line.split(":")[1].strip()))
        if seen:
            return len(seen)
    except Exception:
        pass
    return os.cpu_count() or 1


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def cpu_baseline(n_sample, e_sample, f, grid, order, seed=0, full=False, arch="kan"):
    """The reference's algorithm (oracle/kan_oracle.py: dense bases, F.linear, index_select + scatter_add_, stock
    autograd) on the host cores, on a bounded sample of the same workload (same graph recipe, 1/10 of the nodes and
    edges by default: ~4 s and ~5 GB per pass; the full 1M / 10M layer needs ~45 GB and ~40 s per pass -- `--cpu-full`
    runs it when MemAvailable allows).  torch's CPU elementwise kernels stop scaling beyond a few dozen threads, so
    thread counts up to the physical core count are swept, 1 warm-up + best of 3 each; `cores` = the thread count of
    the best run."""
    from oracle import kan_oracle as orc
    ei = orc.powerlaw_graph(n_sample, e_sample, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n_sample, f, generator=g) * 0.25
    if arch == "fastkan":
        import math

        def init_fastkan(fi, fo):            # the reference's shapes and scales (fastkan.py:60-75); RNG stream not matched
            return {"layernorm.weight": torch.ones(fi), "layernorm.bias": torch.zeros(fi),
                    "rbf.grid": torch.linspace(-2.0, 2.0, grid),
                    "spline_linear.weight": torch.randn(fo, fi * grid, generator=g) * 0.1,
                    "base_linear.weight": (torch.rand(fo, fi, generator=g) * 2 - 1) / math.sqrt(fi),
                    "base_linear.bias": (torch.rand(fo, generator=g) * 2 - 1) / math.sqrt(fi)}
        fk = [init_fastkan(f, f) for _ in range(2)]

        def layer_pass():
            xr = x.clone().requires_grad_(True)
            ps = [{k: (v.clone().requires_grad_(True) if k != "rbf.grid" else v) for k, v in p.items()} for p in fk]
            orc.gin_conv(xr, ei, lambda h: orc.fastkan_forward(h, ps)).sum().backward()
    else:
        layers = [orc.init_kan_linear(f, f, grid, order, g) for _ in range(2)]

        def layer_pass():
            orc.kan_gin_layer_fwd_bwd(x, ei, layers, order)
    host, phys = os.cpu_count() or 1, _physical_cores()
    sweep = sorted({min(t, host) for t in (8, 32, 64, phys)})
    if full:
        sweep = [min(32, host)]                                      # (the sample's best thread count on these hosts)
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    best, best_threads, tried = float("inf"), 1, []
    for th in sweep:
        torch.set_num_threads(th)
        layer_pass()                                                 # warm-up (allocator, thread pool)
        runs = []
        for _ in range(1 if full else 3):
            t0 = time.perf_counter()
            layer_pass()
            runs.append(time.perf_counter() - t0)
        tried.append({"threads": th, "best_s": round(min(runs), 3), "runs_s": [round(r, 3) for r in runs]})
        if min(runs) < best:
            best, best_threads = min(runs), th
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    return {"value": e_sample / best, "unit": "edges/s", "cores": best_threads, "kind": "port",
            "seconds": best, "host_threads": host, "physical_cores": phys, "thread_sweep": tried,
            "peak_rss_GB": rss1 / 1e6, "peak_rss_growth_GB": (rss1 - rss0) / 1e6,
            "sample": f"same recipe at N={n_sample}, E={e_sample}, F={f}, grid={grid}, order={order} "
                      f"({'FULL size' if full else 'bounded sample of the 1M/10M workload'}); per thread count 1 warm-up + best of "
                      f"{1 if full else 3} timed fwd+bwd passes; best thread count reported"}


# ---------------------------------------------------------------------------------------------- PMC traffic
def _profiler_active():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre.lower() or any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ)


def _pmc_pass(counter, argv, timeout_s):
    """one `rocprofv3 --pmc <counter>` pass over a short run of this script; per-kernel average of the counter"""
    out = tempfile.mkdtemp(prefix="kagnn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + argv
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
        agg = {}
        for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
            cur = sqlite3.connect(db).cursor()
            cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            if "kernel_name" not in cols:
                continue
            for kn, cn, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
                if cn == counter:
                    a = agg.setdefault(kn, [0, 0.0])
                    a[0] += 1
                    a[1] += v
        for csv in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            import csv as _csv
            for row in _csv.DictReader(open(csv)):
                if row.get("Counter_Name") == counter:
                    a = agg.setdefault(row.get("Kernel_Name", "?"), [0, 0.0])
                    a[0] += 1
                    a[1] += float(row.get("Counter_Value", 0))
        return {k: v[1] / v[0] for k, v in agg.items()}
    except Exception:
        return {}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_traffic(args, kernel_prefix):
    """HBM-side bytes per launch of the dominant kernel from the PMC counters, collected in THIS run (same box, same
    lease): separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE
    costs 3, WRITE_SIZE 2), bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- the x2 is the guide's gfx950 correction for
    16 B/lane reads (FETCH_SIZE tallies 128-B requests at 64 B), which is what the aggregation and the KAN kernels
    issue; WRITE_SIZE is uncalibrated there.  Counts Infinity-Cache hits (fabric-side, not physical HBM)."""
    if shutil.which("rocprofv3") is None or _profiler_active():
        return None, "rocprofv3 unavailable or this process is itself being profiled"
    argv = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-traffic", "--no-fp32",
            "--workload", args.workload, "--nodes", str(args.nodes), "--edges", str(args.edges), "--order", str(args.order),
            "--precision", args.precision, "--act", args.act]
    fetch = _pmc_pass("FETCH_SIZE", argv, 150)
    write = _pmc_pass("WRITE_SIZE", argv, 150)
    per_kernel = {}
    for k in fetch:
        if k in write:
            short = k.split("(")[0].replace("void ", "").replace("kagnn::", "")
            per_kernel[short] = int((2.0 * fetch[k] + write[k]) * 1024)
    hit = [v for k, v in per_kernel.items() if k.startswith(kernel_prefix)]
    return (hit[0] if hit else None), {k: v for k, v in per_kernel.items() if not k.startswith(("at::", "__amd", "rocprim"))}


# ---------------------------------------------------------------------------------------------- secondary
def copy_bandwidth(dev):
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)       # 1 GiB, well past the 256 MB of MALL
    b = torch.empty_like(a)
    b.copy_(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        b.copy_(a)
    ev1.record()
    torch.cuda.synchronize()
    return 10 * 2.0 * a.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9


def secondary_figures(dev, conv, graph, x, n, e, f, grid, order):
    """SURVEY.md 8(d)'s side figures, measured after (never inside) the timed region: the share of the first
    KANLinear's inputs that falls inside the spline support and the full 3-layer GKAN_Nodes training step quoted as
    3E / t_step."""
    import kagnn_amd
    from kagnn_amd import harness, ops
    out = {}
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
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t_step, _ = harness.time_model(model, x.detach(), graph, y, mask, nb_epochs=5, warmup=2)
    model_peak = torch.cuda.max_memory_allocated()
    out["model_step"] = {"what": f"GKAN_Nodes(gin, 3 conv layers, hidden {f}, {classes} classes, skip, BatchNorm) training step "
                                 "(forward, softmax + cross-entropy, backward, Adam), time_model.py:35-48",
                         "ms_per_step": t_step * 1e3, "edges_per_s": 3 * e / t_step,
                         "conv_layers_share": None, "peak_device_GB": model_peak / 1e9}
    return out


def graph_level_step_figures(dev, epochs=10):
    """BASELINE config 4 (the latency-bound regime): the ZINC-shaped mini-batch training step of the reference's
    graph-regression script (graph_regression/optuna_zinc.py:56-66: KAGIN(1, 1, 4 GINE convolutions, hidden 64, embedding encoders),
    256 molecules of 23 +- 5 atoms per batch, L1 loss, Adam) through kagnn_amd.harness.train_graph_batches over 8 distinct batches
    (a CSR per batch): wall ms per step (the step is host-bound: ~0.8 ms of device time)."""
    import kagnn_amd
    from types import SimpleNamespace
    from kagnn_amd import harness
    B, H = 256, 64
    batches = []
    for k in range(8):
        g = torch.Generator().manual_seed(100 + k)
        sizes = torch.randint(18, 29, (B,), generator=g)
        n = int(sizes.sum()); off = torch.cumsum(sizes, 0) - sizes
        src, dst, batch = [], [], []
        for b in range(B):
            nb = int(sizes[b]); eb = 2 * nb + 4
            src.append(torch.randint(0, nb, (eb,), generator=g) + off[b]); dst.append(torch.randint(0, nb, (eb,), generator=g) + off[b])
            batch.append(torch.full((nb,), b))
        e = sum(len(s_) for s_ in src)
        # (the attributes of a torch_geometric Batch: `ptr` -- the node offsets of the graphs -- is part of what its DataLoader collates)
        batches.append(SimpleNamespace(x=torch.randint(0, 21, (n, 1), generator=g).to(dev), edge_index=torch.stack([torch.cat(src), torch.cat(dst)]).to(dev),
                                       edge_attr=torch.randint(0, 4, (e,), generator=g).to(dev), batch=torch.cat(batch).to(dev), num_graphs=B,
                                       ptr=torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)]).to(dev),
                                       y=torch.randn(B, generator=g).to(dev)))
    torch.manual_seed(0)
    m = kagnn_amd.KAGINRegression(1, 1, 4, H, 2, 4, 3, 1, 0.0, True)
    m.atom_encoder = kagnn_amd.graph_models.AtomEncoder(H, [21])
    m.bond_encoder.bond_embedding_list = torch.nn.ModuleList([torch.nn.Embedding(4, H)])
    m = m.to(dev)
    # three repeats of `epochs` epochs (8 steps each; the model keeps training across them): a single 32-step region is ~27 ms of wall
    # clock and one scheduler hiccup moves it by 10 % -- the MEDIAN repeat is the figure, all three are listed
    reps = []
    for _ in range(3):
        t, means = harness.train_graph_batches(m, batches, nb_epochs=epochs, warmup=1)
        reps.append(t)
    t = sorted(reps)[1]
    return {"what": "KAGIN graph-regression training step, 256-molecule mini-batch (~5.9k nodes / ~12.7k edges), 4 GINE(KAN) convolutions, "
                    "hidden 64, grid 4, embedding encoders, L1 loss, Adam; 8 distinct batches, CSR rebuilt per batch (optuna_zinc.py:56-66); "
                    f"median of 3 repeats of {epochs} epochs",
            "ms_per_step": t * 1e3, "repeats_ms_per_step": [r * 1e3 for r in reps], "graphs_per_s": B / t, "edges_per_s": 12700 / t,
            "final_epoch_mean_loss": means[-1]}


def other_layer_figures(dev, graph, n, e, steps=10):
    """The two other layer workloads of this script (`--workload config3`, `--workload fastkan`) timed in the same process,
    after the headline's timed region, on the same graph: ms per fwd+bwd step and the same roofline arithmetic."""
    import kagnn_amd
    out = {}
    for name, (f, grid) in WORKLOADS.items():
        if name in ("headline", "model"):
            continue
        torch.manual_seed(0)
        if name == "fastkan":
            conv = kagnn_amd.GIFASTKANLayer(f, f, grid_size=grid, hidden_dim=f, nb_layers=2).to(dev)
        else:
            conv = kagnn_amd.GIKANLayer(f, f, grid_size=grid, spline_order=3, hidden_dim=f, nb_layers=2).to(dev)
        x = (torch.randn(n, f, generator=torch.Generator().manual_seed(0)) * 0.25).to(dev).requires_grad_(True)
        gy = torch.randn(n, f, generator=torch.Generator().manual_seed(1)).to(dev)
        params = list(conv.parameters())

        def step():
            x.grad = None
            for p in params:
                p.grad = None
            conv(x, graph).backward(gy)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        gbs = layer_bytes(n, e, f) / (ms * 1e-3) / 1e9
        out[name] = {"what": (f"FastKAN-GIN conv layer fwd+bwd, hidden {f}, {grid} grids" if name == "fastkan"
                              else f"KAN-GIN conv layer fwd+bwd, hidden {f}, grid {grid} (BASELINE config 3's layer, on one GPU)"),
                     "ms_per_step": ms, "edges_per_s": e / (ms * 1e-3), "layer_algorithmic_bytes": layer_bytes(n, e, f),
                     "layer_hbm_frac": gbs / HBM_PEAK_GBS, "steps": steps}
        del conv, x, gy
    return out


# ---------------------------------------------------------------------------------------------- main
class _Reporter:
    """N > 1 only.  Rank 0 hands every candidate for THE line (interim after each finished combination, complete at the end) to a
    child forked before HIP is initialised; the child prints the last one it received when the pipe closes -- i.e. when rank 0
    ends, however it ends (normal return, watchdog exit, SIGKILL from the launcher after another rank died, a GPU fault).
    Exactly one line reaches stdout."""

    def __init__(self):
        import signal
        r, w = os.pipe()
        self.pid = os.fork()
        if self.pid == 0:
            os.close(w)
            for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                signal.signal(sig, signal.SIG_IGN)       # the launcher's clean-up must not take the reporter with it
            last = b""
            with os.fdopen(r, "rb") as f:
                for line in f:
                    if line.endswith(b"\n"):            # (a torn last line -- rank 0 died mid-write -- is dropped)
                        last = line
            if last:
                os.write(1, last)
            os._exit(0)
        os.close(r)
        self.w = os.fdopen(w, "wb", buffering=0)

    def offer(self, obj) -> None:
        self.w.write((json.dumps(obj) + "\n").encode())

    def close(self) -> None:
        self.w.close()
        os.waitpid(self.pid, 0)


class _Watchdog:
    """every phase of the N > 1 run has a deadline; a rank that overruns it leaves with EXIT CODE 3 (a launcher must be
    able to see that a transport hung; the reporter still prints what rank 0 last offered, marked "complete": false).  All ranks
    enter a phase together (barrier), so they all leave within a second of each other."""

    def __init__(self, rank: int):
        import threading
        self.rank, self.deadline, self.name = rank, None, ""
        self.default = float(os.environ.get("KAGNN_BENCH_PHASE_TIMEOUT", "90"))
        threading.Thread(target=self._run, daemon=True).start()

    def phase(self, name: str, seconds: float = None) -> None:
        self.name, self.deadline = name, time.monotonic() + (self.default if seconds is None else seconds)

    def clear(self) -> None:
        self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            d = self.deadline
            if d is not None and time.monotonic() > d:
                sys.stderr.write(f"bench.py: rank {self.rank}: phase {self.name!r} overran its deadline -- leaving; rank 0's reporter "
                                 "prints the last complete result\n")
                sys.stderr.flush()
                os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=os.environ.get("KAGNN_WORKLOAD", "headline"))
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--hidden", type=int, default=None, help="override the workload's hidden width")
    ap.add_argument("--grid", type=int, default=None, help="override the workload's grid size")
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("KAGNN_PRECISION", "split"))
    ap.add_argument("--act", default=os.environ.get("KAGNN_ACT", "fp32"), choices=["fp32", "bf16"],
                    help="bf16: the rows the aggregation gathers are stored as bf16 (build-defined mode for config 2; NOT the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="nodes in the CPU-baseline sample")
    ap.add_argument("--cpu-full", action="store_true", help="(default behaviour now) CPU baseline at the full workload size when MemAvailable >= 48 GB")
    ap.add_argument("--cpu-sample-only", action="store_true", help="CPU baseline on the 1/10-size sample even when the full size would fit")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (copy bandwidth, full model step)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--no-fp32", action="store_true", help="skip the exact-fp32 step timing")
    args = ap.parse_args()
    os.environ["KAGNN_PRECISION"] = args.precision
    os.environ["KAGNN_ACT"] = args.act
    hidden, grid = WORKLOADS[args.workload]
    f = args.hidden or hidden
    grid = args.grid or grid

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # (forked before the first HIP call of this process: the child only reads a pipe and writes one line)
    reporter = _Reporter() if (world > 1 and rank == 0 and os.environ.get("KAGNN_BENCH_REPORTER", "1") == "1") else None
    dog = _Watchdog(rank) if world > 1 else None
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
    # The timed path is the product's default: GIKANLayer makes ONE library call each way (kagnn_gin_kan_layer_fwd / _bwd).
    # Per-kernel times come from the library's own stage timer (kagnn_stage_timer_*: HIP events on the launch stream around
    # every per-operation stage INSIDE those calls), not from composing the layer out of per-op calls as rounds 1-3 did.
    # KAGNN_BENCH_LAYER_ABI=0 restores the composed form (bit-identical results) for A/B runs.
    ops._LAYER_ABI = os.environ.get("KAGNN_BENCH_LAYER_ABI", "1") == "1"

    n, e = args.nodes, args.edges
    units = (3 * e) if args.workload == "model" else e           # edges processed per step (the model has 3 conv layers)
    fp32_mode = args.precision in ("fp32", "exact", "0")
    half_mode = args.precision in ("half", "fp16", "3")
    ei = powerlaw_graph(n, e, seed=0).to(dev)
    gen = torch.Generator().manual_seed(0)
    x_full = torch.randn(n, f, generator=gen) * 0.25
    gy_full = torch.randn(n, f, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(0)
    fastkan = args.workload == "fastkan"
    model_wl = args.workload == "model"
    net = labels = None
    if model_wl:
        net = kagnn_amd.GKAN_Nodes("gin", 3, f, f, 40, skip=True, grid_size=grid, spline_order=args.order, hidden_layers=2)
        labels = torch.randint(0, 40, (n,), generator=torch.Generator().manual_seed(2)).to(dev)
        conv = net.convs[0]
    elif fastkan:
        conv = kagnn_amd.GIFASTKANLayer(f, f, grid_size=grid, hidden_dim=f, nb_layers=2)
    else:
        conv = kagnn_amd.GIKANLayer(f, f, grid_size=grid, spline_order=args.order, hidden_dim=f, nb_layers=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    graph = ops.GraphIndex(ei, n)
    torch.cuda.synchronize()
    graph_build_first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    graph = ops.GraphIndex(ei, n)                       # (both directions: CSR by destination and its transpose)
    torch.cuda.synchronize()
    graph_build_ms = (time.perf_counter() - t0) * 1e3       # one-time per edge_index (cached on its identity): NOT in the timed region

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps):
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    def contract_fields(value_, ms_, parallelism_):
        """the contract's part of the JSON line (everything else is added by the rank-0 block at the end)"""
        return {
            "metric": "edges/sec KAN-GIN fwd+bwd, hidden=64 grid=5, 1M-node synthetic; HBM % peak",
            "value": value_, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_, "higher_is_better": True, "scaling": "strong" if world > 1 else "n/a",
            "vs_baseline": None, "dtype": ("f32" if fp32_mode else
                                          "f16 operands rounded once, fp32 accumulate (KAGNN_PRECISION=half: build-defined config-2 mode, NOT the headline)"
                                          if half_mode else "f32 (fp16 hi/lo split operands, fp32 accumulate)") +
                                          (" + bf16 gather operands (KAGNN_ACT=bf16, build-defined config-2 mode)" if args.act == "bf16" else ""),
            "data": "synthetic",
            "config": {"workload": (f"{args.workload}: GKAN_Nodes(gin, 3 x KAN-GIN conv hidden {f} grid {grid} order {args.order}, BatchNorm, skip read-out, "
                                    "40 classes) TRAINING step = forward, softmax + cross-entropy, backward, Adam (time_model.py:35-48); value = 3E / t_step, "
                                    if model_wl else
                                    f"{args.workload}: FastKAN-GIN conv layer fwd+bwd (aggregate + FastKAN([{f},{f},{f}]) num_grids={grid}), "
                                    if fastkan else
                                    f"{args.workload}: KAN-GIN conv layer fwd+bwd (aggregate + KAN([{f},{f},{f}]) grid={grid} order={args.order}), ")
                                   + f"power-law graph N={n} E={e} seed 0 (SURVEY 8(d))",
                       "nodes": n, "edges": e, "hidden": f, "grid_size": grid, "spline_order": args.order,
                       "precision": args.precision, "activation_storage": args.act, "parallelism": parallelism_},
        }

    def model_step_fn(module, xin):
        """one epoch of the reference's timing loop (time_model.py:35-48) on `module` (the node model, or its column-sharded form)"""
        try:
            opt = torch.optim.Adam(module.parameters(), lr=0.001, fused=True)
        except (TypeError, RuntimeError):
            opt = torch.optim.Adam(module.parameters(), lr=0.001)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = ops.softmax_cross_entropy(module(xin, graph), labels, None, pre_softmax=True)
            loss.backward()
            opt.step()
        return step

    alt = None
    if world == 1 and model_wl:
        net = net.to(dev).train()
        x = x_full.to(dev)
        step = model_step_fn(net, x)
        parallelism = "single GPU"
    elif world == 1:
        conv = conv.to(dev)
        x = x_full.to(dev)
        if args.act == "bf16":
            x = x.to(torch.bfloat16)                      # the mode's storage format: the activation ARRIVES as bf16
        x = x.requires_grad_(True)
        gy = gy_full.to(dev)
        params = list(conv.parameters())

        def step():
            x.grad = None
            for p in params:
                p.grad = None
            y = conv(x, graph)
            y.backward(gy)
        parallelism = "single GPU"
    else:
        from kagnn_amd.sharded import ShardedGIFASTKANLayer, ShardedGIKANLayer, ShardedNodeModel, TransposedShardedGIKANLayer
        classes = {"feature": ShardedGIKANLayer, "transposed": TransposedShardedGIKANLayer}

        def describe(scheme, comm):
            if model_wl:
                return (f"ShardedNodeModel x{world}: every activation as a column shard, 3 x feature-sharded KAN-GIN conv (RCCL reduce-scatter "
                        "fwd / all-gather bwd per KANLinear), shard-local BatchNorm1d, input-sharded skip read-out closed by ONE all-reduce of "
                        "the [N, 40] partial sums, loss on every rank, sharded parameters (local Adam)")
            if fastkan:
                return (f"feature-sharded x{world}: FastKAN coefficients split by input feature; per FastKANLayer the LayerNorm exchange (2 floats "
                        "per row: all-gather of the local moments fwd, all-reduce of the two row sums bwd) + RCCL reduce-scatter (fwd) / "
                        "all-gather (bwd) of the partial sums, row-chunked and overlapped")
            if scheme == "feature":
                return (f"feature-sharded x{world}: spline coefficients split by input feature, " +
                        ("RCCL reduce-scatter (fwd) / all-gather (bwd) per KANLinear, row-chunked and overlapped (north_star's scheme)"
                         if comm == "rccl" else
                         "the same RCCL exchange as ONE library call per KANLinear each way on the layer's own ncclComm_t "
                         "(kagnn_sharded_kan_linear_fwd / _bwd, include/kagnn_rccl.h; north_star's scheme)"
                         if comm == "rccl_c" else
                         "direct peer-to-peer reduce-scatter (fwd) / all-gather (bwd) kernels over hipIpc-mapped peer buffers per "
                         "KANLinear, row-chunked and overlapped (north_star's partitioning, SURVEY 8(e)'s hand-rolled exchange)"))
            return (f"column-sharded aggregation + row-sharded KAN chain x{world}: " +
                    ("RCCL all-to-all both ways" if comm == "rccl" else "direct peer-to-peer pulls over hipIpc-mapped buffers both ways") +
                    ", one flat weight-gradient all-reduce")

        # test aid for the safety net above (tests/test_sharded_gloo.py): KAGNN_BENCH_FAULT="<scheme>/<comm>:hang|kill[:rank]"
        # makes that combination hang on / kill the given rank (default 1) -- what a broken transport would do
        fault = os.environ.get("KAGNN_BENCH_FAULT", "")

        def make(scheme, comm):
            if fault.startswith(f"{scheme}/{comm}:"):
                how, _, who = fault.split(":", 1)[1].partition(":")
                if rank == int(who or 1):
                    if how == "kill":
                        os.kill(os.getpid(), 9)
                    time.sleep(1e6)
            if model_wl:
                torch.manual_seed(0)
                sm = ShardedNodeModel(net, dist.group.WORLD, comm=comm).to(dev).train()
                return model_step_fn(sm, sm.shard_columns(x_full.to(dev)))
            cls = classes[scheme]
            sconv = (ShardedGIFASTKANLayer(conv, dist.group.WORLD) if fastkan else
                     cls(conv, dist.group.WORLD, sync_in_backward=False, comm=comm) if cls is TransposedShardedGIKANLayer
                     else cls(conv, dist.group.WORLD, comm=comm)).to(dev)
            xs = sconv.shard_columns(x_full.to(dev)).requires_grad_(True)
            gs = sconv.shard_columns(gy_full.to(dev))
            ps = list(sconv.parameters())

            def step():
                xs.grad = None
                for p in ps:
                    p.grad = None
                y = sconv(xs, graph)
                y.backward(gs)
                if hasattr(sconv, "sync_gradients"):
                    sconv.sync_gradients()             # one flat all-reduce for all weight gradients
            return step

        # Which combination is `value`?  KAGNN_SHARDING / KAGNN_COMM pin it; otherwise every combination runs under the full
        # contract (W warm-up + K timed steps, max over ranks) and the fastest is reported (re-timed below with the stage timer
        # on).  Order = least exotic first: plain RCCL through torch.distributed, then the direct peer-to-peer kernels, then the
        # library's own RCCL entry points -- after every combination that finishes, rank 0 hands an interim line to the reporter,
        # so a later transport that hangs (watchdog) or kills a rank still leaves a measured line.  A combination that raises is
        # listed with its error and skipped (setup errors are symmetric across ranks; an asymmetric one ends in the watchdog).
        pin_s, pin_c = os.environ.get("KAGNN_SHARDING"), os.environ.get("KAGNN_COMM")
        order = [("feature", "rccl"), ("transposed", "rccl"), ("feature", "p2p"), ("transposed", "p2p"), ("feature", "rccl_c")]
        if fastkan or model_wl:      # (the FastKAN layer and the whole node model exist in north_star's scheme over torch.distributed / RCCL)
            order = [("feature", "rccl")]
        combos = [(sc, cm) for sc, cm in order if (pin_s is None or sc == pin_s) and (pin_c is None or cm == pin_c)]
        if not combos:
            raise SystemExit("KAGNN_SHARDING must be 'feature' or 'transposed', KAGNN_COMM 'rccl', 'rccl_c' (feature only) or 'p2p'")
        probe = []

        NORTH_STAR = (("feature", "rccl"), ("feature", "rccl_c"))      # spline coefficients sharded by input feature + RCCL

        def selection(entries):
            """the combination `value` is quoted on: the fastest of north_star's scheme; only if none of it ran, the fastest of the rest"""
            ok_ = [p_ for p_ in entries if "error" not in p_]
            ns_ = [p_ for p_ in ok_ if (p_["scheme"], p_["comm"]) in NORTH_STAR]
            pool = ns_ or ok_
            return min(pool, key=lambda p_: p_["ms_per_step"]) if pool else None

        def alternative(entries):
            rest = [p_ for p_ in entries if "error" not in p_ and (p_["scheme"], p_["comm"]) not in NORTH_STAR]
            return min(rest, key=lambda p_: p_["ms_per_step"]) if rest else None

        for ci, (sc, cm) in enumerate(combos):
            entry = {"scheme": sc, "comm": cm, "parallelism": describe(sc, cm)}
            dog.phase(f"combination {sc}/{cm}", 300.0 if ci == 0 else None)      # (the first one includes RCCL's start-up)
            try:
                st = make(sc, cm)
                for _ in range(args.warmup):
                    st()
                dtp = timed(st, args.steps)
                entry.update(ms_per_step=dtp / args.steps * 1e3, value=units / (dtp / args.steps), steps=args.steps, warmup=args.warmup)
                del st
            except Exception as ex:                       # noqa: BLE001 -- reported in the line, not swallowed
                entry["error"] = f"{type(ex).__name__}: {ex}"[:300]
            dog.clear()
            torch.cuda.empty_cache()
            probe.append(entry)
            best = selection(probe)
            if reporter is not None and best is not None:
                line = contract_fields(best["value"], best["ms_per_step"], best["parallelism"])
                line["multi_gpu_probe"] = {"selected": {"scheme": best["scheme"], "comm": best["comm"]}, "combinations": list(probe),
                                           "not_finished": [f"{a}/{b}" for a, b in combos[ci + 1:]]}
                line["value_scheme_is_north_star"] = (best["scheme"], best["comm"]) in NORTH_STAR
                line["fastest_alternative"] = alternative(probe)
                line["complete"] = False
                line["interim"] = ("this line was written after the last combination that finished; a later phase of the run hung or "
                                   "lost a rank (stderr names it)")
                reporter.offer(line)
        best = selection(probe)
        if best is None:
            raise SystemExit("bench.py: every multi-GPU combination failed: " + json.dumps(probe))
        dog.phase("final timing of the selected combination", 180.0)
        step = make(best["scheme"], best["comm"])
        parallelism = best["parallelism"]
        alt = {"selected": {"scheme": best["scheme"], "comm": best["comm"]},
               "value_scheme_is_north_star": (best["scheme"], best["comm"]) in NORTH_STAR,
               "fastest_alternative": alternative(probe),
               "how": ("pinned by KAGNN_SHARDING / KAGNN_COMM" if len(combos) == 1 else
                       f"fastest of north_star's scheme (feature/rccl, feature/rccl_c); every combination was run under the contract ({args.warmup} warm-up + {args.steps} timed steps, max over "
                       "ranks); `value` is its re-run with the stage timer on the dominant kernel"),
               "combinations": probe}

    for _ in range(args.warmup):
        step()
    # per-stage breakdown: three extra untimed steps with the library's stage timer recording every stage (HIP events on
    # the launch stream around each per-operation stage inside the layer calls).  In the timed region only the dominant
    # stage keeps its events (the roofline figure is measured there, live): 2 event records per launch of that one kernel.
    PROFILE_STEPS = 3
    sync()
    with ops.LibraryStageTimer(None):
        for _ in range(PROFILE_STEPS):
            step()
        sync()
    warm = ops.LibraryStageTimer.collect()
    # the dominant COMPUTE stage (the exchange kernels of the sharded layers -- kagnn_p2p_* -- are wire-bound and have no HBM / MFMA
    # roofline: they stay in entry_points_ms_per_step, the roofline block describes the largest stage that has one)
    ROOFLINE_STAGES = ("kagnn_aggregate_sum", "kagnn_aggregate_sum_bf16", "kagnn_kan_linear_fwd", "kagnn_kan_linear_fwd_moments",
                       "kagnn_kan_linear_bwd_input", "kagnn_kan_linear_bwd_weight", "kagnn_fastkan_fwd", "kagnn_fastkan_bwd",
                       "kagnn_fastkan_shard_fwd", "kagnn_fastkan_shard_bwd")
    if model_wl:     # (the model's KAN launches have several shapes -- conv layers and the 256 -> 40 read-out --: the roofline block is its aggregation)
        ROOFLINE_STAGES = ("kagnn_aggregate_sum",)
    cand = {k: v for k, v in warm.items() if k in ROOFLINE_STAGES}
    only = max(cand, key=lambda k: cand[k]["total_ms"]) if cand else None
    torch.cuda.synchronize()
    resident_bytes = torch.cuda.memory_allocated()          # graph (CSR + transpose), x, gy / labels, parameters, optimiser state
    torch.cuda.reset_peak_memory_stats()
    with ops.LibraryStageTimer(only):
        dt = timed(step, args.steps)
    peak_bytes = torch.cuda.max_memory_allocated()          # (the memory contract, SURVEY 7.3: the reference needs 45.1 GB RSS at 1M / 64)
    prof_live = ops.LibraryStageTimer.collect()
    if dog is not None:
        dog.phase("per-rank gather", 60.0)
    ms = dt / args.steps * 1e3
    value = units / (dt / args.steps)

    # N > 1: what each rank's device spent INSIDE the library (kernels, summed over the entry points of the profile steps) vs the
    # step time -- the rest is exposed waiting on the collectives (+ launch gaps): explains the driver's scaling curve
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "library_ms_per_step": sum(v["total_ms"] for v in warm.values()) / PROFILE_STEPS}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = [dict(g, not_in_library_ms_per_step=max(0.0, ms - g["library_ms_per_step"])) for g in gathered]
    if dog is not None:
        dog.clear()

    fp32_ms = None
    if not args.no_fp32 and not fp32_mode and world == 1 and not model_wl:
        for l in conv.nn.layers:
            l.precision = ops.PREC_FP32
        for _ in range(2):
            step()
        k32 = max(3, min(args.steps, 5))
        fp32_ms = timed(step, k32) / k32 * 1e3
        for l in conv.nn.layers:
            l.precision = None
    # the build-defined reduced-precision mode of BASELINE config 2 (KAGNN_PRECISION=half: one fp16 product per fp32 product) on
    # the same layer, measured after the timed region like the exact-fp32 figure -- reported beside the headline, never as it
    half_ms = None
    if not args.no_fp32 and not fp32_mode and not half_mode and world == 1 and not fastkan and not model_wl:
        for l in conv.nn.layers:
            l.precision = ops.PREC_HALF
        for _ in range(2):
            step()
        kh = max(3, min(args.steps, 10))
        half_ms = timed(step, kh) / kh * 1e3
        for l in conv.nn.layers:
            l.precision = None

    if rank == 0:
        prof = prof_live
        c = grid if fastkan else grid + args.order          # coefficients (RBF centres) per input feature
        fl = f // world if world > 1 else f
        per_step = {k: v["total_ms"] / PROFILE_STEPS for k, v in warm.items()}
        dom = only if only in prof else max((k for k in per_step if k in ROOFLINE_STAGES), key=per_step.get)
        mfma_peak = FP32_MFMA_PEAK_TF if fp32_mode else F16_MFMA_PEAK_TF
        products = 1.0 if (fp32_mode or half_mode) else 3.0            # split mode: hi*hi + hi*lo + lo*hi on the fp16 matrix cores (half: one)
        nrows = n if world == 1 else n                  # (feature sharding keeps all rows on every rank)
        kan = kan_flops(nrows, fl, f, c)
        spec = {   # entry point -> (what, algorithmic bytes per launch, algorithmic flops per launch)
            "kagnn_aggregate_sum": ("neighbour aggregation (fwd; bwd = same kernel on the transposed CSR)", agg_bytes(n, e, fl), 2.0 * e * fl),
            "kagnn_aggregate_sum_bf16": ("neighbour aggregation, bf16 gather operands (fwd: bf16 in / fp32 out; bwd: bf16 in / bf16 out)",
                                         e * (2 * fl + 4) + n * (2 * fl + 4 * fl + 4), 2.0 * e * fl),
            "kagnn_kan_linear_fwd": ("KANLinear forward", 4.0 * nrows * (fl + f), kan),
            "kagnn_kan_linear_bwd_input": ("KANLinear input gradient (reads x, gy; writes gx)", 4.0 * nrows * (2 * fl + f), kan),
            "kagnn_kan_linear_bwd_weight": ("KANLinear weight gradient (reads x, gy)", 4.0 * nrows * (fl + f), kan),
            "kagnn_fastkan_fwd": ("FastKANLayer forward (LayerNorm statistics, RBF expansion, both linear maps, bias)",
                                  4.0 * nrows * (fl + f) + 8.0 * nrows, kan),
            "kagnn_fastkan_bwd": ("FastKANLayer backward (input gradient through the LayerNorm, LayerNorm / spline / base weight and "
                                  "bias gradients)", 4.0 * nrows * (3 * fl + 2 * f), 2.0 * kan),
            "kagnn_fastkan_shard_fwd": ("feature-sharded FastKANLayer forward on this rank's input columns (merged LayerNorm statistics given): "
                                        "partial sums for all outputs", 4.0 * nrows * (fl + f) + 8.0 * nrows, kan),
            "kagnn_fastkan_shard_bwd": ("feature-sharded FastKANLayer backward without the LayerNorm finish (two launches per layer: input-gradient "
                                        "half, weight-gradient half)", 4.0 * nrows * (1.5 * fl + f) + 8.0 * nrows, kan),
        }
        spec["kagnn_kan_linear_fwd_moments"] = spec["kagnn_kan_linear_fwd"]
        if model_wl:
            spec = {"kagnn_aggregate_sum": spec["kagnn_aggregate_sum"]}
        kernels = []
        for name, (what, nbytes, flops) in spec.items():
            if name not in warm:
                continue
            avg = (prof[name]["avg_ms"] if name in prof else warm[name]["avg_ms"]) * 1e-3
            gbs = nbytes / avg / 1e9
            tf = products * flops / avg / 1e12
            kernels.append({"entry_point": name, "what": what, "avg_launch_ms": avg * 1e3,
                            "launches_per_step": warm[name]["launches"] / PROFILE_STEPS,
                            "ms_per_step": per_step[name],
                            "measured_in": "timed region (library stage timer)" if name in prof else "3 untimed profile steps (library stage timer)",
                            "algorithmic_bytes_per_launch": nbytes, "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                            "algorithmic_flops_per_launch": flops,
                            "mfma_TFs_incl_split_products": tf, "mfma_frac": tf / mfma_peak if not name.startswith("kagnn_aggregate_sum") else 0.0,
                            "bound": "hbm" if name.startswith("kagnn_aggregate_sum") else "mfma/valu issue"})
        by_name = {k["entry_point"]: k for k in kernels}
        d = by_name.get(dom)
        if d is not None and dom.startswith("kagnn_aggregate_sum"):
            roof = {"kernel": dom, "bound": "hbm", "achieved": d["hbm_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                    # the gathered matrix (N x F x 4 = 256 MB at the headline shape) is the size of the Infinity Cache: most
                    # of the E gathered rows are served on-die, so `achieved` is FABRIC-side bandwidth and may exceed what
                    # HBM alone delivers (6.3 TB/s achievable); the bytes that must cross the HBM pins are listed beside it
                    "served_from": "fabric (HBM + 256 MB Infinity Cache)",
                    "compulsory_hbm_bytes_per_launch": (2 if args.act == "bf16" else 4) * n * fl + 4 * n * fl + 4 * e + 4 * n}
        elif d is not None:
            roof = {"kernel": dom, "bound": "mfma", "achieved": d["mfma_TFs_incl_split_products"], "peak": mfma_peak,
                    "unit": "TFLOP/s", "algorithmic_flops_per_launch": d["algorithmic_flops_per_launch"]}
        else:       # (an entry point without an algorithmic-bytes / flops figure must not end up as a silent 0 in the line)
            raise RuntimeError(f"bench.py: the dominant entry point {dom!r} has no roofline specification (spec table above)")
        roof["frac"] = roof["achieved"] / roof["peak"]
        # the honest whole-layer figure next to the dominant kernel's: all of B_layer over the step time
        lbytes = layer_bytes(n, e, f) * (3 if model_wl else 1)          # (model: the three conv layers' B_layer; read-out / norms / loss not counted)
        roof["layer_frac"] = lbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["avg_launch_ms"] = d["avg_launch_ms"] if d else None
        roof["traffic"] = None
        roof["note"] = ("dominant = largest device time per step; it is the kernel closest to its roofline and its bytes are "
                        "mostly served by the Infinity Cache (served_from) -- the limiting kernels are in roofline_kernels, and "
                        "the honest headline figure is layer_hbm_frac")
        layer_gbs = lbytes / (ms * 1e-3) / 1e9
        out = contract_fields(value, ms, parallelism)
        out.update({
            "layer_algorithmic_bytes": lbytes,
            "layer_hbm_GBs": layer_gbs, "layer_hbm_frac": layer_gbs / HBM_PEAK_GBS,
            "fp32_mode_ms_per_step": fp32_ms,
            "half_mode_ms_per_step": half_ms,      # KAGNN_PRECISION=half on the same layer (build-defined config-2 mode; ~3e-4 from fp32)
            "fp32_mode_layer_hbm_frac": (layer_bytes(n, e, f) / (fp32_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if fp32_ms else None,
            "roofline": roof,
            "roofline_kernels": kernels,
            "entry_points_ms_per_step": per_step,
            "entry_points_measured_in": "3 extra untimed steps after the warm-up; kagnn_stage_timer_* (HIP events on the launch stream "
                                        "around every stage inside the library calls)",
            "timed_path": ("product default: one library call per convolution each way (kagnn_gin_kan_layer_fwd / _bwd)" if ops._LAYER_ABI
                           else "composed from the per-operation entry points (KAGNN_BENCH_LAYER_ABI=0)") if world == 1 and not fastkan and not model_wl
                          else "the node model's default path (fused conv + norm nodes, lazy norms, one-launch skip read-out)" if world == 1 and model_wl
                          else "per-operation entry points (sharded / FastKAN layers)",
            # one-time per edge_index (cached on its identity); outside the timed region
            # device memory over the timed region (torch.cuda.max_memory_allocated; this rank): everything resident before the step
            # (graph index, inputs, parameters) and the step's own peak on top -- saved activations are the layer INPUTS only
            "peak_device_GB": peak_bytes / 1e9,
            "device_memory": {"resident_before_step_GB": resident_bytes / 1e9, "step_peak_over_resident_GB": (peak_bytes - resident_bytes) / 1e9,
                              "activation_matrix_GB": n * f * 4 / 1e9,
                              "reference_cpu_rss_GB": 45.1 if (args.workload == "headline" and n == 1_000_000) else None,
                              "how": "torch.cuda.max_memory_allocated() around the timed region; reference figure: BASELINE.md section 2"},
            "graph_index_build_ms": {"steady": graph_build_ms, "first_call": graph_build_first_ms,
                                     "what": "CSR by destination + its transpose (stable radix sort, hub segments), int64 edge_index already in HBM"},
        })
        out["complete"] = True
        if alt is not None:
            out["value_scheme_is_north_star"] = alt.pop("value_scheme_is_north_star")
            out["fastest_alternative"] = alt.pop("fastest_alternative")
            out["multi_gpu_probe"] = alt
            ns = [p_ for p_ in alt["combinations"] if p_["scheme"] == "feature" and p_["comm"] in ("rccl", "rccl_c")]
            if ns:                                        # (spline coefficients sharded by input feature + RCCL: the faster of its two hosts)
                out["north_star_scheme"] = min(ns, key=lambda p_: p_.get("ms_per_step", float("inf")))
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["per_rank_note"] = ("library_ms_per_step: HIP-event time of this rank's kernels (3 profile steps); not_in_library: the rest of "
                                    "the step = exposed waiting on RCCL collectives + launch gaps")
        if world == 1:
            copy_gbs = copy_bandwidth(dev)
            out["hbm_copy_GBs"] = copy_gbs
            out["layer_frac_of_copy_bw"] = layer_gbs / copy_gbs
            roof["frac_of_copy_bw"] = (roof["achieved"] / copy_gbs) if roof["unit"] == "GB/s" else None
            for k in kernels:
                k["hbm_frac_of_copy_bw"] = k["hbm_GBs"] / copy_gbs
        if not args.no_extras and world == 1 and not fastkan and not model_wl:
            out["secondary"] = secondary_figures(dev, conv, graph, x.detach().float(), n, e, f, grid, args.order)
            conv_ms = 3 * ms
            out["secondary"]["model_step"]["conv_layers_share"] = conv_ms / out["secondary"]["model_step"]["ms_per_step"]
            if args.workload == "headline" and args.act == "fp32" and not fp32_mode:
                out["secondary"]["other_layers"] = other_layer_figures(dev, graph, n, e)
                out["secondary"]["graph_level_step"] = graph_level_step_figures(dev)
        if not args.no_traffic and world == 1 and not model_wl:
            torch.cuda.synchronize()
            prefix = {"kagnn_aggregate_sum": "agg_rows", "kagnn_kan_linear_fwd": "kan_sparse_fwd",
                      "kagnn_kan_linear_bwd_input": "kan_split_dx", "kagnn_kan_linear_bwd_weight": "kan_split_dw",
                      "kagnn_fastkan_fwd": "kan_split_fwd", "kagnn_fastkan_bwd": "kan_split_dw"}.get(dom, "agg_rows")
            traffic, detail = measure_traffic(args, prefix)
            roof["traffic"] = traffic
            roof["traffic_how"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes spawned by this run over 2 steps of the same "
                                   "command; (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 correction for 16 B/lane reads); "
                                   "fabric-side bytes, Infinity-Cache hits included") if traffic else str(detail)
            if isinstance(detail, dict):
                out["traffic_per_kernel_bytes"] = detail
        if not args.no_cpu_baseline and world == 1 and not model_wl:
            # the FULL workload when the host has the memory for the reference algorithm's ~46 GB of dense
            # bases (1 warm-up + 1 timed pass at 32 threads, ~75 s on the GPU boxes' 2 x 64-core hosts), else the 1/10 sample
            mem = _mem_available_gb()
            if mem >= 48 and not args.cpu_sample_only:
                out["cpu_baseline"] = cpu_baseline(n, e, f, grid, args.order, full=True, arch="fastkan" if fastkan else "kan")
                out["cpu_baseline"]["selected"] = f"full size (MemAvailable {mem:.0f} GB >= 48 GB)"
            else:
                ns = min(args.cpu_sample, n)
                out["cpu_baseline"] = cpu_baseline(ns, ns * (e // n if n else 10), f, grid, args.order,
                                                   arch="fastkan" if fastkan else "kan")
                out["cpu_baseline"]["selected"] = ("1/10-size sample (--cpu-sample-only)" if args.cpu_sample_only
                                                   else f"1/10-size sample (MemAvailable {mem:.0f} GB < 48 GB)")
        if reporter is not None:
            reporter.offer(out)                          # (the reporter prints it when this process ends)
        else:
            print(json.dumps(out), flush=True)
    if dog is not None:
        dog.phase("shutdown", 60.0)
    if dist is not None:
        dist.destroy_process_group()
    if reporter is not None:
        reporter.close()


if __name__ == "__main__":
    main()
