#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time stats and per-kernel PMC averages."""
import re, sqlite3, sys, glob, os, json

def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("kagnn::", "")
    return n[:70]

def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, (end-start) from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0, 1e18, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = []
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((n, a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / tot))
    return out

def pmc_stats(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    q = "select kernel_name, counter_name, value from counters_collection" if "kernel_name" in cols else None
    if q is None:
        print(cols); return {}
    agg = {}
    for kn, cn, v in cur.execute(q):
        a = agg.setdefault((short(kn), cn), [0, 0.0])
        a[0] += 1; a[1] += v
    return {k: v[1] / v[0] for k, v in agg.items()}

if __name__ == "__main__":
    d = sys.argv[1]
    t = glob.glob(os.path.join(d, "trace", "*.db"))
    if t:
        print(f"{'kernel':70s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
        for r in kernel_stats(t[0]):
            print(f"{r[0]:70s} {r[1]:6d} {r[2]:10.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {r[6]:6.2f}")
    allp = {}
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        for db in glob.glob(os.path.join(sub, "*.db")):
            allp.update(pmc_stats(db))
    kernels = sorted({k[0] for k in allp})
    ctrs = sorted({k[1] for k in allp})
    for kn in kernels:
        vals = {c: allp[(kn, c)] for c in ctrs if (kn, c) in allp}
        print("\n" + kn)
        print("   " + "  ".join(f"{c}={v:.4g}" for c, v in vals.items()))
