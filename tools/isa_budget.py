#!/usr/bin/env python3
"""Instruction budget of the hot loops of the three KAN kernels, read from the COMPILED ISA (hipcc --offload-device-only -S),
not from source-level counting: per kernel, the instructions between the loop header the matrix instructions sit in and its
back edge, by class, and normalised per scalar x[n][f] (one wave-iteration covers `scalars_per_lane` scalars per lane).

    python tools/isa_budget.py [--write profiles/r03_isa_budget.txt]

The loop is found as the innermost `; =>This Inner Loop Header` region that contains the kernel's matrix instructions; rarely
executed side paths inside it (the exact-fp32 SiLU branch: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 and the instructions of its
basic blocks) are reported separately."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kagnn_amd", "csrc")
KERNELS = [
    # (source, mangled-name prefix, label, scalars per lane and loop iteration, what one iteration covers)
    ("kan_sparse_fwd.hip", "_ZN5kagnn21kan_sparse_fwd_kernelILi2ELb0ELb0ELb0ELin1ELb0E", "forward  kan_sparse_fwd_kernel<2,false,false,false>", 8 * 4,
     "one row tile: 4 groups x 8 scalars per lane (32 rows x 64 features per wave), 64 outputs"),
    ("kan_split_bwd.hip", "_ZN5kagnn19kan_split_dx_kernelILi3ELi2ELb0ELi1ELb0E", "dX       kan_split_dx_kernel<3,2,false,1,false>", 8,
     "one (row tile, 16-feature tile): 8 scalars per lane (32 rows x 16 features per wave), 64 outputs"),
    ("kan_split_bwd.hip", "_ZN5kagnn19kan_split_dw_kernelILi3ELb0ELi1ELi4E", "dW       kan_split_dw_kernel<3,false,1,4>", 8,
     "one 32-row chunk: 8 scalars per lane (32 rows x 16 features per wave), 64 outputs"),
]
CLASSES = [
    ("MFMA (fp16 / sparse)", r"^v_(s?mfmac?|mfma)_f32_(16x16x32|32x32x16|32x32x32)"),
    ("MFMA (exact fp32, rare path)", r"^v_mfma_f32_(16x16x4|32x32x2)"),
    ("v_perm_b32", r"^v_perm_b32"),
    ("fp16 conversion (v_cvt_pkrtz)", r"^v_cvt_pkrtz"),
    ("hi/lo residual (v_fma_mix)", r"^v_fma_mix"),
    ("packed fp32 (v_pk_*)", r"^v_pk_"),
    ("transcendental (exp, rcp)", r"^v_(exp|rcp|log|sqrt|rsq)_"),
    ("v_mov / v_accvgpr_*", r"^v_(mov_b|accvgpr)"),
    ("other VALU", r"^v_"),
    ("LDS (ds_*)", r"^ds_"),
    ("global / buffer memory", r"^(buffer|global|flat|scratch)_"),
    ("s_waitcnt / s_nop / s_barrier", r"^s_(waitcnt|nop|barrier)"),
    ("other scalar", r"^s_"),
]


def loops(lines):
    """[(start, end)] of all loops of one kernel: the basic blocks LLVM annotates with `Loop Header` / `in Loop: Header=BBx_y` /
    `Parent Loop BBx_y`, from the first such block to the end of the last one"""
    labels = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)]
    out = []
    for i, l in enumerate(lines):
        if "Loop Header" not in l:
            continue
        j = i
        while j >= 0 and not re.match(r"^\.LBB\d+_\d+:", lines[j]):
            j -= 1
        hid = lines[j].split(":")[0].lstrip(".L")
        member = [k for k in labels if k == j or re.search(r"(Header=|Parent Loop )" + hid + r"\b", lines[k] + " " + lines[min(k + 1, len(lines) - 1)])]
        first, last = min(member), max(member)
        nxt = [k for k in labels if k > last]
        out.append((first, (nxt[0] if nxt else len(lines)) - 1))
    return out


def main(argv):
    rep = []
    with tempfile.TemporaryDirectory() as tmp:
        asm = {}
        for src in sorted({k[0] for k in KERNELS}):
            o = os.path.join(tmp, src + ".s")
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DNDEBUG", "--offload-device-only", "-S",
                            "-I", os.path.join(ROOT, "include"), "-o", o, os.path.join(CSRC, src)], check=True, capture_output=True)
            asm[src] = open(o).read().splitlines()
        for src, prefix, label, spl, what in KERNELS:
            L = asm[src]
            s = next(i for i, l in enumerate(L) if l.startswith(prefix) and l.rstrip().endswith(":") is False and ":" in l)
            e = next(i for i in range(s, len(L)) if ".end_amdhsa_kernel" in L[i] or (i > s and re.match(r"^_ZN5kagnn", L[i])))
            body = L[s:e]
            hot_mfma = r"\s*v_s?mfmac?_f32_(16x16x32|32x32x16|32x32x32)"
            total = sum(1 for x in body if re.match(hot_mfma, x))
            cand = [(a, b) for a, b in loops(body) if sum(1 for x in body[a:b] if re.match(hot_mfma, x)) >= 0.9 * total]
            a, b = min(cand, key=lambda ab: ab[1] - ab[0])          # the smallest loop that holds the kernel's matrix instructions
            # basic blocks holding the exact-fp32 MFMAs = the rare path
            blocks, cur = [], []
            for x in body[a:b + 1]:
                if re.match(r"^\.LBB", x) and cur:
                    blocks.append(cur); cur = []
                cur.append(x)
                if re.match(r"\s*s_c?branch", x):          # a basic block also ends at a branch
                    blocks.append(cur); cur = []
            blocks.append(cur)
            hot, rare = {}, {}
            for blk in blocks:
                is_rare = any(re.match(r"\s*v_mfma_f32_(16x16x4|32x32x2)", x) for x in blk)
                for x in blk:
                    ins = x.strip().split()[0] if x.strip() and not x.strip().startswith((";", ".")) else None
                    if not ins or ins.endswith(":"):
                        continue
                    for name, pat in CLASSES:
                        if re.match(pat, ins):
                            (rare if is_rare else hot)[name] = (rare if is_rare else hot).get(name, 0) + 1
                            break
            tot = sum(hot.values())
            rep.append(f"{label}\n  loop = {what}\n  {'class':34s} {'per iteration':>14s} {'per scalar':>11s}")
            for name, _ in CLASSES:
                if hot.get(name):
                    rep.append(f"  {name:34s} {hot[name]:14d} {hot[name] / spl:11.1f}")
            rep.append(f"  {'all instructions (hot path)':34s} {tot:14d} {tot / spl:11.1f}")
            if rare:
                rep.append(f"  (+ {sum(rare.values())} instructions in the blocks of the exact-fp32 SiLU branch, executed only for values beyond fp16 range)")
            rep.append("")
    txt = "\n".join(rep)
    print(txt)
    if "--write" in argv:
        open(argv[argv.index("--write") + 1], "w").write(
            "Instruction budget of the hot loops, from the compiled ISA (tools/isa_budget.py; gfx950, hipcc -O3 of HEAD)\n"
            "A lone wave issues ~one instruction per 4-5 cycles whatever its class (profiles/r03_experiments.md), so `all instructions`\n"
            "x ~4.7 cycles is the time of one wave-iteration up to stalls.\n\n" + txt)


if __name__ == "__main__":
    main(sys.argv[1:])
