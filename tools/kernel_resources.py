#!/usr/bin/env python3
"""Register / scratch / LDS usage of every device kernel in libkagnn_hip.so, from the compiler's own
code-object metadata (the AMDGPU msgpack note: .vgpr_count, .agpr_count, .vgpr_spill_count,
.private_segment_fixed_size, .group_segment_fixed_size, .sgpr_count).

    python tools/kernel_resources.py                 # table for every kernel of the built objects
    python tools/kernel_resources.py --hot           # only the gated hot-path instantiations
    python tools/kernel_resources.py --write profiles/r03_kernel_resources.txt

`hot_path_report()` is what tests/test_host_cpu.py gates: a hot-path instantiation that spills
registers (scratch traffic inside an MFMA loop was the round-2 pathology) fails the CPU suite.
No GPU needed: it reads the objects that `kagnn_amd._build.build()` leaves in kagnn_amd/lib/obj.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDIR = os.path.join(ROOT, "kagnn_amd", "lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

# Kernels on the measured paths (demangled-name regex, max spilled VGPRs, max scratch bytes/lane).
# Everything that runs inside bench.py's timed region at the headline, config-3 and FastKAN workloads
# and in the model step.
HOT = [
    (r"kan_sparse_fwd_kernel<", 0, 0),
    (r"kan_split_dx_kernel<3,[12],|kan_split_dx_kernel<3,4,(true|false),0,", 0, 0),   # 128 outputs: only the plain schedule is launched
    (r"kan_split_dw_kernel<3,", 0, 0),
    (r"kan_split_dx_w2_kernel<", 0, 0),
    (r"kan_split_dw_w2_kernel<", 0, 0),
    (r"kan_split_dx_kernel<0,", 0, 0),
    (r"kan_split_dw_kernel<0,", 0, 0),
    (r"kan_split_dw_shared_kernel<", 0, 0),            # wide layers (round 4)
    (r"kan_split_fwd_kernel<", 0, 0),
    (r"kan_dx_f32_kernel<|kan_dw_f32_kernel<|kan_fwd_f32_kernel<", 0, 0),   # exact-fp32 mode: the documented fallback (round 4: was up to 53 spilled)
    (r"agg_rows_v4_kernel<", 0, 0),
    (r"agg_hub", 0, 0),
    (r"agg16_", 0, 0),
    (r"bn_", 0, 0),
]


def _run(*cmd: str) -> str:
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def _device_code_object(obj: str, tmp: str) -> str:
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    # (an explicit output operand: without one llvm-objcopy re-emits `obj` IN PLACE -- a test then bumped the mtime of every
    # build object and the next build() relinked both libraries)
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "discard.o")],
                       capture_output=True)
    if r.returncode != 0:            # a translation unit without device code (the C-ABI dispatcher)
        return ""
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    f"--targets={TARGET}", f"--output={co}"], check=True, capture_output=True)
    return co


_KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
         "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def _parse_notes(text: str):
    """The note prints as YAML: kernel records are the list items of amdhsa.kernels (2-space indent),
    their scalar fields sit at 4 spaces; argument records (deeper indent) are skipped."""
    kernels, cur = [], None
    for line in text.splitlines():
        m = re.match(r"^(  - | {4})\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        if m.group(1) == "  - ":
            cur = {}
            kernels.append(cur)
        if cur is None:
            continue
        key, val = m.group(2), m.group(3).strip()
        if key in _KEYS:
            cur[key] = int(val)
        elif key == "name":
            cur["name"] = val.strip("'\"")
    return [k for k in kernels if "name" in k and "vgpr_count" in k]


def collect(objdir: str = OBJDIR):
    if not os.path.isdir(objdir):
        raise RuntimeError(f"{objdir} missing: run `python -m kagnn_amd._build` first")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(os.listdir(objdir)):
            if not o.endswith(".o"):
                continue
            co = _device_code_object(os.path.join(objdir, o), tmp)
            if not co:
                continue
            ks = _parse_notes(_run(f"{LLVM}/llvm-readelf", "--notes", co))
            names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), check=True, capture_output=True, text=True).stdout.splitlines() if ks else []
            for k, dn in zip(ks, names):
                k["demangled"] = re.sub(r"\(.*$", "", dn).replace("void kagnn::", "").replace(", ", ",")
                k["tu"] = o[:-2]
                rows.append(k)
    return rows


def waves_per_simd(k) -> int:
    alloc = -(-k["vgpr_count"] // 8) * 8            # .vgpr_count is the unified total (arch + acc)
    alloc = max(alloc, 8)
    return min(8, 512 // alloc)


def hot_path_report(rows=None):
    """[(kernel, spilled VGPRs, scratch bytes, allowed spills, allowed scratch)] for the gated kernels."""
    rows = rows if rows is not None else collect()
    out = []
    for k in rows:
        for pat, max_spill, max_scratch in HOT:
            if re.search(pat, k["demangled"]):
                out.append((k["demangled"], k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0),
                            max_spill, max_scratch))
                break
    return out


def table(rows, only_hot=False) -> str:
    lines = [f"{'kernel':<72} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'spill':>5} {'scratch':>7} {'lds':>6} {'w/simd':>6}"]
    hot = {r[0] for r in hot_path_report(rows)}
    for k in sorted(rows, key=lambda k: (k["tu"], k["demangled"])):
        if only_hot and k["demangled"] not in hot:
            continue
        mark = "*" if k["demangled"] in hot else " "
        lines.append(f"{mark}{k['demangled'][:71]:<71} {k['vgpr_count']:>4} {k.get('agpr_count', 0):>4} {k['sgpr_count']:>4} "
                     f"{k.get('vgpr_spill_count', 0):>5} {k.get('private_segment_fixed_size', 0):>7} "
                     f"{k.get('group_segment_fixed_size', 0):>6} {waves_per_simd(k):>6}")
    return "\n".join(lines)


def main(argv):
    rows = collect()
    txt = table(rows, only_hot="--hot" in argv)
    bad = [r for r in hot_path_report(rows) if r[1] > r[3] or r[2] > r[4]]
    txt += f"\n\n{len(rows)} kernels; hot-path instantiations (*) that spill: {len(bad)}\n"
    for r in bad:
        txt += f"  SPILL {r[0]}: {r[1]} VGPRs, {r[2]} B scratch/lane\n"
    if "--write" in argv:
        path = argv[argv.index("--write") + 1]
        with open(path, "w") as f:
            f.write(txt)
    print(txt)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
