#!/bin/bash
# usage: tools/make_profiles.sh <tag>   (GPU box)  -- everything the committed profiles/ summaries are made from:
#   rocprofv3 --kernel-trace --stats of the default bench command, separate --pmc passes (counters + HBM traffic),
#   kernel traces of the other BASELINE.json configs.  Text summaries land in gpurun_out/profiles_<tag>/ (copy to profiles/).
set -u
TAG=${1:-r01}
R=$PWD; OUT=$R/gpurun_out/profiles_$TAG; RAW=$R/gpurun_out/prof_$TAG
rm -rf $OUT $RAW; mkdir -p $OUT
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
tools/prof.sh $TAG > /dev/null 2>&1
python tools/profsum.py $RAW > $OUT/${TAG}_bench_kernel_trace_and_pmc.txt
python - <<PY
import json, re
txt = open("$OUT/${TAG}_bench_kernel_trace_and_pmc.txt").read()
out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 3 (tools/prof.sh); per-launch "
               "averages; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 -- the x2 on FETCH_SIZE is the gfx950 correction of "
               "MI355X_MICROARCH.md (HBM) for 16 B/lane reads, which is what agg_rows_v4_kernel issues; WRITE_SIZE uncalibrated"}
m = re.search(r"\nagg_rows_v4_kernel<16(?:, false)?>\n(.*)", txt)
if m:
    vals = dict(kv.split("=") for kv in m.group(1).split())
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        out["kagnn_aggregate_sum"] = int((2 * float(vals["FETCH_SIZE"]) + float(vals["WRITE_SIZE"])) * 1024)
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(out)
PY
for k in 2 3 5 1 4; do tools/prof_cfg.sh $k ${TAG}_cfg$k 40 > $OUT/${TAG}_config${k}_kernel_trace.txt 2>&1; done
tools/prof_model.sh > $OUT/${TAG}_model_step_kernel_trace.txt 2>&1
KAGNN_ACT=bf16 python tools/configs_sweep.py 2 2>/dev/null | grep "cfg2 " > $OUT/${TAG}_config2_bf16.txt
# BASELINE config 2's build-defined modes side by side: fp32 storage / bf16 gathers x split (three products) / half (one product)
for a in fp32 bf16; do for p in split half; do
  echo "KAGNN_ACT=$a KAGNN_PRECISION=$p: $(KAGNN_ACT=$a KAGNN_PRECISION=$p python tools/configs_sweep.py 2 2>/dev/null | grep 'cfg2 ')"
done; done > $OUT/${TAG}_config2_modes.txt
python bench.py --precision half --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | tail -1 > $OUT/${TAG}_bench_half.json
python bench.py --precision half --act bf16 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | tail -1 > $OUT/${TAG}_bench_half_bf16_gather.json
python bench.py --act bf16 --no-cpu-baseline --no-extras --no-fp32 2>/dev/null | tail -1 > $OUT/${TAG}_bench_bf16_gather.json
python bench.py --workload config3 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | tail -1 > $OUT/${TAG}_bench_config3.json
python bench.py --workload fastkan 2>/dev/null | tail -1 > $OUT/${TAG}_bench_fastkan.json
tools/prof_fastkan.sh ${TAG}_fk 40 > $OUT/${TAG}_fastkan_layer_kernel_trace.txt 2>&1
python tools/shard_plan_probe.py --link-gbs=50 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_shard_plan.txt; cp gpurun_out/shard_plan.json $OUT/${TAG}_shard_plan.json
# round 6: the other sharded paths (FastKAN-GIN layer, GKAN_Nodes step), the model workload as a bench line, config 4's step repeated
python tools/shard_plan_probe_e2.py --link-gbs=50 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_shard_plan_e2.txt; cp gpurun_out/shard_plan_e2.json $OUT/${TAG}_shard_plan_e2.json
python bench.py --workload model 2>/dev/null | tail -1 > $OUT/${TAG}_bench_model.json
python tools/cfg4_step.py 2>/dev/null | tail -1 > $OUT/${TAG}_config4_step.json
KAGNN_CFG4_MT=1 python tools/cfg4_step.py 2>/dev/null | tail -1 > $OUT/${TAG}_config4_step_autograd_threads.json
python tools/kernel_resources.py > $OUT/${TAG}_kernel_resources.txt 2>&1
rm -rf $R/gpurun_out/prof_${TAG}* 
ls -la $OUT
