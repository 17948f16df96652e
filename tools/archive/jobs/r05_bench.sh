cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( time python bench.py --no-cpu-baseline --no-traffic ) 2> gpurun_out/bench_time.txt | tail -1 > gpurun_out/bench_line.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line.json").read())
print(round(d["ms_per_step"], 3), d["roofline"]["frac"], d["secondary"]["model_step"]["ms_per_step"], d["secondary"]["graph_level_step"])
PY
tail -3 gpurun_out/bench_time.txt
