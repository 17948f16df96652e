cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 2000 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/suite.log 2>&1; tail -12 gpurun_out/suite.log
cp gpurun_out/parity_errors.json gpurun_out/suite_parity_errors.json
( time python bench.py ) 2> gpurun_out/bench_time.txt | tail -1 > gpurun_out/bench_line.json
tail -3 gpurun_out/bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line.json").read())
print(round(d["ms_per_step"], 3), d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["secondary"]["model_step"]["ms_per_step"], d["secondary"]["graph_level_step"]["ms_per_step"])
PY
