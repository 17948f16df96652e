cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_epilogue.py tests/test_gpu_fullsize.py tests/test_gpu_half.py -q -k "not arxiv and not seeds and not 4gib" > gpurun_out/run9_tests.log 2>&1; tail -12 gpurun_out/run9_tests.log
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.04})'
for i in 1 2; do
for w in 1 0; do KAGNN_FWD_WIDE=$w python bench.py --workload config3 --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 2>/dev/null | python -c "$P" wide=$w; done
done | tee gpurun_out/run9_ab.log
python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 2>/dev/null | python -c "$P" headline | tee -a gpurun_out/run9_ab.log
