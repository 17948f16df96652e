cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_epilogue.py -q -x -k "kanlinear or g2 or g3 or chain or hidden128 or ragged or moments or wide or gin" > gpurun_out/run8_tests.log 2>&1; tail -6 gpurun_out/run8_tests.log
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.04})'
for i in 1 2 3; do
for w in 1 0; do KAGNN_FWD_WIDE=$w python bench.py --workload config3 --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 2>/dev/null | python -c "$P" wide=$w; done
done | tee gpurun_out/run8_ab.log
