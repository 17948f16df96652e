set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_half.py "tests/test_gpu_fullsize.py::test_fullsize_one_call_gin_layer_backward_vs_oracle" tests/test_gpu_parity.py tests/test_gpu_models.py -k "half or one_call or harness or gfastkan_nodes or kanlinear or g2 or g3 or chain" -x -q --durations=8 > gpurun_out/run1_tests.log 2>&1; tail -25 gpurun_out/run1_tests.log
B="python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20"
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.04})'
for i in 1 2; do
$B 2>/dev/null | python -c "$P" split | tee -a gpurun_out/run1_bench.log
$B --precision half 2>/dev/null | python -c "$P" half | tee -a gpurun_out/run1_bench.log
$B --act bf16 2>/dev/null | python -c "$P" split+bf16 | tee -a gpurun_out/run1_bench.log
$B --precision half --act bf16 2>/dev/null | python -c "$P" half+bf16 | tee -a gpurun_out/run1_bench.log
done
rm -f gpurun_out/ab.log
tools/run_ab_libs.sh "cur fwdnorefill dxw2nostage1 dxw2nostage2" 2 --workload config3
cp gpurun_out/ab.log gpurun_out/run1_ab_config3.log
