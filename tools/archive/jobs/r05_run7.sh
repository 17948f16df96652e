cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; rm -f gpurun_out/ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -q -k "kanlinear or g2 or g3 or chain or hidden128 or ragged or fuzz_kan" > gpurun_out/run7_tests.log 2>&1; tail -4 gpurun_out/run7_tests.log
tools/run_ab_libs.sh "base cur" 3 --workload config3
cp gpurun_out/ab.log gpurun_out/run7_ab.log
