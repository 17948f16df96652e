cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_epilogue.py -q -k "wide or moments or hidden128 or fuzz or cora" > gpurun_out/run10_tests.log 2>&1; tail -6 gpurun_out/run10_tests.log
