cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_poison.py -m gpu -q -x --durations=5 > gpurun_out/poison.log 2>&1; tail -40 gpurun_out/poison.log
