cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/suite.log 2>&1; tail -40 gpurun_out/suite.log
