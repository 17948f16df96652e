set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "csr or gine or zinc or embedding" -q --durations=5 > gpurun_out/run5_tests.log 2>&1; tail -8 gpurun_out/run5_tests.log
for i in 1 2 3; do python tools/configs_sweep.py 4 2>&1 | tail -1; done | tee gpurun_out/run5_configs.log
tools/prof_cfg.sh 4 r05_cfg4 60 2>&1 | tail -64 > gpurun_out/run5_cfg4_trace.log
