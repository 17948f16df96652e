set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "csr or gine or zinc" -q --durations=5 > gpurun_out/run4_tests.log 2>&1; tail -15 gpurun_out/run4_tests.log
for i in 1 2 3; do python tools/configs_sweep.py 4 2>&1 | tail -2; KAGNN_GINE_STACK_ABI=0 python tools/configs_sweep.py 4 2>&1 | tail -2; done | tee gpurun_out/run4_configs.log
tools/prof_cfg.sh 4 r05_cfg4 60 2>&1 | tail -64 | tee gpurun_out/run4_cfg4_trace.log
