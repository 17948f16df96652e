cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 2000 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/suite.log 2>&1; tail -16 gpurun_out/suite.log
cp gpurun_out/parity_errors.json gpurun_out/suite_parity_errors.json
tools/prof_cfg.sh 4 r05_cfg4 60 > gpurun_out/r05_config4_kernel_trace.txt 2>&1
rm -rf gpurun_out/prof_r05_cfg4
head -4 gpurun_out/r05_config4_kernel_trace.txt
