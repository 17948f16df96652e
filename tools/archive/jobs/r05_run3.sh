set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "csr or gine or zinc or aggregat" -q --durations=5 > gpurun_out/run3_tests.log 2>&1; tail -15 gpurun_out/run3_tests.log
rm -f gpurun_out/ab.log
tools/run_ab_libs.sh "cur fwdnorefill dxw2nostage1 dxw2nostage2" 2 --workload config3
cp gpurun_out/ab.log gpurun_out/run3_ab_config3.log
for i in 1 2 3; do python tools/configs_sweep.py 4 2>&1 | tail -2; KAGNN_SMALL_CSR=0 python tools/configs_sweep.py 4 2>&1 | tail -2; done | tee gpurun_out/run3_configs.log
python tools/host_profile_cfg4.py 2>&1 | tail -45 | tee gpurun_out/run3_hostprof.log
tools/prof_cfg.sh 4 r05_cfg4 60 2>&1 | tail -70 | tee gpurun_out/run3_cfg4_trace.log
