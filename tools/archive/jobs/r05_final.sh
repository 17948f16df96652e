cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 2000 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/suite.log 2>&1; tail -25 gpurun_out/suite.log
cp gpurun_out/parity_errors.json gpurun_out/suite_parity_errors.json
tools/make_profiles.sh r05 > gpurun_out/make_profiles_r05.log 2>&1
tail -5 gpurun_out/make_profiles_r05.log
