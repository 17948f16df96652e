cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
tools/make_profiles.sh r05 > gpurun_out/make_profiles_r05.log 2>&1
tail -30 gpurun_out/make_profiles_r05.log
cat gpurun_out/profiles_r05/r05_bench.json | head -c 1500
