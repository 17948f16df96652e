cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python tools/host_profile_cfg4.py 2>&1 | grep -v amdgpu.ids > gpurun_out/hostprof.txt; head -150 gpurun_out/hostprof.txt | cut -c1-160
