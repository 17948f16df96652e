cd ${GRAFT_REPO_ROOT:-.}
python tools/experiments/adam_host_time.py 2>&1 | grep -v amdgpu.ids | head -40 | cut -c1-160
