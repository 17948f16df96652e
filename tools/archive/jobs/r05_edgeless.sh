cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python tools/experiments/edgeless_graphs.py 2>&1 | grep -v amdgpu.ids | tail -30
