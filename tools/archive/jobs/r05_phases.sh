cd ${GRAFT_REPO_ROOT:-.}
python tools/experiments/cfg4_node_phases.py 2>&1 | grep -v amdgpu.ids | tail -22
