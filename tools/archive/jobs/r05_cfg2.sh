cd ${GRAFT_REPO_ROOT:-.}
python tools/experiments/cfg2_eager_vs_graphed.py 2>&1 | grep -v amdgpu.ids | tail -14
