cd ${GRAFT_REPO_ROOT:-.}
python tools/experiments/embedding_bwd_scaling.py 2>&1 | grep -v amdgpu.ids
