cd ${GRAFT_REPO_ROOT:-.}
python tools/experiments/adam_in_model_time.py 2>&1 | grep -v amdgpu.ids | tail -2
