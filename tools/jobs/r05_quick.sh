cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "model_node or gine" 2>&1 | tail -3
