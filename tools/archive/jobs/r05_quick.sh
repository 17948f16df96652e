cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py -m gpu -q -x -k "current_stream or two_host" 2>&1 | tail -3
