cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "current_stream or two_host_threads" 2>&1 | tail -25
