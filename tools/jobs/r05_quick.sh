cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_epilogue.py tests/test_gpu_poison.py -m gpu -q -x 2>&1 | tail -3
