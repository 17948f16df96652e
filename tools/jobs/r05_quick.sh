cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py -m gpu -q -x -k "arxiv_shaped_kan_gin_model_vs_oracle or current_stream or two_host or fullsize or wide_forward" 2>&1 | tail -3
