cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_poison.py -m gpu -q -x -k "l1_loss or train_graph or zinc or embedding or graph_level or graph-level" 2>&1 | tail -4
