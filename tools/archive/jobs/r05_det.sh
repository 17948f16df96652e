cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python tools/experiments/train_determinism.py > gpurun_out/det.log 2>&1; tail -30 gpurun_out/det.log
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "train_graph_batches" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_sharded_gloo.py -m gpu -q -k "four_ranks_one_gpu_runs_every" 2>&1 | tail -5
