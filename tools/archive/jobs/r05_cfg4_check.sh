cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_epilogue.py tests/test_gpu_poison.py -m gpu -q -x -k "embedding or gine or batchnorm or bn or norm or zinc or graph or poison or heap or buffers or epilogue or fold or kagin" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "train_graph or one_tape_node or folds or harness or golden" 2>&1 | tail -2
bash tools/jobs/r05_cfg4_launches.sh 2>&1 | head -34
