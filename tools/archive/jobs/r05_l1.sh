cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_poison.py -m gpu -q -x -k "adam or l1_loss or train_graph or zinc or embedding or graph_level or graph-level" 2>&1 | tail -3
python tools/host_profile_cfg4.py 2>&1 | grep -v amdgpu.ids > gpurun_out/hostprof.txt; head -7 gpurun_out/hostprof.txt | cut -c1-250
