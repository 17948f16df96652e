cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_poison.py -m gpu -q -x -k "model_node or gine or train_graph or zinc or graph_level or graph-level or embedding" 2>&1 | tail -25
python tools/experiments/cfg4_forward_phases.py 2>&1 | grep -v amdgpu.ids | tail -3
python tools/host_profile_cfg4.py 2>&1 | grep -v amdgpu.ids | head -6 | cut -c1-250
