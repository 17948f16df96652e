cd ${GRAFT_REPO_ROOT:-.}
for s in 5; do timeout 1500 python tools/fuzz_graph_models.py 100 $s 2>&1 | grep "FAIL\|failures\|kink crossed" | tail -12 | cut -c1-330; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gine or zinc or model_node" 2>&1 | tail -2
