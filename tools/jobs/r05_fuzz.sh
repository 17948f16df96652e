cd ${GRAFT_REPO_ROOT:-.}
for s in 501 502 503; do timeout 900 python tools/fuzz_graph.py 100 $s 2>&1 | grep -v "^ok" | tail -8; done
