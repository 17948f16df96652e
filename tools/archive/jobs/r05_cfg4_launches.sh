cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/configs_sweep.py 4h 2>/dev/null | grep "cfg4 harness (" 
python tools/configs_sweep.py 4h 2>/dev/null | grep "cfg4 harness ("
rm -rf gpurun_out/prof_cfg4h
tools/prof_cfg.sh 4h cfg4h 90 > gpurun_out/cfg4h_trace.txt 2>&1
cat gpurun_out/cfg4h_trace.txt | cut -c1-150
rm -rf gpurun_out/prof_cfg4h
