#!/bin/bash
# usage: tools/prof_fastkan.sh [tag] [lines]  -- rocprofv3 kernel trace of a short run of the FastKAN-GIN layer workload (GPU box only)
set -u
TAG=${1:-fastkan}
R=$PWD; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --workload fastkan --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-traffic --no-fp32 > $OUT/trace.log 2>&1
cd $R
grep '"metric"' $OUT/trace.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['entry_points_ms_per_step'])"
python tools/profsum.py $OUT | head -${2:-30}
