#!/bin/bash
# usage: tools/prof_cfg.sh <config number> [tag]  -- rocprofv3 kernel trace of one configs_sweep.py config (GPU box only)
set -u
K=$1; TAG=${2:-cfg$K}
R=$PWD; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/tools/configs_sweep.py $K > $OUT/trace.log 2>&1
cd $R
grep cfg $OUT/trace.log
python tools/profsum.py $OUT | head -${3:-30}
