#!/bin/bash
# usage: tools/pmc.sh <tag> <counter> [<counter> ...]  -- one rocprofv3 --pmc pass over a short bench run
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc "$@" -d $OUT/pmc_x -o x -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic --no-fp32 > $OUT/log.txt 2>&1
