#!/bin/bash
# usage: tools/prof.sh <tag>   -- rocprofv3 passes over a short bench run (GPU box only)
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-traffic --no-fp32"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_a -o a -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_b -o b -- $CMD > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f -o f -- $CMD > $OUT/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w -o w -- $CMD > $OUT/pmc_w.log 2>&1
find $OUT -name "*.csv" | head -30
du -sh $OUT
