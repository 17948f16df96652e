#!/bin/bash
# usage: tools/pmc_model.sh <tag>  -- kernel trace + two SQ counter passes over 3 steps of the full GKAN_Nodes training step
# (tools/model_step.py), summarised per kernel into gpurun_out/pmc_model_<tag>.txt (tools/profsum.py)
TAG=${1:-model}
OUT=$PWD/gpurun_out/pmc_model_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/model_step.py 3"
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_a -o a -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f -o f -- $CMD > $OUT/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w -o w -- $CMD > $OUT/pmc_w.log 2>&1
cd $ROOT
python tools/profsum.py $OUT > gpurun_out/pmc_model_$TAG.txt
rm -rf $OUT
head -60 gpurun_out/pmc_model_$TAG.txt
