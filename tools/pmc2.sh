#!/bin/bash
# usage: tools/pmc2.sh <tag> [env assignments...]  -- kernel trace + the two SQ counter passes over a 3-step bench run, summarised
# into gpurun_out/pmc2_<tag>.txt (tools/profsum.py)
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc2_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env $* python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-traffic --no-fp32"
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_a -o a -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_b -o b -- $CMD > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT -d $OUT/pmc_c -o c -- $CMD > $OUT/pmc_c.log 2>&1
cd $ROOT
python tools/profsum.py $OUT | grep -A1 -E "^kernel|^kan_|^agg_rows" > gpurun_out/pmc2_$TAG.txt
rm -rf $OUT
cat gpurun_out/pmc2_$TAG.txt
