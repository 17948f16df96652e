#!/bin/bash
# usage: tools/pmc_l2.sh <workload> [tag]  -- L2-side request counters of the KAN kernels (how much the packed W re-streams through
# LDS-DMA shows here, not in FETCH_SIZE: the pack stays L2 / Infinity-Cache resident).  Separate pass per counter group.
WL=${1:-config3}; TAG=${2:-l2_$WL}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic --no-fp32"
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_a -o a -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/pmc_b -o b -- $CMD > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f -o f -- $CMD > $OUT/pmc_f.log 2>&1
cd $ROOT
python tools/profsum.py $OUT | grep -A1 -E "^kernel|^kan_|^agg_rows" > gpurun_out/pmc_$TAG.txt
tail -3 $OUT/pmc_a.log $OUT/pmc_b.log | grep -i "error\|invalid" | head -5
rm -rf $OUT
cat gpurun_out/pmc_$TAG.txt
