R=$PWD; OUT=$R/gpurun_out/prof_small; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/tools/host_overhead_probe.py 125000 > $OUT/trace.log 2>&1
cd $R; tail -1 $OUT/trace.log; python tools/profsum.py $OUT | head -14
