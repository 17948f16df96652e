#!/bin/bash
# usage: tools/run_ab_libs.sh "<name1> <name2> ..." [reps] [bench flags]  -- alternate bench runs over builds of the library on ONE box
# (name "cur" = kagnn_amd/lib/libkagnn_hip.so, anything else = kagnn_amd/lib/libkagnn_hip_<name>.so); appends to gpurun_out/ab.log
NAMES=$1; REPS=${2:-3}; shift 2
mkdir -p gpurun_out
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.04})'
for i in $(seq $REPS); do
  for n in $NAMES; do
    if [ "$n" == "cur" ]; then L=$PWD/kagnn_amd/lib/libkagnn_hip.so; else L=$PWD/kagnn_amd/lib/libkagnn_hip_$n.so; fi
    KAGNN_LIB=$L python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 "$@" 2>/dev/null | python -c "$P" $n | tee -a gpurun_out/ab.log
  done
done
