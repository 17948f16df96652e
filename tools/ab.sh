#!/bin/bash
# usage: tools/ab.sh <name> [reps]   -- alternate bench runs of the current library and kagnn_amd/lib/libkagnn_hip_<name>.so
NAME=$1; REPS=${2:-3}
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.05})'
for i in $(seq $REPS); do
  python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 | python -c "$P" new
  KAGNN_LIB=$PWD/kagnn_amd/lib/libkagnn_hip_$NAME.so python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 | python -c "$P" $NAME
done
