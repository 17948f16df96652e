#!/bin/bash
# usage: tools/ab_vals.sh VAR "v1 v2 ..." [reps] [extra bench flags]   -- alternate bench runs over the values of one
# environment switch on ONE box (box-to-box spread on the pool exceeds most kernel-level effects)
VAR=$1; VALS=$2; REPS=${3:-3}; shift 3
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.04})'
for i in $(seq $REPS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 "$@" | python -c "$P" "$VAR=$v"
  done
done
