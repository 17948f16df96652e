#!/bin/bash
# usage: tools/ab_env.sh VAR [reps]   -- alternate bench runs with VAR=1 and VAR=0 on one box
VAR=$1; REPS=${2:-3}
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.04})'
for i in $(seq $REPS); do
  env $VAR=1 python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 | python -c "$P" "$VAR=1"
  env $VAR=0 python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20 | python -c "$P" "$VAR=0"
done
