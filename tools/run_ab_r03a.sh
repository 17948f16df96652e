set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_epilogue.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t1.log
P='import json,sys; d=json.loads(sys.stdin.read()); e=d["entry_points_ms_per_step"]; print(sys.argv[1], round(d["ms_per_step"],4), {k[6:]:round(v,3) for k,v in e.items() if v>0.04})'
B="python bench.py --no-cpu-baseline --no-extras --no-traffic --no-fp32 --steps 20"
for i in 1 2 3; do
  $B | python -c "$P" new >> gpurun_out/ab.log
  KAGNN_DW_NTO=2 $B | python -c "$P" nto2 >> gpurun_out/ab.log
  KAGNN_LIB=$PWD/kagnn_amd/lib/libkagnn_hip_r02.so $B | python -c "$P" r02 >> gpurun_out/ab.log
done
for w in config3 fastkan; do
  $B --workload $w | python -c "$P" new-$w >> gpurun_out/ab.log
  KAGNN_LIB=$PWD/kagnn_amd/lib/libkagnn_hip_r02.so $B --workload $w | python -c "$P" r02-$w >> gpurun_out/ab.log
done
cat gpurun_out/t1.log gpurun_out/ab.log
