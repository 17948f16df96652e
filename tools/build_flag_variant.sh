#!/bin/bash
# usage: tools/build_flag_variant.sh <name> <tu.hip> "<extra hipcc flags>"  -- rebuild ONE translation unit of the current tree with
# extra flags (e.g. -DKAGNN_EXPERIMENT_X) and link it with the other current objects into kagnn_amd/lib/libkagnn_hip_<name>.so
# (A/B on one GPU box: KAGNN_LIB=$PWD/kagnn_amd/lib/libkagnn_hip_<name>.so python bench.py ...)
set -e
NAME=$1; TU=$2; FLAGS=$3
python -m kagnn_amd._build > /dev/null
T=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -DNDEBUG $FLAGS -c kagnn_amd/csrc/$TU -o $T/v.o
OBJS=""
for o in kagnn_amd/lib/obj/*.o; do
  if [ "$(basename $o .o)" == "rccl_sharded" ]; then continue; fi      # (libkagnn_rccl.so's object, not part of the core library)
  if [ "$(basename $o .o)" == "$(basename $TU .hip)" ]; then OBJS="$OBJS $T/v.o"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o kagnn_amd/lib/libkagnn_hip_$NAME.so $OBJS
rm -rf $T
echo built kagnn_amd/lib/libkagnn_hip_$NAME.so
