#!/bin/bash
# usage: tools/build_variant.sh <git-rev> <name>   -- build kagnn_amd/csrc of <git-rev> into kagnn_amd/lib/libkagnn_hip_<name>.so
# (A/B timing on one GPU box:  KAGNN_LIB=$PWD/kagnn_amd/lib/libkagnn_hip_<name>.so python bench.py ...)
set -e
REV=$1; NAME=$2
T=$(mktemp -d)
git archive $REV kagnn_amd/csrc include | tar -x -C $T
OBJS=""
for f in $T/kagnn_amd/csrc/*.hip; do
  if [ "$(basename $f)" == "rccl_sharded.hip" ]; then continue; fi      # (libkagnn_rccl.so's source, not part of the core library)
  o=$T/$(basename $f .hip).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -DNDEBUG -c $f -o $o &
  OBJS="$OBJS $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o kagnn_amd/lib/libkagnn_hip_$NAME.so $OBJS
rm -rf $T
echo built kagnn_amd/lib/libkagnn_hip_$NAME.so
