#!/bin/bash
# Kernel trace of bench.py under library variants (through gpurun):
#   tools/gpu_variant_trace.sh <tag> "<batches>" <game> variant ...     ("default" = in-tree)
set -u
tag=$1; batches=$2; game=$3; shift; shift; shift
export TMPDIR=/tmp
for v in "$@"; do
  [ "$v" = default ] && unset CAMPX_LIB || export CAMPX_LIB=$GRAFT_REPO_ROOT/build/variants/$v/libcampx_hip.so
  bash $GRAFT_REPO_ROOT/tools/gpu_small_trace.sh $tag/$v "$batches" $game | sed "s/^/[$v] /"
done
