#!/bin/bash
# Rollouts whose trace outgrows the caches, run as chunks of frames (launch_split): kernel times
# for several chunk sizes (through gpurun).   tools/gpu_chunk_ab.sh <tag> "<bench args>" mb...
tag=$1; args=$2; shift 2
for mb in "$@"; do
  echo "== CAMPX_TRACE_CHUNK_MB=$mb  $args"
  CAMPX_TRACE_CHUNK_MB=$mb BENCH_ARGS="$args" "$GRAFT_REPO_ROOT"/tools/gpu_ktrace.sh $tag boat_race base | grep KT
done
