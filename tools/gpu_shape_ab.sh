#!/bin/bash
# Shape tier under library variants (through gpurun): tools/gpu_shape_ab.sh <tag> variant ...
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$1; tag=$1; shift
for v in "$@"; do
  [ "$v" = default ] && unset CAMPX_LIB || export CAMPX_LIB=$GRAFT_REPO_ROOT/build/variants/$v/libcampx_hip.so
  echo "== $v"
  if [ "$v" != default ]; then timeout 600 python -m pytest tests/test_shape_parity.py -m gpu -x -q 2>&1 | tail -1; fi
  python tools/bench_shapes.py 2>/dev/null | grep "TB/s" | grep -v split
done | tee gpurun_out/$tag/shape_ab.txt
