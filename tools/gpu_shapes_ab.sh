#!/bin/bash
# Shape tier: parity tests, then tools/bench_shapes.py for the default build and A/B variants
# of the two-kernel path (window span / waves per block) - through gpurun.
#   tools/gpu_shapes_ab.sh <tag> [variants...]
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_shape_parity.py -m gpu -q -x 2>&1 | tail -3
for v in default "$@"; do
  unset CAMPX_LIB
  [ $v = default ] || export CAMPX_LIB=$GRAFT_REPO_ROOT/build/variants/$v/libcampx_hip.so
  echo "== $v"
  python tools/bench_shapes.py 2>&1 | grep "zoo4\|hello_world B=32768" | grep -v play
done
