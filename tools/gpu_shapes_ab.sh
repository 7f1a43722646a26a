#!/bin/bash
# A/B of library variants on the shape-tier bench (through gpurun): tools/gpu_shapes_ab.sh variants...
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for v in "$@"; do
  CAMPX_LIB=build/variants/$v/libcampx_hip.so python tools/bench_shapes.py 2>&1 | grep "hello_world" | sed "s/^/AB $v rep$rep /"
done
done
