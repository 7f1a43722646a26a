#!/bin/bash
# Kernel trace of the shape tier (tools/bench_shapes.py) - through gpurun.  tools/gpu_shape_trace.sh <tag>
set -u
tag=$1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_shapes.py > $out/trace.log 2>&1
grep -v amdgpu $out/trace.log | grep "TB/s"
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_summary.py gpurun_out/$tag > gpurun_out/$tag/summary.txt 2>&1
find gpurun_out/$tag -name "*.db" -delete
head -16 gpurun_out/$tag/summary.txt | cut -c1-200
