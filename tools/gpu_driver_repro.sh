#!/bin/bash
# One gpurun session: the driver's bench command over and over (tools/driver_repro.py), as it is,
# with the gather a third of the way in, and without the process group.   tools/gpu_driver_repro.sh <tag> [N]
set -u
tag=$1; n=${2:-10}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
{
timeout 1200 python tools/driver_repro.py $n
timeout 1200 python tools/driver_repro.py $n -- --gather-at 0.34
timeout 1200 python tools/driver_repro.py $n -- --no-group
} > $O/repro.txt 2>&1
cat $O/repro.txt
