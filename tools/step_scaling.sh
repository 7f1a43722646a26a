#!/bin/bash
# kernel durations of the split path for several episode lengths (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in 16 64 128 256; do
  CAMPX_SPLIT=1 rocprofv3 --kernel-trace -d gpurun_out/prof_scaling -o f$f -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --frames $f > /dev/null 2>&1
done
