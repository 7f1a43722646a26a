#!/bin/bash
# Kernel trace of Engine.play() for every library game (through gpurun):
#   tools/gpu_play_games_trace.sh <tag> [B] [env assignments ...]
set -u
tag=$1; B=${2:-65536}; shift; shift || true
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $GRAFT_REPO_ROOT/tools/play_trace_games.py $B > $out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
grep "per play" $out/trace.log
python3 tools/rocpd_summary.py gpurun_out/$tag > gpurun_out/$tag/summary.txt 2>&1
grep -v "^#" gpurun_out/$tag/summary.txt | head -40
find gpurun_out/$tag -name "*.db" -delete
