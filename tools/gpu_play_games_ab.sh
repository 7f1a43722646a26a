#!/bin/bash
# Kernel-trace A/B of Engine.play() across knob settings (through gpurun):
#   tools/gpu_play_games_ab.sh <tag> <B> "NAME=V,NAME=V" ...   ("none" = defaults)
set -u
tag=$1; B=$2; shift; shift
export TMPDIR=/tmp
for knobs in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/$(echo $knobs | tr ',=' '__')
  mkdir -p $out
  ( IFS=,; for kv in $knobs; do [ "$kv" != none ] && export "$kv"; done
    cd /tmp
    rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $GRAFT_REPO_ROOT/tools/play_trace_games.py $B > $out/trace.log 2>&1 )
  cd $GRAFT_REPO_ROOT
  python3 tools/rocpd_summary.py $out > $out/summary.txt 2>&1
  echo "== $knobs (B=$B)"
  grep "step_\|FillFunctor<signe" $out/summary.txt | awk '$2 >= 100 || $3 >= 100' | cut -c1-150
  find $out -name "*.db" -delete
done
