#!/bin/bash
# Kernel trace of the rollout at small and middle batches (through gpurun):
#   tools/gpu_small_trace.sh <tag> "<batches>" [game] [NAME=V ...]     (BENCH_FLAGS=--deferred: more bench.py flags)
set -u
tag=$1; batches=$2; game=${3:-boat_race}; shift; shift; shift || true
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
for b in $batches; do
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/${game}_$b
  mkdir -p $out
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --game $game --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-} > $out/bench.json 2> $out/trace.log
  cd $GRAFT_REPO_ROOT
  python3 tools/rocpd_summary.py $out > $out/summary.txt 2>&1
  echo "== $game B=$b $(python3 -c "import json;d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]);print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))")"
  grep "campx_impl" $out/summary.txt | cut -c1-150 | head -6
  find $out -name "*.db" -delete
done
