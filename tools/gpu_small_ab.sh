#!/bin/bash
# Small / middle batches: the two-kernel path, pipelined rollouts (update pass of launch i+1
# under the render of launch i) and the single fused kernel (through gpurun):
#   tools/gpu_small_ab.sh <tag> "<batches>" [game]
set -u
cd "$GRAFT_REPO_ROOT"
game=${3:-boat_race}
mkdir -p gpurun_out/$1
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %-10s B=%7d  %8.4f ms  kernel %8.4f  frac %.3f  %s' % ('$1', '$game', $2, d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel']))"; }
for b in $2; do
  python bench.py --game $game --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | line split $b
  python bench.py --game $game --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras --pipeline 2>/dev/null | line pipelined $b
  CAMPX_SPLIT=0 python bench.py --game $game --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | line fused-kernel $b
  for extra in "$@"; do :; done
done | tee -a gpurun_out/$1/small_ab.txt
