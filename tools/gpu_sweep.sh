#!/bin/bash
# Roofline fraction across batch sizes and episode lengths (through gpurun):
#   [BENCH_FLAGS=--deferred] tools/gpu_sweep.sh <game> "<batches>" "<frames>"
cd "$GRAFT_REPO_ROOT"
game=$1
for b in $2; do for t in $3; do
  # (small batches: a 30-launch window of 20-60 us launches is 1-2 ms, and what closes it - the
  # last launch's completion, the log's wait - is several per cent of that: 300 launches)
  steps=30; [ $b -lt 32768 ] && [ $t -le 400 ] && steps=300
  python bench.py --game $game --batch $b --frames $t --steps $steps --warmup 20 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('SWEEP %-10s B=%7d T=%5d  %8.4f ms  frac %.3f  %s' % ('$game', $b, $t, d['ms_per_step'], r['frac'], r['kernel'] if '--deferred' not in '${BENCH_FLAGS:-}' else 'pipe_table_kernel (deferred)'))"
done; done
