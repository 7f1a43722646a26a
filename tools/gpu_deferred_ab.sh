#!/bin/bash
# Rollouts pipelined across calls (bench.py --deferred: update pass of rollout i+1 and render
# pass of rollout i in one launch) against the two launches per rollout (through gpurun):
#   tools/gpu_deferred_ab.sh <tag> "<batches>" "<games>" [test]
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
if [ "${4:-}" = test ]; then
  timeout 900 python -m pytest tests/test_deferred.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
fi
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-16s %-10s B=%7d  %8.4f ms  kernel %8.4f  frac %.3f' % ('$1', '$2', $3, d['ms_per_step'], r['kernel_ms'], r['frac']))"; }
for g in $3; do for b in $2; do
  timeout 300 python bench.py --game $g --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras 2>>$O/stderr.log | line two-launches $g $b
  CAMPX_PIPE_LEAN_B=100000000 timeout 300 python bench.py --game $g --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras --deferred 2>>$O/stderr.log | line deferred-nat $g $b
  CAMPX_PIPE_LEAN_B=0 timeout 300 python bench.py --game $g --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras --deferred 2>>$O/stderr.log | line deferred-6w $g $b
done; done | tee $O/deferred_ab.txt
