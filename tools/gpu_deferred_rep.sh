#!/bin/bash
# Repeated, interleaved runs of deferred rollouts against the two launches (run-to-run spread);
# "own-obs": an observation buffer per buffer set instead of one shared by the two
#   tools/gpu_deferred_rep.sh <tag> "<batches>" <repeats> [test]
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
if [ "${4:-}" = test ]; then
  timeout 900 python -m pytest tests/test_deferred.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
fi
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-14s B=%7d  %8.4f ms  kernel %8.4f  frac %.3f' % ('$1', $2, d['ms_per_step'], r['kernel_ms'], r['frac']))"; }
for b in $2; do for rep in $(seq 1 $3); do
  timeout 300 python bench.py --batch $b --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>>$O/stderr.log | line two-launches $b
  timeout 300 python bench.py --batch $b --steps 100 --warmup 20 --no-cpu-baseline --no-extras --deferred 2>>$O/stderr.log | line deferred $b
  CAMPX_BENCH_OWN_OBS=1 timeout 300 python bench.py --batch $b --steps 100 --warmup 20 --no-cpu-baseline --no-extras --deferred 2>>$O/stderr.log | line deferred-own-obs $b
done; done | tee $O/deferred_rep.txt
