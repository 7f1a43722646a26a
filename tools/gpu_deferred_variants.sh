#!/bin/bash
# Shapes of the shared launch of deferred rollouts (build/variants/<name>, tools/build_variants.py):
#   tools/gpu_deferred_variants.sh <tag> "<batches>" "<variant names>"
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s B=%7d  %8.4f ms  kernel %8.4f  frac %.3f' % ('$1', $2, d['ms_per_step'], r['kernel_ms'], r['frac']))"; }
for b in $2; do
  timeout 300 python bench.py --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras --deferred 2>>$O/stderr.log | line default $b
  for v in $3; do
    CAMPX_LIB=build/variants/$v/libcampx_hip.so timeout 300 python bench.py --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras --deferred 2>>$O/stderr.log | line $v $b
  done
done | tee $O/deferred_variants.txt
