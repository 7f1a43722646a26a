#!/bin/bash
# the one-launch rollout against the two launches (CAMPX_NO_FLOW=1), optionally with timing-experiment
# library variants whose results are WRONG (tools/build_variants.py nopoll:-DCAMPX_FLOW_NOPOLL=1: the
# render role never waits for the update role)
cd $GRAFT_REPO_ROOT
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s B=%7d  %8.4f ms  kernel %8.4f  frac %.3f' % ('$1', $2, d['ms_per_step'], r['kernel_ms'], r['frac']))"; }
export CAMPX_FLOW_MAX_B=1000000
for b in ${BATCHES:-4096 16384 65536}; do
  CAMPX_NO_FLOW=1 python bench.py --batch $b --steps 50 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | line two-launches $b
  python bench.py --batch $b --steps 50 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | line one-launch $b
  for v in ${VARIANTS-nopoll}; do
    CAMPX_LIB=build/variants/$v/libcampx_hip.so python bench.py --batch $b --steps 50 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | line one-launch-$v $b
  done
done
