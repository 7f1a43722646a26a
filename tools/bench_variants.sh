#!/bin/bash
# A/B the kernel's tuning knobs on the GPU box: prints one bench line per variant.
for notable in 0 1; do
for envs in 64 32; do
for nt in 1 0; do
  echo "== CAMPX_NO_TABLE=$notable CAMPX_ENVS_PER_WAVE=$envs CAMPX_STORE_NT=$nt"
  CAMPX_NO_TABLE=$notable CAMPX_ENVS_PER_WAVE=$envs CAMPX_STORE_NT=$nt python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.3e  kernel_ms %.4f  GB/s %.0f  frac %.3f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']))
"
done
done
done
