#!/bin/bash
# A/B the kernel's tuning knobs on the GPU box: prints one bench line per variant.
run() {
  echo "== $*"
  env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.3e  ms/step %.4f kernel_ms %.4f  GB/s %.0f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']))
"
}
for rep in 1 2; do
for w in 1 2 4; do
run CAMPX_RENDER_PER_THREAD=$w
done
done
run CAMPX_SPLIT=0
