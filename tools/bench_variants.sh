#!/bin/bash
run() {
  echo "== $*"
  env "$@" python bench.py --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('steps %d warm %d value %.3e  ms/step %.4f kernel_ms %.4f  GB/s %.0f  frac %.3f' % (d['steps'], d['warmup'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']))
"
}
EXTRA="--steps 30 --warmup 5" run A=1
EXTRA="--steps 30 --warmup 5" run A=2
EXTRA="--steps 30 --warmup 5" run A=3
EXTRA="--steps 100 --warmup 20" run A=4
EXTRA="--steps 100 --warmup 20" run CAMPX_LIB=$PWD/tools/probes/libcampx_stepstream.so
EXTRA="--steps 100 --warmup 20" run A=5
EXTRA="--steps 100 --warmup 20" run CAMPX_LIB=$PWD/tools/probes/libcampx_stepstream.so
