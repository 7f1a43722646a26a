#!/bin/bash
# Does what precedes the timed region matter?  Alternating runs of bench.py (through gpurun).
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for a in "--steps 30 --warmup 5" "--steps 30 --warmup 50" "--steps 100 --warmup 5" "--steps 100 --warmup 50"; do
    python bench.py --no-cpu-baseline --no-extras $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WARM %-28s %.4f %.3f' % ('$a', d['ms_per_step'], d['roofline']['frac']))"
  done
done
