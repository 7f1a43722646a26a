#!/bin/bash
# Round-6 GPU check of what changed since the last full session (through gpurun):
#   tools/gpu_r6_quick.sh <tag> "<pytest -k expression or test files>"
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 2400 python -m pytest $* -m gpu -q --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open('$O/bench.log').read().strip().splitlines()[-1])
c = d['config']
print('value %.4g ms %.4f kms %.4f frac %.4f ofmeas %.4f ceiling %.0f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['frac_of_measured'], d['roofline']['measured_write_ceiling_gbs']))
for k in sorted(c):
  if isinstance(c[k], str) and c[k].startswith('v='): print(k, c[k])
  if k.startswith(('play', 'policy', 'deferred')): print(k, c[k])
PY
tail -3 $O/bench.err | cut -c1-300
