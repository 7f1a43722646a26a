#!/bin/bash
# Wide tier (boards above 128 cells): parity tests, bench lines for maze16 / maze32, and the
# profile set (kernel trace + PMC passes, summarised on the box) - through gpurun.
#   tools/gpu_wide.sh <tag> [notest] [noprofile]
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
case " $* " in *" notest "*) ;; *)
  timeout 900 python -m pytest tests/test_wide_parity.py -m gpu -q -x 2>&1 | tail -3;;
esac
for g in maze16 maze32; do
  python bench.py --game $g --no-extras --steps 30 --warmup 10 > $O/bench_$g.json 2> $O/bench_$g.err
  python3 - "$O/bench_$g.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(d['config']['workload'], 'B =', d['config'].get('batch'), '%.3f ms/step' % d['ms_per_step'],
      '%.3e env-steps/s' % d['value'], 'frac %.3f' % r['frac'], r['kernel'],
      'cpu %.3e on %d cores' % (d['cpu_baseline']['value'], d['cpu_baseline']['cores']))
PY
done
case " $* " in *" noprofile "*) exit 0;; esac
bash tools/gpu_profile_all.sh $tag maze16 maze32
