#!/bin/bash
# Round-6 final GPU session (through gpurun): the whole GPU suite, the driver's bench command
# several times (a fresh process each), the launcher path with one RCCL rank.
#   tools/gpu_r6_final.sh <tag> [repeats]
set -u
tag=$1; n=${2:-6}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-300
for i in $(seq 1 $n); do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.log 2>$O/bench_$i.err
  python3 - <<PY
import json
d = json.loads(open('$O/bench_$i.log').read().strip().splitlines()[-1])
c, r = d['config'], d['roofline']
print('RUN $i value %.4g ms %.4f kms %.4f ratio %.3f frac %.4f ofmeas %.3f ceil %.0f | ww %s | sk %s | hw %s | play %.2f graph %.2f policy %.1f/%.1f' % (
    d['value'], d['ms_per_step'], r['kernel_ms'], d['ms_per_step'] / r['kernel_ms'], r['frac'], r['frac_of_measured'], r['measured_write_ceiling_gbs'],
    c['wall_world_262144'].split('frac=')[1][:6], c['sokoban_131072'].split('frac=')[1][:6], c['hello_world_32768'].split('frac=')[1][:6],
    c['play_us_per_call'], c['play_graph32_us_per_call'], c['policy_eager_us_per_frame'], c['policy_in_graph_us_per_frame']))
PY
done
timeout 600 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_dist.log 2>$O/bench_dist.err; echo "force-dist rc=$?"; tail -1 $O/bench_dist.log | cut -c1-400
