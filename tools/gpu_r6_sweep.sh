#!/bin/bash
# Round 6: the sweep across batch sizes / episode lengths, and what gathering the episode-return log
# EVERY episode costs the driver's window on one rank (through gpurun).  tools/gpu_r6_sweep.sh <tag>
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$1
bash tools/gpu_sweep_all.sh $1 > /dev/null 2>&1
bash tools/gpu_sweep_all.sh $1 wide > /dev/null 2>&1
cat gpurun_out/$1/sweep.txt gpurun_out/$1/sweep_wide.txt | cut -c1-130
{
echo "# the episode-return log gathered every E episodes (python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --gather-every E), 4 runs each"
for e in 32 4 1; do
  for i in 1 2 3 4; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --gather-every $e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('GATHER every %2d (clamped to %2d): ms_per_step %.4f kernel_ms %.4f frac %.4f gathers %d call_us_max %.0f exposed_us_max %.0f' % ($e, c['gather_every'], d['ms_per_step'], r['kernel_ms'], r['frac'], c['gather_count'], c['gather_call_us_max'], c['gather_exposed_us_max']))"
  done
done
} | tee gpurun_out/$1/gather_cadence.txt
