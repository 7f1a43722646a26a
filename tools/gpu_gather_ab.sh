#!/bin/bash
# Cost of the episode-return all-gather cadence in the one-rank RCCL bench (through gpurun).
cd "$GRAFT_REPO_ROOT"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for g in "$@"; do
  timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-extras --gather-every $g 2>/dev/null | show "DIST every=$g"
done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | show SOLO
