#!/bin/bash
# A/B of pipelined vs in-order rollouts (through gpurun).
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-pipe}
mkdir -p $O
timeout 900 python -m pytest tests/test_torch_ops.py -m gpu -q 2>&1 | tail -5
for rep in 1 2 3; do
  for g in boat_race sokoban wall_world; do
    for mode in "" "--pipeline"; do
      timeout 300 python bench.py --game $g --steps 40 --warmup 5 --no-cpu-baseline --no-extras $mode > $O/${g}_${rep}_${mode:-inorder}.log 2>&1
      echo "$g rep$rep ${mode:-in-order} rc=$? $(tail -1 $O/${g}_${rep}_${mode:-inorder}.log | python3 -c 'import sys,json
try:
  d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_per_step=%.4f kernel_ms=%.4f median=%.4f min=%.4f frac=%.3f" % (d["ms_per_step"], r["kernel_ms"], r["per_launch_ms"]["median"], r["per_launch_ms"]["min"], r["frac"]))
except Exception as e: print("parse-fail", e)')"
    done
  done
done
