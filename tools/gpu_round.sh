#!/bin/bash
# One GPU session (run through gpurun): smoke, GPU tests, bench, launcher, A/B variants, profile.
#   tools/gpu_round.sh <tag> [variants...]
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-extras > $O/bench_dist.log 2>&1; echo "bench_dist rc=$?"; tail -1 $O/bench_dist.log
for v in "$@"; do
  for g in boat_race sokoban wall_world; do
    CAMPX_LIB=build/variants/$v/libcampx_hip.so timeout 300 python bench.py --game $g --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/var_${v}_$g.log 2>&1
    echo "variant $v $g rc=$? $(tail -1 $O/var_${v}_$g.log | python3 -c 'import sys,json
try:
  d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_per_step=%.4f kernel_ms=%.4f median=%.4f frac=%.3f" % (d["ms_per_step"], r["kernel_ms"], r["per_launch_ms"]["median"], r["frac"]))
except Exception as e: print("parse-fail", e)')"
  done
done
mkdir -p $O/prof
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof/trace.log 2>&1
cd "$GRAFT_REPO_ROOT"; echo "prof rc=$?"; ls $O/prof | head
