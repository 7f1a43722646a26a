#!/bin/bash
# A/B session (through gpurun): GPU tests, then every variant under build/variants on the
# three BASELINE games, then profiles of the default build.
#   tools/gpu_variants.sh <tag> [--no-tests] [--profile] variants...
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
tests=1; prof=0
while [ $# -gt 0 ]; do
  case "$1" in
    --no-tests) tests=0; shift;;
    --profile) prof=1; shift;;
    *) break;;
  esac
done
if [ $tests = 1 ]; then
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
fi
for rep in 1 2; do
for v in "$@"; do
  for g in boat_race sokoban wall_world; do
    CAMPX_LIB=build/variants/$v/libcampx_hip.so timeout 300 python bench.py --game $g --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/var_${v}_${g}_$rep.log 2>&1
    echo "variant $v $g rep$rep rc=$? $(tail -1 $O/var_${v}_${g}_$rep.log | python3 -c 'import sys,json
try:
  d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_per_step=%.4f kernel_ms=%.4f median=%.4f min=%.4f frac=%.3f" % (d["ms_per_step"], r["kernel_ms"], r["per_launch_ms"]["median"], r["per_launch_ms"]["min"], r["frac"]))
except Exception as e: print("parse-fail", e)')"
  done
done
done
if [ $prof = 1 ]; then
  bash tools/profile.sh ${tag}_boat_race
  bash tools/profile.sh ${tag}_sokoban --game sokoban
  bash tools/profile.sh ${tag}_wall_world --game wall_world
fi
