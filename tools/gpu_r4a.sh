#!/bin/bash
# round 4, first call: the driver's bench command (closing barrier off the clock), play() per game
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r4a}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for i in 1 2 3; do
  CAMPX_BENCH_DEBUG=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/driver_$i.json 2> $O/driver_$i.err
  python3 -c 'import sys,json
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("driver-style: value=%.4g ms_per_step=%.5f kernel_ms=%.5f frac=%.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"]))' $O/driver_$i.json
  grep "TIMED WINDOW" $O/driver_$i.err
done
timeout 600 python tools/bench_play_games.py > $O/play_games.log 2>&1; tail -12 $O/play_games.log
