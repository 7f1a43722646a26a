#!/bin/bash
# Kernel trace of Engine.play() per frame: duration of the step kernel and the start-to-start
# interval between consecutive calls (through gpurun).   tools/gpu_play_trace.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
d=/tmp/playtrace
rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $d -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_play.py > $O/play.log 2>&1)
cat $O/play.log | tail -8
python3 - $d/trace_results.db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, grid_x, start, end from kernels order by start").fetchall()
by = {}
for n, g, s, e in rows:
  n = n.replace('(anonymous namespace)::', '').split('(')[0]
  by.setdefault((n, g), []).append((s, e))
for (n, g), v in sorted(by.items(), key=lambda kv: -len(kv[1]))[:8]:
  dur = [e - s for s, e in v]
  gaps = sorted(v[i + 1][0] - v[i][0] for i in range(len(v) - 1))
  med = gaps[len(gaps) // 2] if gaps else 0
  print('PT %-50s grid=%8d n=%4d dur avg=%.2f min=%.2f us  start-to-start median=%.2f us' % (
      n[:50], g, len(v), sum(dur) / len(dur) / 1e3, min(dur) / 1e3, med / 1e3))
PY
