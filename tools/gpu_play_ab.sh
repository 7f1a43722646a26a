#!/bin/bash
# Engine.play() A/B on one box: the row-group-major step kernel (default) against the
# 64-environments-per-wave one (CAMPX_NO_ROWS_STEP=1): parity tests, wall time per call,
# kernel trace.   tools/gpu_play_ab.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
[ -n "${SKIP_TESTS:-}" ] || timeout 900 python -m pytest tests/test_fused_parity.py tests/test_fuzz_parity.py tests/test_torch_ops.py tests/test_tabulate.py -m gpu -q -x -k "play or fuzz or sixteen or graph or tabulate or local" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for mode in ${MODES:-rows rows_nt old}; do
  unset CAMPX_NO_ROWS_STEP CAMPX_STEP_NT CAMPX_LIB CAMPX_STEP_LDS_KB
  case $mode in
    old) export CAMPX_NO_ROWS_STEP=1;;
    rows_nt) export CAMPX_STEP_NT=1;;
    lds*) export CAMPX_STEP_LDS_KB=${mode#lds};;
    ntlds*) export CAMPX_STEP_NT=1 CAMPX_STEP_LDS_KB=${mode#ntlds};;
    sw*) export CAMPX_LIB=$GRAFT_REPO_ROOT/build/variants/$mode/libcampx_hip.so;;
  esac
  d=/tmp/playtrace_$mode
  rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $d -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_play.py > $O/play_$mode.log 2>&1)
  echo "== $mode (under rocprofv3)"; grep "B= 65536" $O/play_$mode.log
  python3 - $d/trace_results.db $mode <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, grid_x, start, end from kernels order by start").fetchall()
by = {}
for n, g, s, e in rows:
  n = n.replace('(anonymous namespace)::', '').replace('campx_impl::', '').split('(')[0]
  by.setdefault((n, g), []).append((s, e))
for (n, g), v in sorted(by.items(), key=lambda kv: -len(kv[1]))[:6]:
  dur = [e - s for s, e in v]
  gaps = sorted(v[i + 1][0] - v[i][0] for i in range(len(v) - 1))
  med = gaps[len(gaps) // 2] if gaps else 0
  print('PT[%s] %-46s grid=%8d n=%4d dur avg=%.2f min=%.2f us  start-to-start median=%.2f us' % (
      sys.argv[2], n[:46], g, len(v), sum(dur) / len(dur) / 1e3, min(dur) / 1e3, med / 1e3))
PY
  timeout 300 python tools/bench_play.py > $O/play_${mode}_plain.log 2>&1
  echo "== $mode (no profiler)"; grep "B= 65536" $O/play_${mode}_plain.log
done
