#!/bin/bash
# Per-kernel timings of A/B variants (through gpurun): each variant x game runs bench.py under
# rocprofv3 --kernel-trace and prints avg/min/max of the update and render kernels.
#   tools/gpu_ktrace.sh <tag> "<games>" variants...
set -u
tag=$1; games=$2; shift 2
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  for g in $games; do
    d=/tmp/kt_${v}_$g
    rm -rf $d
    (cd /tmp && CAMPX_LIB=$GRAFT_REPO_ROOT/build/variants/$v/libcampx_hip.so timeout 300 rocprofv3 --kernel-trace -d $d -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --game $g --steps 30 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $O/kt_${v}_$g.log 2>&1)
    python3 - $d/trace_results.db $v $g <<'PY'
import sqlite3, sys
db, v, g = sys.argv[1:4]
try:
  cur = sqlite3.connect(db).cursor()
  rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
                     "where name like '%render_kernel%' or name like '%update_%' or name like '%rollout_kernel%' "
                     "group by name having count(*) > 5 order by sum(end-start) desc").fetchall()
  tot = 0
  for n, c, a, mn, mx in rows:
    n = n.replace('(anonymous namespace)::', '').split('(')[0]
    tot += a
    print('KT %-10s %-10s %-44s n=%3d avg=%8.2f min=%8.2f max=%8.2f' % (v, g, n[:44], c, a/1e3, mn/1e3, mx/1e3))
  print('KT %-10s %-10s SUM avg=%8.2f %s' % (v, g, tot/1e3, __import__('os').environ.get('BENCH_ARGS','')))
  if __import__('os').environ.get('PRINT_SEQ'):
    seq = cur.execute("select end-start from kernels where name like '%render_kernel%' order by start").fetchall()
    print('SEQ', v, g, ' '.join('%.0f' % (x[0]/1e3) for x in seq))
except Exception as e:
  print('KT', v, g, 'failed', e)
PY
    rm -rf $d
  done
done
