#!/bin/bash
# Kernel-trace any python script (through gpurun) and print per-kernel avg/min/max.
#   tools/gpu_ktrace_cmd.sh tools/bench_play.py [args]
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
d=/tmp/ktcmd
rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $d -o trace -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/ktcmd.log 2>&1)
tail -8 /tmp/ktcmd.log
python3 - $d/trace_results.db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, c, a, mn, mx in cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name having count(*) > 20 order by sum(end-start) desc"):
  print('KT %-60s n=%5d avg=%8.2f min=%8.2f max=%8.2f' % (n.replace('(anonymous namespace)::','').split('(')[0][:60], c, a/1e3, mn/1e3, mx/1e3))
PY
