#!/bin/bash
# Kernel trace of the device-side state enumeration (through gpurun): tools/gpu_enumerate_trace.sh <tag>
set -u
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $GRAFT_REPO_ROOT/tools/enumerate_workload.py > $out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
grep its_showtime $out/trace.log
python3 - $out/trace_results.db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), max(end-start) from kernels group by name order by sum(end-start) desc limit 14").fetchall()
total = cur.execute("select sum(end-start) from kernels").fetchone()[0]
print('%-90s %7s %10s %9s %9s' % ('kernel', 'calls', 'total ms', 'avg us', 'max us'))
for n, c, s, a, m in rows:
  print('%-90s %7d %10.2f %9.2f %9.2f' % (n.replace('(anonymous namespace)::', '').split('(')[0][:90], c, s / 1e6, a / 1e3, m / 1e3))
print('all kernels: %.1f ms' % (total / 1e6))
PY
find $out -name "*.db" -delete
