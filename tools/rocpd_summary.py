#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (kernel trace + PMC passes) as text.

    python tools/rocpd_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt

Per kernel: dispatch count, avg/min/max duration from the kernel trace, resource
usage, and for each PMC database the per-dispatch average of every counter (summed
over the instances rocprofv3 reports, i.e. whole-chip totals).  HBM bytes follow
/opt/skills/guides/MI355X_MICROARCH.md: WRITE_SIZE and FETCH_SIZE are in KiB;
FETCH_SIZE reads half of a wide coalesced stream on gfx950 (documented there), so
reads are reported raw and x2.
"""
import glob
import os
import sqlite3
import sys


def short(name):
  name = name.replace('(anonymous namespace)::', '')
  return name.split('(')[0][:80]


def kernel_stats(db):
  cur = sqlite3.connect(db).cursor()
  rows = cur.execute(
      'select name, count(*), avg(end-start), min(end-start), max(end-start), '
      'max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), '
      'grid_x, max(workgroup_x) from kernels group by name, grid_x '
      'order by sum(end-start) desc').fetchall()
  return rows


def pmc_stats(db):
  cur = sqlite3.connect(db).cursor()
  # value per (dispatch, counter) summed over instances, then averaged over dispatches
  rows = cur.execute(
      'select kernel_name, counter_name, avg(v), count(*) from ('
      ' select kernel_name, counter_name, dispatch_id, sum(value) as v '
      ' from counters_collection group by kernel_name, counter_name, dispatch_id) '
      'group by kernel_name, counter_name').fetchall()
  out = {}
  for k, c, v, n in rows:
    out.setdefault(k, {})[c] = (v, n)
  return out


def main(d):
  trace = os.path.join(d, 'trace_results.db')
  print('# rocprofv3 summary of', d)
  line = os.path.join(d, 'bench_line.json')
  if os.path.exists(line):
    print('# bench line of the traced run:')
    print('#', open(line).read().strip())
  print('\n## kernel trace (rocprofv3 --kernel-trace --stats), durations in us')
  print('%-82s %6s %10s %10s %10s %5s %5s %7s %7s %9s %5s' % (
      'kernel', 'calls', 'avg', 'min', 'max', 'vgpr', 'sgpr', 'lds', 'scratch', 'grid_x', 'wg_x'))
  for r in kernel_stats(trace):
    print('%-82s %6d %10.2f %10.2f %10.2f %5d %5d %7d %7d %9d %5d' % (
        short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], r[6], r[7], r[8], r[9], r[10]))
  for db in sorted(glob.glob(os.path.join(d, 'pmc_*_results.db'))):
    print('\n## PMC pass %s (per-dispatch average, summed over counter instances)' % os.path.basename(db))
    stats = pmc_stats(db)
    for k in sorted(stats, key=lambda k: -max(v for v, _ in stats[k].values())):
      if 'rocclr' in k or 'at::native' in k:
        continue
      print(' ', short(k))
      for c, (v, n) in sorted(stats[k].items()):
        extra = ''
        if c == 'WRITE_SIZE':
          extra = '  -> %.1f MB written per dispatch' % (v * 1024 / 1e6)
        if c == 'FETCH_SIZE':
          extra = '  -> %.1f MB read per dispatch (raw; x2 = %.1f MB if wide coalesced)' % (
              v * 1024 / 1e6, 2 * v * 1024 / 1e6)
        print('      %-22s %18.1f  (n=%d)%s' % (c, v, n, extra))


if __name__ == '__main__':
  main(sys.argv[1])
