#!/usr/bin/env python3
"""Where a stream's time goes BETWEEN kernels: from a rocprofv3 kernel-trace database, the
kernels in start order with the idle gap in front of each, summed per kernel name, and the busy /
idle split of the last N dispatches (the timed window of a bench run is its tail).

    python tools/kernel_gaps.py trace_results.db [last_n]
"""
import sqlite3
import sys


def main(db, last_n=None):
  cur = sqlite3.connect(db).cursor()
  rows = cur.execute('select name, start, end from kernels order by start').fetchall()
  rows = [(n.replace('(anonymous namespace)::', '').replace('campx_impl::', '').split('(')[0][:60], s, e)
          for n, s, e in rows]
  if last_n:
    rows = rows[-int(last_n):]
  per = {}
  busy = idle = 0
  prev_end = None
  for name, s, e in rows:
    gap = 0 if prev_end is None else max(0, s - prev_end)
    d = per.setdefault(name, [0, 0, 0])
    d[0] += 1
    d[1] += e - s
    d[2] += gap
    busy += e - s
    idle += gap
    prev_end = e if prev_end is None else max(prev_end, e)
  print('%-62s %7s %12s %12s %10s' % ('kernel', 'calls', 'busy us', 'gap-before us', 'gap/call'))
  for name, (n, b, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print('%-62s %7d %12.1f %12.1f %10.2f' % (name, n, b / 1e3, g / 1e3, g / 1e3 / n))
  print('total busy %.1f us, idle between kernels %.1f us (%.2f %%)' % (busy / 1e3, idle / 1e3, 100.0 * idle / max(1, busy + idle)))


if __name__ == '__main__':
  main(*sys.argv[1:])
