#!/usr/bin/env python3
"""Run the driver's bench command N times and tabulate ms_per_step against kernel_ms.

    python tools/driver_repro.py [N] [-- extra bench.py flags]

Each run is a fresh process (`python bench.py --gpus 1 --steps 20 --warmup 5
--no-cpu-baseline --no-extras` + the extra flags: the headline measurement comes first in
a full run too, so what follows it cannot change it).  One row per run: ms_per_step,
kernel_ms, their ratio, the timed window's host-side breakdown (loop / log_wait /
synchronize, us) and when the window's gather could start and was done on the device clock
against the moment the last launch finished.  VERDICT r4 item 2: the driver's round-4 line
read ms_per_step = 1.219 x kernel_ms; this is the script that looks for such a run.
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  argv = sys.argv[1:]
  extra = []
  if '--' in argv:
    k = argv.index('--')
    argv, extra = argv[:k], argv[k + 1:]
  n = int(argv[0]) if argv else 20
  cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '20',
         '--warmup', '5', '--no-cpu-baseline', '--no-extras'] + extra
  print('# ' + ' '.join(cmd[1:]))
  print('# run  ms_per_step  kernel_ms  ratio   loop launches log_wait  sync | launches_done(dev)  gather: ready(dev) | issued+call..seen_done(host), done when the launches ended? (us)')
  worst = 0.0
  for i in range(n):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
      d = json.loads(r.stdout.strip().splitlines()[-1])
    except (IndexError, ValueError):
      print('%3d  rc=%d no line' % (i, r.returncode))
      continue
    extra_rows = [l for l in r.stderr.splitlines() if l.startswith('BENCH_DETAILS ')]
    details = json.loads(extra_rows[-1][len('BENCH_DETAILS '):]) if extra_rows else {}
    w = details.get('window_us') or {'loop': 0.0, 'log_wait': 0.0, 'synchronize': 0.0}
    g = details.get('gathers') or []
    ratio = d['ms_per_step'] / d['roofline']['kernel_ms']
    worst = max(worst, ratio)
    print('%3d  %.4f  %.4f  %.3f  %6.0f %6.0f %6.0f %6.0f | %6.0f  %s' % (
        i, d['ms_per_step'], d['roofline']['kernel_ms'], ratio, w['loop'], w.get('launches', 0.0),
        w['log_wait'], w['synchronize'], w.get('launches_done', 0.0),
        ' '.join('%.0f | %.0f+%.0f..%s %s' % (
            x['ready_us'], x['issued_us'], x['call_us'],
            '-' if x['seen_done_us'] is None else '%.0f' % x['seen_done_us'],
            'yes' if x.get('done_when_launches_ended') else 'NO') for x in g)))
    sys.stdout.flush()
  print('# worst ratio %.3f over %d runs' % (worst, n))


if __name__ == '__main__':
  main()
