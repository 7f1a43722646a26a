#!/usr/bin/env python3
"""profiles/<round>_traffic.json from the rocprofv3 PMC passes of tools/profile.sh.

    python tools/make_traffic.py [--round r04] [--out FILE] gpurun_out/prof_r04_boat_race:boat_race:65536:100 ...

(runs on the GPU box right after the passes - tools/gpu_profile_all.sh - because the rocpd
databases are too large to travel back; the json and the text summaries do)

Per configuration: WRITE_SIZE and FETCH_SIZE (KiB, separate passes) of the kernels of one
rollout launch, per dispatch (median over the run's dispatches), summed.  WRITE_SIZE is taken at face value (it reads
1146.9 MB for the render kernel's exactly 1146.88 MB of stores).  FETCH_SIZE follows
MI355X_MICROARCH.md: it reports half the bytes of a wide (16 B/lane) coalesced stream,
which is what the update kernels' action loads are (3.34 MB reported for 6.55 MB of
actions), so their reading is doubled; the render kernel's loads are 1 B/lane trace bytes
and 16-byte L1-resident table reads, taken raw.
"""
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(db, counter):
  """kernel -> (bytes per dispatch, dispatches): the MEDIAN over its dispatches (a run also
  holds one-off dispatches of the same kernel - the one-frame render of its_showtime() - that
  an average would mix in)."""
  cur = sqlite3.connect(db).cursor()
  rows = cur.execute(
      'select kernel_name, dispatch_id, sum(value) from counters_collection '
      'where counter_name = ? group by kernel_name, dispatch_id', (counter,)).fetchall()
  per = {}
  for k, _, v in rows:
    per.setdefault(k, []).append(v)
  return {k: (sorted(v)[len(v) // 2] * 1024.0, len(v)) for k, v in per.items()}


def totals(db, counter):
  """kernel -> (bytes over ALL its dispatches of the run, dispatches)."""
  cur = sqlite3.connect(db).cursor()
  rows = cur.execute('select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection '
                     'where counter_name = ? group by kernel_name', (counter,)).fetchall()
  return {k: (v * 1024.0, n) for k, n, v in rows}


def launches_of(log):
  """Rollout launches of the bench run whose output is in `log`: warm-up + settle + the timed
  steps + the untimed per-launch pass (bench.py measure_rollout)."""
  for raw in reversed(open(log).read().splitlines()):
    if raw.startswith('{'):
      line = json.loads(raw)
      return line['warmup'] + line['config']['settle_launches'] + 2 * line['steps']
  raise ValueError('no bench line in ' + log)


def shape_entry(d, game, batch, frames, rnd):
  """The shape tier runs a launch as several chunks of update pass + render (+ the backdrop /
  trail-word conversions at its ends): per LAUNCH = the kernels' bytes over the whole run / the
  run's launches (each pass prints its own bench line: the settle count differs under counters)."""
  out = {}
  for name, counter, log in (('write_bytes', 'WRITE_SIZE', 'pmc_write'), ('fetch_bytes', 'FETCH_SIZE', 'pmc_fetch')):
    per = totals(os.path.join(d, log + '_results.db'), counter)
    n = launches_of(os.path.join(d, log + '.log'))
    names = [k for k in per if 'shape_' in k]
    # (FETCH_SIZE doubled for the update pass's wide action loads, as for the one-cell tier)
    out[name] = sum(per[k][0] * (2.0 if counter == 'FETCH_SIZE' and 'shape_update' in k else 1.0)
                    for k in names) / n
    out['kernels'] = ' + '.join(sorted({k.replace('(anonymous namespace)::', '').replace('campx_impl::', '').split('(')[0].split('<')[0]
                                        for k in names}))
    out[name.replace('bytes', 'launches')] = n
  out['traffic_bytes'] = out['write_bytes'] + out['fetch_bytes']
  out['source'] = 'profiles/{}_{}_rocprofv3.txt'.format(rnd, game)
  return out


def main(specs):
  rnd = 'r04'
  if specs and specs[0] == '--round':
    rnd, specs = specs[1], specs[2:]
  path = os.path.join(REPO, 'profiles', rnd + '_traffic.json')
  if specs and specs[0] == '--out':
    path, specs = specs[1], specs[2:]
  table = {}
  if os.path.exists(path):          # configurations not named on the command line are kept
    with open(path) as f:
      table = json.load(f)
  table['_comment'] = __doc__.split('\n\n', 2)[2].strip().replace('\n', ' ')
  for spec in specs:
    d, game, batch, frames = spec.split(':')
    if game.startswith('hello') or game.startswith('shape'):
      table['{}:{}:{}:split'.format(game, batch, frames)] = shape_entry(d, game, batch, frames, rnd)
      continue
    w = per_dispatch(os.path.join(d, 'pmc_write_results.db'), 'WRITE_SIZE')
    f = per_dispatch(os.path.join(d, 'pmc_fetch_results.db'), 'FETCH_SIZE')
    names = [k for k in w if 'render_kernel' in k or 'update_' in k or 'rollout_kernel' in k]
    # (kernels of the rollout launches only: not the one-off board render of its_showtime())
    most = max([w[k][1] for k in names] or [0])
    names = [k for k in names if 2 * w[k][1] >= most]
    w = {k: v[0] for k, v in w.items()}
    f = {k: v[0] for k, v in f.items()}
    short = [k.replace('(anonymous namespace)::', '').replace('campx_impl::', '').split('(')[0]
             for k in names]
    wb = sum(w[k] for k in names)
    fb = sum(f.get(k, 0.0) * (2.0 if 'update_' in k else 1.0) for k in names)
    table['{}:{}:{}:split'.format(game, batch, frames)] = {
        'kernels': ' + '.join(sorted(short)), 'write_bytes': wb, 'fetch_bytes': fb,
        'traffic_bytes': wb + fb,
        'source': 'profiles/{}_{}_rocprofv3.txt'.format(rnd, game)}
  with open(path, 'w') as out:
    json.dump(table, out, indent=1)
  print(json.dumps(table, indent=1))


if __name__ == '__main__':
  main(sys.argv[1:])
