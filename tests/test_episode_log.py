"""CSV episode log against a hand computation from the boat-race golden."""

import csv
import io

import numpy as np
import torch

from campx_amd.episode_log import EpisodeCsvLog, FIELDNAMES


def test_header_is_the_references():
  # examples/reinforce.py:281
  assert FIELDNAMES == ['id', 'step', 't(s)', 'ep', 'L', 'R', 'R_av_5', 'P', 'P_av']


def test_rows_from_the_boat_race_golden(golden):
  gold = golden('boat_race')
  reward, perf = gold['reward'], gold['perf']          # [T, N]
  T, N = reward.shape
  # cut the golden stream into 4 "episodes" of T//4 frames, environments = batch
  n = T // 4
  buf = io.StringIO()
  ticks = iter(np.arange(0.0, 100.0, 1.5))
  log = EpisodeCsvLog(buf, run_id=7, frames_per_episode=n, clock=lambda: next(ticks))
  want = []
  rs, ps = [], []
  for e in range(4):
    r = reward[e * n:(e + 1) * n].sum(0)
    p = perf[e * n:(e + 1) * n].astype(np.int64).sum(0)
    log.episode(torch.from_numpy(r), torch.from_numpy(p), loss=0.125 * e)
    rs.append(float(r.astype(np.float32).mean()))
    ps.append(float(p.astype(np.float32).mean()))
    want.append([7, (e + 1) * n, 1.5, e, round(0.125 * e, 2), rs[-1], sum(rs[-5:]) / len(rs[-5:]),
                 ps[-1], sum(ps) / len(ps)])
  rows = list(csv.reader(io.StringIO(buf.getvalue())))
  assert rows[0] == FIELDNAMES and len(rows) == 5
  for got, exp in zip(rows[1:], want):
    assert [int(got[0]), int(got[1]), float(got[2]), int(got[3])] == exp[:4]
    assert np.allclose([float(x) for x in got[4:]], exp[4:], rtol=1e-6, atol=1e-6)
  # the scripted lap of boat_race.py:154-184 is in the golden: P is not identically 0
  assert any(abs(p) > 0 for p in ps)


def test_block_of_gathered_returns_and_missing_columns():
  buf = io.StringIO()
  log = EpisodeCsvLog(buf, frames_per_episode=100, clock=lambda: 0.0)
  gathered = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)   # [world, E, B]
  rows = log.block(gathered)
  assert [r[3] for r in rows] == [0, 1, 2] and [r[1] for r in rows] == [100, 200, 300]
  assert rows[1][5] == float(gathered[:, 1].mean())
  assert rows[2][6] == sum(float(gathered[:, e].mean()) for e in range(3)) / 3
  assert rows[0][4] == '' and rows[0][7] == '' and rows[0][8] == ''
