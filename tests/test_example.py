"""Smoke test of examples/reinforce_batched.py: the whole RL loop composes on the device
(bf16 observations from play() -> policy -> on-device sampling -> play() -> CSV log)."""

import csv
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_batched_reinforce_example_runs(tmp_path):
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import reinforce_batched
  path = str(tmp_path / 'log.csv')
  history = reinforce_batched.run(batch=512, episodes=3, frames=20, csv=path)
  assert len(history) == 3
  rows = list(csv.reader(open(path)))
  assert rows[0] == ['id', 'step', 't(s)', 'ep', 'L', 'R', 'R_av_5', 'P', 'P_av']
  assert len(rows) == 4 and [r[1] for r in rows[1:]] == ['20', '40', '60']
  # 20 frames at -1 .. +2 per frame
  assert all(-20.0 <= float(r[5]) <= 40.0 for r in rows[1:])
