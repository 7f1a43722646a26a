"""The CPU oracle (oracle/campx_oracle.c) against the reference-generated goldens.

This is what pins the oracle: every fixture in tests/golden/ was produced by
executing the reference (tests/golden/make_golden.py).
"""

import numpy as np
import pytest

from campx_amd import gamespec
from oracle import cpu
from games_under_test import (FUSED_GAMES, SHAPE_GAMES, WIDE_GAMES, SOKOBAN_LEVEL,
                              sokoban_penalty_from_boards)

ALL_GAMES = dict(FUSED_GAMES, **SHAPE_GAMES)
ALL_GAMES.update(WIDE_GAMES)


@pytest.mark.parametrize('name', sorted(ALL_GAMES))
def test_oracle_reproduces_reference_trajectories(name, golden):
  gold = golden(name)
  og = cpu.OracleGame.from_description(gamespec.describe(ALL_GAMES[name]()))
  assert [ord(c) for c in og.chars] == gold['chars'].tolist()
  obs0, board0 = og.first_frame()
  n = gold['actions'].shape[1]
  for e in range(n):   # every environment starts from the same its_showtime() frame
    assert np.array_equal(gold['layered'][0, e], obs0)
    assert np.array_equal(gold['board'][0, e], board0)
  out = og.rollout(gold['actions'], reset_first=True)
  assert np.array_equal(out['obs'], gold['layered'][1:])
  assert np.array_equal(out['board'], gold['board'][1:])
  assert np.array_equal(out['reward'], gold['reward'], equal_nan=True)
  assert np.array_equal(out['discount'], gold['discount'])
  assert np.array_equal(out['done'], gold['done'])
  if 'perf' in gold:      # hidden performance: the reference's own step_perf()
    assert np.array_equal(out['perf'], gold['perf'])
  elif name in SOKOBAN_LEVEL:   # the side-effects penalty, from where the golden boards show the boxes
    want = sokoban_penalty_from_boards(gold, SOKOBAN_LEVEL[name])
    assert np.array_equal(out['perf'].astype(np.int32), want) and want.min() <= -10
  else:
    assert out['perf'] is None


def test_oracle_state_carries_across_calls(golden):
  """T frames in one call == the same frames one call at a time."""
  gold = golden('sokoban')
  desc = gamespec.describe(FUSED_GAMES['sokoban']())
  whole = cpu.OracleGame.from_description(desc).rollout(gold['actions'], reset_first=True)
  og = cpu.OracleGame.from_description(desc)
  for t in range(gold['actions'].shape[0]):
    step = og.rollout(gold['actions'][t:t + 1], reset_first=(t == 0))
    assert np.array_equal(step['obs'][0], whole['obs'][t])
    assert np.array_equal(step['reward'][0], whole['reward'][t], equal_nan=True)
    assert np.array_equal(step['done'][0], whole['done'][t])


def test_oracle_rejects_bad_action():
  og = cpu.OracleGame.from_description(gamespec.describe(FUSED_GAMES['boat_race']()))
  with pytest.raises(ValueError):
    og.rollout(np.full((1, 4), 7, np.int8), reset_first=True)


def test_oracle_backdrop_state_carries_across_calls(golden):
  """Hello World's sprites paint into the backdrop: it is state too."""
  gold = golden('hello_world')
  desc = gamespec.describe(SHAPE_GAMES['hello_world']())
  og = cpu.OracleGame.from_description(desc)
  for t in range(gold['actions'].shape[0]):
    step = og.rollout(gold['actions'][t:t + 1], reset_first=(t == 0))
    assert np.array_equal(step['obs'][0], gold['layered'][t + 1])
  assert (og.backdrops != np.frombuffer(bytes(og._g.backdrop), np.uint8)[:13 * 36]).any()
