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


def test_own_game_example_runs_on_the_generic_tier():
  """examples/own_game_batched.py: its classes are ordinary CampX classes - one environment,
  no GPU: walk to the key, through the door it opens, to the gem."""
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import torch
  import own_game_batched as ex
  game = ex.make_game()
  obs, reward, discount = game.its_showtime()
  assert reward is None and chr(int(obs.board[1, 1])) == 'A' and chr(int(obs.board[1, 14])) == ' '
  total = 0.0
  for a in [1] * 5 + [3] * 5 + [1] * 3 + [2] * 5 + [1] * 5:
    obs, reward, discount = game.play(torch.eye(5)[a])
    total += float(reward)
  assert game.game_over and discount == 0.0
  assert total == -0.25 * 23 + 1.0 + 0.5 + 10.0


@pytest.mark.gpu
def test_own_game_example_runs_batched():
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import own_game_batched as ex
  from campx_amd import wide
  got = ex.run(batch=2048, frames=60, launches=2)
  f = got['game'].fused
  assert isinstance(f, wide.WideGame) and f.traced.movers == ['A', 'k', 'D', '$']
  assert got['rate'] > 1e6 and got['out']['obs'].shape == (60, 2048, 6, 12, 16)


def test_hello_world_example_classes_are_recognised_on_the_cpu():
  """examples/hello_world_batched.py: its make_game() (the notebook's classes as the notebook
  types them) is recognised to the committed Hello World spec - the part that needs no GPU."""
  import ctypes
  import numpy as np
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import hello_world_batched as ex
  from campx_amd import gamespec, recognise
  game = ex.make_game()
  assert game.batch is None and not gamespec.is_rule_game(game)
  spec = gamespec.lower_shapes(recognise.shapes(game))
  with np.load(os.path.join(REPO, 'tests', 'golden', 'hello_world_spec.npz')) as f:
    assert ctypes.string_at(ctypes.addressof(spec), ctypes.sizeof(spec)) == f['spec'].tobytes()


@pytest.mark.gpu
def test_hello_world_example_runs_batched(capsys):
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import hello_world_batched as ex
  from campx_amd import engine
  try:
    ex.main()
  finally:
    engine.set_default_batch(None)
  out = capsys.readouterr().out
  assert 'ShapeGame' in out and 'for every environment: True' in out and 'TB/s' in out


@pytest.mark.gpu
def test_deferred_rollouts_example_runs():
  """examples/random_rollouts_deferred.py: every episode's observations reach the consumer, in
  order, complete (a 5x5 boat race frame shows 25 cells: one 1 per cell over the 7 layers)."""
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import random_rollouts_deferred as ex
  got = ex.run(batch=1024, frames=40, episodes=5)
  assert got['seen'] == [40 * 1024 * 25] * 5
  assert -3.0 * 40 <= got['mean_return'] <= 2.0 * 40


def test_coins_example_runs_on_the_generic_tier_and_tabulates():
  """examples/coins_batched.py: ordinary CampX classes - a drape of three coins, a Backdrop that
  repaints the whole floor - on one environment without a GPU, and through the tabulator: the floors
  and the coins that are left are variants of the scenery; without the switch the coins are pieces in
  a mask."""
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import torch
  import coins_batched as ex
  from campx_amd import tabulate
  game = ex.make_game()
  obs, reward, _ = game.its_showtime()
  assert reward is None
  row = lambda r: ''.join(chr(int(c)) for c in obs.board[r])
  assert row(1) == '#A  o   .#' and row(3) == '#o  s # o#'
  total = 0.0
  for a in [1, 1, 1, 3, 3, 0, 0]:          # to the coin, down onto the switch, away: the floor has turned
    obs, reward, _ = game.play(torch.eye(5)[a])
    total += float(reward)
  assert row(1) == '#.......' + ' #' and row(3)[4] == 's' and row(3)[2] == 'A'
  assert total == 7 * -0.125 + 1.0 + 2 * 0.25          # a coin, two frames of night
  traced = tabulate.trace(ex.make_game())
  # one tracked thing, the walker; the scenery - two floors x the coins that are left - in variants
  assert traced.movers == ['A'] and 2 < len(traced.variants) <= 16
  assert all(sorted(m) == ['o'] for m in traced.variant_masks)
  assert traced.variant_masks[0]['o'].reshape(-1).nonzero()[0].tolist() == [14, 31, 38]
  assert len({v.tobytes() for v in traced.variants}) == 2               # day and night
  # the coin field alone (what bench.py's `coin_field` row runs): three pieces beside the walker
  field = tabulate.trace(ex.make_game(floor=False))
  assert field.movers == ['A', 'o', 'o', 'o'] and field.piece_cell == [None, 14, 31, 38] and field.pieces_as_mask
  spec, arrays = tabulate.to_wide_spec(field)
  assert (spec.n_dyn, spec.n_pieces, spec.n_variants, spec.n_layers) == (1, 3, 0, 5)
  # (every subset of the coins is reached but one: the way to the third leads over one of the others)
  assert sorted(set(arrays['state_pieces'].tolist())) == [0, 1, 2, 4, 5, 6, 7]


@pytest.mark.gpu
def test_coins_example_runs_batched():
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import coins_batched as ex
  from campx_amd import wide
  got = ex.run(batch=2048, frames=60, launches=2)
  f = got['game'].fused
  assert isinstance(f, wide.WideGame) and 2 < got['variants'] <= 16 and got['movers'] == ['A']
  assert f.spec.n_variants == got['variants'] and got['rate'] > 1e6 and got['out']['obs'].shape == (60, 2048, 7, 6, 10)
  # ... and bench.py's `coin_field` row builds the game it says it does
  sys.path.insert(0, REPO)
  import bench
  game = bench.build_game('coin_field', batch=512, device='cuda')
  game.its_showtime()
  assert isinstance(game.fused, wide.WideGame) and (game.fused.spec.n_dyn, game.fused.spec.n_pieces) == (1, 3)
  row = game.fused.n_layers * game.fused.rows * game.fused.cols
  assert bench.BYTES_PER_ENV_STEP['coin_field'] == row + 4 + 1 + 2 * game.fused._n_planes + 1
