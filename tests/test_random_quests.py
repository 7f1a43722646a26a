"""A seeded family of 12 games of ARBITRARY Python classes on random boards (tests/random_quests.py:
ice rinks, toll roads, burrows, vaults - tests/traced_games.py's classes) against what the
REFERENCE's engine, renderer and Plot did with the very same classes
(tests/golden/random_quests.npz, make_random_golden.py quests).  Sliding many cells a frame,
per-frame discounts, an episode end with discount 0.75, a thing that changes its place in the
z-order and is paid for being hidden, things that leave the board, a sprite that shows itself
only sometimes: what reaches the device through the tabulator.

Per game: (a) the generator still makes the fixture's game; (b) this repo's generic tier gives
the reference engine's frames; (c) so does the table tabulated from the classes, walked on the
host; (d, GPU) so does the HIP path, rollout() and play()."""

import json
import os

import numpy as np
import pytest
import torch

from campx_amd import tabulate
from conftest import GOLDEN_DIR
import random_quests

DEFS = random_quests.definitions()
IDS = ['quest{}-{}'.format(k, d['kind']) for k, d in enumerate(DEFS)]


def _gold(k):
  with np.load(os.path.join(GOLDEN_DIR, 'random_quests.npz')) as f:
    pre = 'k{}_'.format(k)
    return {name[len(pre):]: f[name] for name in f.files if name.startswith(pre)}


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f' or b.dtype.kind == 'f':
    return np.array_equal(a.astype(np.float32), b.astype(np.float32), equal_nan=True)
  return np.array_equal(a, b)


def test_the_generator_still_makes_the_games_of_the_fixture():
  assert len(DEFS) == random_quests.N_GAMES == 12
  for k, d in enumerate(DEFS):
    gold = _gold(k)
    assert [''.join(chr(c) for c in row) for row in gold['art']] == d['art'], k
    assert json.loads(str(gold['meta'])) == dict(kind=d['kind']), k
  discounts = set()
  for k in range(len(DEFS)):
    discounts |= set(np.unique(_gold(k)['discount']).tolist())
  assert discounts == {0.0, 0.25, 0.5, 0.75, 1.0}


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_generic_tier_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_quests.builder(DEFS[k])
  onehot = tabulate.default_actions()
  for n in range(N):
    game = build()
    obs, _, _ = game.its_showtime()
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      if game.game_over:
        game = build()
        game.its_showtime()
      obs, reward, discount = game.play(onehot[int(gold['actions'][t, n])])
      assert np.array_equal(obs.board.numpy(), gold['board'][t + 1, n].astype(np.uint8)), (n, t)
      assert np.array_equal(obs.layered_board.numpy(), gold['layered'][t + 1, n]), (n, t)
      assert _same(np.float32(np.nan if reward is None else float(reward)), gold['reward'][t, n]), (n, t)
      assert np.float32(discount) == gold['discount'][t, n], (n, t)
      assert int(game.game_over) == gold['done'][t, n]


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_the_table_tabulated_from_the_classes_gives_them_too(k):
  from oracle.table_replay import StateWalker, TableWalker
  gold = _gold(k)
  T, N = gold['actions'].shape
  traced = tabulate.trace(random_quests.builder(DEFS[k])(), cache=False)
  assert [ord(c) for c in traced.chars] == gold['chars'].tolist()
  if traced.dense_reason is not None:
    walker = StateWalker(traced, N)
    want = walker.rollout(gold['actions'], reset_first=True)
    render = lambda t: walker.render(want['state'][t])
  else:
    walker = TableWalker(traced, N)
    want = walker.rollout(gold['actions'], reset_first=True)
    render = lambda t: walker.render(want['cells'][:, t].astype(np.int64))
  for name in ('reward', 'discount', 'done'):
    assert _same(want[name], gold[name]), name
  for t in range(T):
    board, layered = render(t)
    assert np.array_equal(board, gold['board'][t + 1]), t
    assert np.array_equal(layered, gold['layered'][t + 1].astype(np.int8)), t


@pytest.mark.gpu
@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_hip_path_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_quests.builder(DEFS[k])
  game = build(batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert game.fused is not None and game.fused.traced is not None
  assert np.array_equal(first.board.cpu().numpy(), gold['board'][0])
  assert np.array_equal(first.layered_board.cpu().numpy(), gold['layered'][0].astype(np.int8))
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert np.array_equal(out['obs'].cpu().numpy(), gold['layered'][1:].astype(np.int8))
  assert np.array_equal(out['board'].cpu().numpy(), gold['board'][1:])
  for name in ('reward', 'discount', 'done'):
    assert _same(out[name].cpu().numpy(), gold[name]), name
  game = build(batch=N, device='cuda')
  game.its_showtime()
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert np.array_equal(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert np.array_equal(obs.layered_board.cpu().numpy(), gold['layered'][t + 1].astype(np.int8)), t
    assert _same(reward.cpu().numpy(), gold['reward'][t]), t
    assert _same(discount.cpu().numpy(), gold['discount'][t]), t
    assert np.array_equal(game.fused.done.cpu().numpy(), gold['done'][t])
