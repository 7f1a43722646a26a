"""A seeded family of 16 coin-field games (tests/random_coins.py) against what Demo 3's own AgentDrape
(examples/Demo 3: Hover Reward Example.ipynb cell 3, exec'd from the notebook) did on the reference's
engine (tests/golden/random_coins.npz, make_random_golden.py coins): an agent rewarded when it
ENTERS a cell showing a reward tile - through the rendered, occluded layers kept in the Plot -, random
boards, one or two kinds of tiles, z-orders with tiles in front of the agent or behind it, random
update schedules, a tile kind that blocks.  The same five legs as tests/test_random_tracks.py:
generator, generic tier, rule lowering (C oracle), tabulated table, HIP path through both routes."""

import json
import os

import numpy as np
import pytest
import torch

from campx_amd import gamespec, tabulate
from conftest import GOLDEN_DIR
from oracle import cpu
import random_coins

DEFS = random_coins.definitions()
IDS = ['coins{}'.format(k) for k in range(len(DEFS))]


def _gold(k):
  with np.load(os.path.join(GOLDEN_DIR, 'random_coins.npz')) as f:
    pre = 'k{}_'.format(k)
    return {name[len(pre):]: f[name] for name in f.files if name.startswith(pre)}


def _same(a, b):
  a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
  return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)


def test_the_generator_still_makes_the_games_of_the_fixture():
  assert len(DEFS) == random_coins.N_GAMES == 16
  for k, d in enumerate(DEFS):
    gold = _gold(k)
    assert [''.join(chr(c) for c in row) for row in gold['art']] == d['art'], k
    meta = json.loads(str(gold['meta']))
    assert meta == dict(tiles=d['tiles'], z_order=d['z_order'], schedule=d['schedule'],
                        blocking=d['blocking'], rewarding=d['rewarding']), k
  # the family covers what it is for
  assert sum(len(d['blocking']) > 1 for d in DEFS) >= 3
  assert sum(any(d['z_order'].index(c) > d['z_order'].index('A') for c in d['tiles']) for d in DEFS) >= 8


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_generic_tier_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_coins.library_builder(DEFS[k])
  onehot = tabulate.default_actions()
  for n in range(N):
    game = build()
    obs, reward, discount = game.its_showtime()
    assert reward is None and discount == 1.0
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      obs, reward, discount = game.play([int(i == int(gold['actions'][t, n])) for i in range(5)])     # (a plain list, as the notebook passes)
      assert np.array_equal(obs.board.numpy(), gold['board'][t + 1, n].astype(np.uint8)), (n, t)
      assert np.array_equal(obs.layered_board.numpy(), gold['layered'][t + 1, n]), (n, t)
      assert _same(np.nan if reward is None else float(reward), gold['reward'][t, n]), (n, t)
      assert np.float32(discount) == gold['discount'][t, n] and not game.game_over


def _walk_table(traced, actions):
  """The tabulated game walked on the host: (reward, discount, done, render(t)) - through the
  cell-indexed tables, or through the state table where the game runs from that (the wide tier)."""
  from oracle.table_replay import StateWalker, TableWalker
  B = actions.shape[1]
  if traced.dense_reason is not None:
    walker = StateWalker(traced, B)
    want = walker.rollout(actions, reset_first=True)
    return want, lambda t: walker.render(want['state'][t])
  walker = TableWalker(traced, B)
  want = walker.rollout(actions, reset_first=True)
  return want, lambda t: walker.render(want['cells'][:, t].astype(np.int64))


def _lowers(d):
  try:
    desc = gamespec.describe(random_coins.library_builder(d)())
    gamespec.lower(desc)
    return desc
  except ValueError:                     # ('fused tier: ...': not a game of the rule lowering)
    return None


def test_both_lowerings_are_exercised():
  took = [_lowers(d) is not None for d in DEFS]
  assert sum(took) >= 3 and len(took) - sum(took) >= 3, took


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_rule_lowering_run_by_the_c_oracle_gives_them_too(k):
  gold = _gold(k)
  desc = _lowers(DEFS[k])
  if desc is None:
    pytest.skip('a tile in front of the agent: the batched engine tabulates this game (next test)')
  og = cpu.OracleGame.from_description(desc)
  assert [ord(c) for c in og.chars] == gold['chars'].tolist()
  obs0, board0 = og.first_frame()
  assert np.array_equal(gold['layered'][0, 0], obs0) and np.array_equal(gold['board'][0, 0], board0)
  out = og.rollout(gold['actions'], reset_first=True)
  assert np.array_equal(out['obs'], gold['layered'][1:])
  assert np.array_equal(out['board'], gold['board'][1:])
  assert _same(out['reward'], gold['reward']) and np.array_equal(out['discount'], gold['discount'])
  assert np.array_equal(out['done'], gold['done'])


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_the_table_tabulated_from_the_classes_gives_them_too(k):
  """Every game, its classes bound afresh - arbitrary Python classes to the engine, as a user's
  own are: tabulated (on lanes, many states per call), the table walked on the host."""
  gold = _gold(k)
  T, N = gold['actions'].shape
  traced = tabulate.trace(random_coins.library_builder(DEFS[k], rebound=True)(), cache=False)
  # (coins11: the agent is walled in by what blocks it - nothing ever moves; the lane walker hands
  # such a game to the one-frame walk, which tracks the agent's one cell all the same)
  assert tabulate.LAST_WALK[0].startswith('lanes: ') or 'nothing moves' in tabulate.LAST_WALK[0], tabulate.LAST_WALK[0]
  assert [ord(c) for c in traced.chars] == gold['chars'].tolist()
  assert traced.dense_reason is None
  want, render = _walk_table(traced, gold['actions'])
  for name in ('reward', 'discount', 'done'):
    assert _same(want[name], gold[name]), name
  for t in range(T):
    board, layered = render(t)
    assert np.array_equal(board, gold['board'][t + 1]), t
    assert np.array_equal(layered, gold['layered'][t + 1].astype(np.int8)), t


@pytest.mark.gpu
@pytest.mark.parametrize('rebound', [False, True], ids=['library', 'rebound'])
@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_hip_path_gives_the_reference_engines_frames(k, rebound):
  """`library`: the rule classes as the engine knows them (the rule lowering where it takes the
  game, the tabulator where it does not); `rebound`: as arbitrary classes (always tabulated)."""
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_coins.library_builder(DEFS[k], rebound=rebound)
  game = build(batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert game.fused is not None
  assert (game.fused.traced is not None) == (rebound or _lowers(DEFS[k]) is None)
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
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


@pytest.mark.gpu
def test_a_large_batch_of_every_game_against_the_lowering_on_the_host():
  """B = 4 096 random action streams per game: the HIP path against the C oracle (rule lowering)
  or the table walker (tabulated games) - whichever the fixture pinned above."""
  B, T = 4096, 40
  for k, d in enumerate(DEFS):
    build = random_coins.library_builder(d)
    actions = np.random.RandomState(900 + k).randint(0, 5, size=(T, B)).astype(np.int8)
    game = build(batch=B, device='cuda')
    game.its_showtime()
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    desc = _lowers(d)
    if desc is not None:
      ref = cpu.OracleGame.from_description(desc).rollout(actions, reset_first=True)
      assert np.array_equal(out['obs'].cpu().numpy(), ref['obs']), k
      assert np.array_equal(out['board'].cpu().numpy(), ref['board']), k
      assert _same(out['reward'].cpu().numpy(), ref['reward']), k
    else:
      traced = tabulate.trace(build(), cache=False)
      want, render = _walk_table(traced, actions)
      assert _same(out['reward'].cpu().numpy(), want['reward']), k
      for t in (0, T // 2, T - 1):
        board, layered = render(t)
        assert np.array_equal(out['board'][t].cpu().numpy(), board), (k, t)
        assert np.array_equal(out['obs'][t].cpu().numpy(), layered), (k, t)
