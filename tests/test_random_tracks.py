"""A seeded family of 40 race-track games (tests/random_tracks.py; the last 8 on boards of 140 to 384
cells, which the batched engine runs from their state table - the wide tier) against what the REFERENCE's own
classes did on the reference's engine (tests/golden/random_tracks.npz, make_random_golden.py):
random boards, tiles, reward vectors, z-orders (tiles in front of the agent or behind it), update
schedules of one to three groups, tiles that block.  Outside the hand-made example games, this is
where `gamespec.describe()` / `lower()` - which the oracle and the kernels both consume - and the
tabulator meet frames that neither of them produced.

Per game: (a) the generator still produces the game the fixture's frames belong to; (b) this
repo's generic tier (batch=None) gives the reference's frames; (c) so does the rule lowering run by
the C oracle, where `lower()` accepts the game (not with a tile painted in front of the agent), and
(d) the table tabulated from the classes - bound afresh, arbitrary Python to the engine - walked by
oracle/table_replay.py, for every game; (e, GPU) so does the HIP path, rollout and play() frame by
frame, through the route the engine picks for the library classes and through the tabulator."""

import json
import os

import numpy as np
import pytest
import torch

from campx_amd import gamespec, tabulate
from conftest import GOLDEN_DIR
from oracle import cpu
import random_tracks

DEFS = random_tracks.definitions()
IDS = ['track{}'.format(k) for k in range(len(DEFS))]


def _gold(k):
  with np.load(os.path.join(GOLDEN_DIR, 'random_tracks.npz')) as f:
    pre = 'k{}_'.format(k)
    return {name[len(pre):]: f[name] for name in f.files if name.startswith(pre)}


def _same(a, b):
  a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
  return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)


def test_the_generator_still_makes_the_games_of_the_fixture():
  assert len(DEFS) == random_tracks.N_GAMES + random_tracks.N_BIG == 40
  for k, d in enumerate(DEFS):
    gold = _gold(k)
    assert [''.join(chr(c) for c in row) for row in gold['art']] == d['art'], k
    meta = json.loads(str(gold['meta']))
    assert meta == dict(tiles=d['tiles'], dctns=d['dctns'], z_order=d['z_order'], schedule=d['schedule'],
                        blocking=d['blocking']), k
  # the family covers what it is for
  assert sum(len(d['schedule']) > 1 for d in DEFS) >= 10
  assert sum(len(d['blocking']) > 1 for d in DEFS) >= 8
  assert sum(any(d['z_order'].index(c) > d['z_order'].index('A') for c in d['tiles']) for d in DEFS) >= 12


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_generic_tier_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_tracks.library_builder(DEFS[k])
  onehot = tabulate.default_actions()
  for n in range(N):
    game = build()
    obs, reward, discount = game.its_showtime()
    assert reward is None and discount == 1.0
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      obs, reward, discount = game.play(onehot[int(gold['actions'][t, n])])
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
    desc = gamespec.describe(random_tracks.library_builder(d)())
    gamespec.lower(desc)
    return desc
  except ValueError:                     # ('fused tier: ...': not a game of the rule lowering)
    return None


def test_both_lowerings_are_exercised():
  took = [_lowers(d) is not None for d in DEFS]
  assert sum(took) >= 8 and len(took) - sum(took) >= 8, took


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
  traced = tabulate.trace(random_tracks.library_builder(DEFS[k], rebound=True)(), cache=False)
  assert tabulate.LAST_WALK[0].startswith('lanes: '), tabulate.LAST_WALK[0]
  assert [ord(c) for c in traced.chars] == gold['chars'].tolist()
  assert (traced.dense_reason is not None) == (k >= random_tracks.N_GAMES)       # the big boards
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
  build = random_tracks.library_builder(DEFS[k], rebound=rebound)
  game = build(batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert game.fused is not None
  assert (game.fused.traced is not None) == (rebound or _lowers(DEFS[k]) is None)
  assert (type(game.fused).__name__ == 'WideGame') == (k >= random_tracks.N_GAMES)     # the big boards
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
def test_a_large_batch_of_every_track_against_the_lowering_on_the_host():
  """B = 4 096 random action streams per game: the HIP path against the C oracle (rule lowering)
  or the table walker (tabulated games) - whichever the fixture pinned above."""
  B, T = 4096, 40
  for k, d in enumerate(DEFS):
    build = random_tracks.library_builder(d)
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
