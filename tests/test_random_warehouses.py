"""A seeded family of 16 sokoban levels (tests/random_warehouses.py: one to three boxes on random
boards) against what the reference's engine, renderer and Plot did with them
(tests/golden/random_warehouses.npz, make_random_golden.py warehouses: the reference's own
AgentDrape as the agent, this repo's Box / Goal rules bound to the reference's `things`).  Games
of two to four movers: the pair / tuple tables the device builds from the rules, and the
multi-mover kernels - update_pair / update_tuple, and since round 5 the one launch that holds
both passes (pipe_multi_kernel) - held to frames neither the oracle nor a kernel produced.

Per level: (a) the generator still makes the fixture's level; (b) the generic tier gives the
reference engine's frames; (c) so does the rule lowering run by the C oracle; (d, GPU) so does the
HIP path, rollout() and play(); (e, GPU) at B = 4 096 - one-launch rollouts - and B = 12 288 - two
launches - the HIP path equals the oracle, and deferred rollouts equal in-order ones."""

import json
import os

import numpy as np
import pytest
import torch

from campx_amd import gamespec, tabulate
from conftest import GOLDEN_DIR
from oracle import cpu
import random_warehouses

DEFS = random_warehouses.definitions()
IDS = ['warehouse{}'.format(k) for k in range(len(DEFS))]


def _gold(k):
  with np.load(os.path.join(GOLDEN_DIR, 'random_warehouses.npz')) as f:
    pre = 'k{}_'.format(k)
    return {name[len(pre):]: f[name] for name in f.files if name.startswith(pre)}


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f' or b.dtype.kind == 'f':
    return np.array_equal(a.astype(np.float32), b.astype(np.float32), equal_nan=True)
  return np.array_equal(a, b)


def test_the_generator_still_makes_the_levels_of_the_fixture():
  assert len(DEFS) == random_warehouses.N_GAMES == 16
  for k, d in enumerate(DEFS):
    gold = _gold(k)
    assert [''.join(chr(c) for c in row) for row in gold['art']] == d['art'], k
    assert json.loads(str(gold['meta'])) == dict(boxes=d['boxes'], z_order=d['z_order'],
                                                 schedule=d['schedule']), k
  assert {len(d['boxes']) for d in DEFS} == {1, 2, 3}


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_generic_tier_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_warehouses.library_builder(DEFS[k])
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
      assert np.float32(discount) == gold['discount'][t, n]
      assert int(game.game_over) == gold['done'][t, n]


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_rule_lowering_run_by_the_c_oracle_gives_them_too(k):
  gold = _gold(k)
  og = cpu.OracleGame.from_description(gamespec.describe(random_warehouses.library_builder(DEFS[k])()))
  assert [ord(c) for c in og.chars] == gold['chars'].tolist()
  out = og.rollout(gold['actions'], reset_first=True)
  assert np.array_equal(out['obs'], gold['layered'][1:])
  assert np.array_equal(out['board'], gold['board'][1:])
  for name in ('reward', 'discount', 'done'):
    assert _same(out[name], gold[name]), name


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_the_rule_classes_bound_afresh_are_tabulated_to_the_same_frames(k):
  """Every level with its classes bound afresh - arbitrary Python classes to the engine, Python
  branches on what differs from state to state and all (`if pushed and not blocked:`) - tabulated
  MANY STATES PER CALL: a frame whose branches disagree between states is run again for each
  group of states that agree (round 5, late; before, such classes went one frame of Python per
  state and action, and levels past 60 000 frames - two boxes on 8x7: 19 646 states - were refused).
  Seven states to 879 570 (three boxes on 8x8, 26 s); the table walked on the host gives the
  reference engine's frames once more, through a route that shares nothing with the rule lowering."""
  from campx import things
  from campx.ascii_art import ascii_art_to_game, Partial
  from campx_amd import rules
  from oracle.table_replay import StateWalker, TableWalker
  R = rules.bind(things)
  gold = _gold(k)
  T, N = gold['actions'].shape
  game = random_warehouses.build(DEFS[k], ascii_art_to_game, Partial, R.AgentDrape, R.BoxDrape, R.GoalDrape,
                                 R.FixedDrape)
  assert not gamespec.is_rule_game(game)
  traced = tabulate.trace(game, cache=False)
  assert tabulate.LAST_WALK[0].startswith('lanes: '), tabulate.LAST_WALK[0]
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
  build = random_warehouses.library_builder(DEFS[k])
  game = build(batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert game.fused.n_dyn == 1 + len(DEFS[k]['boxes']) and game.fused.traced is None
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


@pytest.mark.gpu
@pytest.mark.parametrize('k', [0, 3, 7, 9, 13])
def test_hip_path_through_the_tabulator_gives_the_reference_engines_frames(k):
  """... and the tables so tabulated, on the device (cell-indexed tables or the state table)."""
  from campx import things
  from campx.ascii_art import ascii_art_to_game, Partial
  from campx_amd import rules
  R = rules.bind(things)
  gold = _gold(k)
  T, N = gold['actions'].shape
  game = random_warehouses.build(DEFS[k], ascii_art_to_game, Partial, R.AgentDrape, R.BoxDrape, R.GoalDrape,
                                 R.FixedDrape, batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert game.fused.traced is not None
  assert np.array_equal(first.board.cpu().numpy(), gold['board'][0])
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert np.array_equal(out['obs'].cpu().numpy(), gold['layered'][1:].astype(np.int8))
  assert np.array_equal(out['board'].cpu().numpy(), gold['board'][1:])
  for name in ('reward', 'discount', 'done'):
    assert _same(out[name].cpu().numpy(), gold[name]), name


@pytest.mark.gpu
@pytest.mark.parametrize('B', [4096, 12288])
def test_larger_batches_of_every_level_against_the_oracle_in_order_and_deferred(B):
  """4 096 environments: update pass and render in ONE launch (pipe_multi_kernel<K, ., true>; the
  four-mover levels two launches); 12 288: two launches.  Then the same action streams as deferred
  rollouts (the shared launch / the two-stream form): the same bytes, one call later."""
  T = 32
  for k, d in enumerate(DEFS):
    build = random_warehouses.library_builder(d)
    rng = np.random.RandomState(2100 + k)
    streams = [rng.randint(0, 5, size=(T, B)).astype(np.int8) for _ in range(3)]
    og = cpu.OracleGame.from_description(gamespec.describe(build()))
    game = build(batch=B, device='cuda')
    game.its_showtime()
    want = []
    for i, actions in enumerate(streams):
      out = game.rollout(torch.from_numpy(actions), reset_first=(i == 0))
      ref = og.rollout(actions, reset_first=(i == 0))
      assert np.array_equal(out['obs'].cpu().numpy(), ref['obs']), (k, i)
      assert _same(out['reward'].cpu().numpy(), ref['reward']), (k, i)
      assert np.array_equal(out['done'].cpu().numpy(), ref['done']), (k, i)
      want.append((ref['obs'], ref['reward']))
    twin = build(batch=B, device='cuda')
    twin.its_showtime()
    f = twin.fused
    first = f.rollout_buffers(T)
    bufs = [first, f.rollout_buffers(T, share=first)]
    got = []
    for i, actions in enumerate(streams):
      done = f.rollout_deferred(torch.from_numpy(actions).cuda(), out=bufs[i & 1], reset_first=(i == 0))
      if done is not None:
        got.append((done['obs'].cpu().numpy().copy(), done['reward'].cpu().numpy().copy()))
    last = f.flush()
    got.append((last['obs'].cpu().numpy(), last['reward'].cpu().numpy()))
    assert len(got) == len(want)
    for i, ((obs, reward), (wobs, wreward)) in enumerate(zip(got, want)):
      assert np.array_equal(obs, wobs), (k, i)
      assert _same(reward, wreward), (k, i)
