"""A seeded family of 20 games of the Hello World kind (tests/random_hellos.py) against what the
NOTEBOOK's own RollingDrape / SlidingSprite did on the reference's engine
(tests/golden/random_hellos.npz, make_random_golden.py hellos): random boards of 9 to 40 columns,
one or two rolling drapes, one to five sliding sprites, backdrop patches, random z-orders (sprites
painted before the first drape leave trails in the backdrop) and update schedules; environments
that quit and start over, and one that walks a long straight trail.

Per game: (a) the generator still makes the fixture's game; (b) this repo's generic tier gives
the reference's frames; (c) so does the shape lowering of `gamespec.describe()` run by the C
oracle; (d) the same classes bound afresh - a user's own, to the engine - are recognised
(campx_amd/recognise.py: the per-thing proof) to the very same CampxShapeSpec bytes; (e, GPU) the
HIP path - shape tier, the frame-major path where the board allows it and the serial kernel where
not - gives the reference's frames through both routes, rollout() and play()."""

import ctypes
import json
import os

import numpy as np
import pytest
import torch

from campx_amd import gamespec, recognise
from conftest import GOLDEN_DIR
from oracle import cpu
import random_hellos

DEFS = random_hellos.definitions()
IDS = ['hello{}'.format(k) for k in range(len(DEFS))]


def _gold(k):
  with np.load(os.path.join(GOLDEN_DIR, 'random_hellos.npz')) as f:
    pre = 'k{}_'.format(k)
    return {name[len(pre):]: f[name] for name in f.files if name.startswith(pre)}


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f' or b.dtype.kind == 'f':
    return np.array_equal(a.astype(np.float32), b.astype(np.float32), equal_nan=True)
  return np.array_equal(a, b)


def _spec_bytes(spec):
  return ctypes.string_at(ctypes.addressof(spec), ctypes.sizeof(spec))


def test_the_generator_still_makes_the_games_of_the_fixture():
  assert len(DEFS) == random_hellos.N_GAMES == 20
  for k, d in enumerate(DEFS):
    gold = _gold(k)
    assert [''.join(chr(c) for c in row) for row in gold['art']] == d['art'], k
    assert json.loads(str(gold['meta'])) == dict(drapes=d['drapes'], sprites=d['sprites'],
                                                 z_order=d['z_order'], schedule=d['schedule']), k
  widths = {len(d['art'][0]) for d in DEFS}
  assert min(widths) < 16 and max(widths) >= 32         # the serial kernel's boards and the frame-major path's
  assert sum(min(d['z_order'].index(c) for c in d['drapes']) > 0 for d in DEFS) >= 8       # trails


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_generic_tier_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_hellos.library_builder(DEFS[k])
  for n in range(N):
    game = build()
    obs, reward, discount = game.its_showtime()
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      if game.game_over:
        game = build()
        game.its_showtime()
      obs, reward, discount = game.play(int(gold['actions'][t, n]))
      assert np.array_equal(obs.board.numpy(), gold['board'][t + 1, n].astype(np.uint8)), (n, t)
      assert np.array_equal(obs.layered_board.numpy(), gold['layered'][t + 1, n]), (n, t)
      assert _same(np.float32(np.nan if reward is None else float(reward)), gold['reward'][t, n]), (n, t)
      assert np.float32(discount) == gold['discount'][t, n]
      assert int(game.game_over) == gold['done'][t, n]


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_shape_lowering_run_by_the_c_oracle_gives_them_too(k):
  gold = _gold(k)
  desc = gamespec.describe(random_hellos.library_builder(DEFS[k])())
  assert desc.is_shape_game
  og = cpu.OracleGame.from_description(desc)
  assert [ord(c) for c in og.chars] == gold['chars'].tolist()
  out = og.rollout(gold['actions'], reset_first=True)
  assert _same(out['obs'], gold['layered'][1:].astype(np.int8))
  assert _same(out['board'], gold['board'][1:])
  for name in ('reward', 'discount', 'done'):
    assert _same(out[name], gold[name]), name


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_the_classes_bound_afresh_are_recognised_to_the_same_spec(k):
  lowered = gamespec.lower_shapes(gamespec.describe(random_hellos.library_builder(DEFS[k])()))
  game = random_hellos.library_builder(DEFS[k], rebound=True)()
  assert not gamespec.is_rule_game(game)
  actions = recognise.detect_actions(game)
  assert actions == [0, 1, 2, 3, 4]
  recognised = gamespec.lower_shapes(recognise.shapes(game, actions))
  assert _spec_bytes(recognised) == _spec_bytes(lowered)


@pytest.mark.gpu
@pytest.mark.parametrize('rebound', [False, True], ids=['library', 'rebound'])
@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_hip_path_gives_the_reference_engines_frames(k, rebound):
  from campx_amd import shapes
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_hellos.library_builder(DEFS[k], rebound=rebound)
  game = build(batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert isinstance(game.fused, shapes.ShapeGame)
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
  assert np.array_equal(first.board.cpu().numpy(), gold['board'][0])
  assert np.array_equal(first.layered_board.cpu().numpy(), gold['layered'][0].astype(np.int8))
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert np.array_equal(out['obs'].cpu().numpy(), gold['layered'][1:].astype(np.int8))
  assert np.array_equal(out['board'].cpu().numpy(), gold['board'][1:])
  for name in ('reward', 'discount', 'done'):
    assert _same(out[name].cpu().numpy(), gold[name]), name
  if rebound:
    return
  game = build(batch=N, device='cuda')
  game.its_showtime()
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert np.array_equal(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert np.array_equal(obs.layered_board.cpu().numpy(), gold['layered'][t + 1].astype(np.int8)), t
    assert _same(reward.cpu().numpy(), gold['reward'][t]), t
    assert _same(discount.cpu().numpy(), gold['discount'][t]), t


@pytest.mark.gpu
def test_a_large_batch_of_every_game_on_the_frame_major_path_against_the_oracle():
  """B = 2 048 random action streams with quits, T = 37 (not a multiple of the keyframe interval):
  the kernels the batch takes - the frame-major pair where the spec has tables - against the C
  oracle, observations of every frame."""
  B, T = 2048, 37
  took = 0
  for k, d in enumerate(DEFS):
    build = random_hellos.library_builder(d)
    rng = np.random.RandomState(1300 + k)
    actions = rng.choice(5, size=(T, B), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    game = build(batch=B, device='cuda')
    game.its_showtime()
    took += int(game.fused._tables is not None)
    out = game.rollout(torch.from_numpy(actions))
    ref = cpu.OracleGame.from_description(gamespec.describe(build())).rollout(actions, reset_first=True)
    assert np.array_equal(out['obs'].cpu().numpy(), ref['obs']), k
    assert _same(out['reward'].cpu().numpy(), ref['reward']), k
    assert np.array_equal(out['done'].cpu().numpy(), ref['done']), k
  assert took >= 8, took          # (boards of 16 to 64 columns: the frame-major path)
