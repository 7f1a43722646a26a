"""Shape tier (Hello World: rolling drape, sliding sprites, quit action) on the HIP
device vs the reference-generated golden and the CPU oracle.  Bit-exact."""

import numpy as np
import pytest
import torch

from campx_amd import gamespec
from campx_amd.games import hello_world
from oracle import cpu
from games_under_test import SHAPE_GAMES

pytestmark = pytest.mark.gpu


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def _game(batch, name='hello_world'):
  game = SHAPE_GAMES[name](batch=batch, device='cuda')
  first = game.its_showtime()
  from campx_amd import shapes
  assert isinstance(game.fused, shapes.ShapeGame)
  return game, first


ZOO = sorted(n for n in SHAPE_GAMES if n != 'hello_world')   # tests/shape_zoo.py


@pytest.mark.parametrize('name', ZOO)
def test_zoo_golden_rollout_and_play(name, golden):
  """More arrangements of the two rule classes (odd / 8k+4 / 8k / tiny boards, eight things,
  trails or none, a 150-cell drape, three drapes rewarding in one frame) vs what the REFERENCE
  engine did with them; one rollout, then frame by frame."""
  gold = golden(name)
  T, N = gold['actions'].shape
  game, (obs, reward, discount) = _game(N, name)
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
  assert _same(obs.layered_board.cpu().numpy(), gold['layered'][0])
  assert _same(obs.board.cpu().numpy(), gold['board'][0])
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  for key, want in (('obs', gold['layered'][1:]), ('board', gold['board'][1:]),
                    ('reward', gold['reward']), ('discount', gold['discount']),
                    ('done', gold['done'])):
    assert _same(out[key].cpu().numpy(), want), key
  game, _ = _game(N, name)
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert _same(obs.layered_board.cpu().numpy(), gold['layered'][t + 1]), t
    assert _same(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert _same(reward.cpu().numpy(), gold['reward'][t])
    assert _same(discount.cpu().numpy(), gold['discount'][t])


@pytest.mark.parametrize('name', sorted(SHAPE_GAMES))
def test_keep_obs_false_keeps_the_last_frame_and_the_trails(name, golden):
  """Frames nobody sees are not expanded, but the sprites that paint into the backdrop are
  still painted every frame: the last frame shows every trail."""
  gold = golden(name)
  T, N = gold['actions'].shape
  game, _ = _game(N, name)
  half = T // 2
  out = game.rollout(torch.from_numpy(gold['actions'][:half]), keep_obs=False, want_board=True)
  assert _same(out['obs'].cpu().numpy(), gold['layered'][half])
  assert _same(out['board'].cpu().numpy(), gold['board'][half])
  assert _same(out['reward'].cpu().numpy(), gold['reward'][:half])
  assert _same(out['done'].cpu().numpy(), gold['done'][:half])
  out = game.rollout(torch.from_numpy(gold['actions'][half:]), keep_obs=False)
  assert _same(out['obs'].cpu().numpy(), gold['layered'][-1])
  assert _same(out['discount'].cpu().numpy(), gold['discount'][half:])


@pytest.mark.parametrize('name', ZOO)
@pytest.mark.parametrize('batch', [5, 130])
def test_zoo_random_streams_vs_oracle(name, batch):
  rng = np.random.RandomState(batch)
  game, _ = _game(batch, name)
  og = cpu.OracleGame.from_description(gamespec.describe(SHAPE_GAMES[name]()))
  for launch, T in enumerate([1, 70, 33]):        # across the 64-frame scalar buffer
    actions = rng.choice(5, size=(T, batch), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'board', 'reward', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), ref[k]), (launch, k)


def _kernels_of(fn):
  from torch.profiler import ProfilerActivity, profile
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fn()
    torch.cuda.synchronize()
  return [e.key for e in prof.key_averages() if 'campx_impl' in e.key]


FRAME_MAJOR_GAMES = ['hello_world', 'shape_zoo3', 'shape_zoo4']     # rows of 16 to 64 cells


@pytest.mark.parametrize('name', FRAME_MAJOR_GAMES)
@pytest.mark.parametrize('batch', [4, 64, 1000, 4096])
def test_frame_major_path_against_the_oracle_and_the_serial_kernel(name, batch):
  """Round 5: rollouts that keep every frame run as an update pass + a frame-major render pass
  that computes every row of the observation from 64-bit row words (csrc/k_shape.hip
  shape_render_split_kernel) - Hello World and shape_zoo3 WITH their trails (per-environment
  trail words, a keyframe every fourth frame, the frames since replayed from the offset trace),
  shape_zoo4 without.  Against the oracle, several launches with the state carried over (the
  carried backdrop <-> trail words), a quit in the middle of a launch, and against the serial
  one-wave-per-environment kernel on a twin engine, byte for byte; play() in between runs the
  serial kernel on the same state."""
  from campx_amd import shapes
  game, _ = _game(batch, name)
  assert game.fused._tables is not None
  shapes.FRAME_MAJOR = False
  try:
    serial, _ = _game(batch, name)
  finally:
    shapes.FRAME_MAJOR = True
  assert serial.fused._tables is None
  og = cpu.OracleGame.from_description(gamespec.describe(SHAPE_GAMES[name]()))
  rng = np.random.RandomState(batch)
  quits = 0
  for launch, T in enumerate([1, 70, 33, 9, 4, 5]):
    actions = rng.choice(5, size=(T, batch), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    acts = torch.from_numpy(actions)
    bufs = game.fused.rollout_buffers(T)
    assert bufs['trace'] is not None
    out = game.rollout(acts, out=bufs)
    sbufs = serial.fused.rollout_buffers(T)
    assert sbufs['trace'] is None
    alone = serial.rollout(acts, out=sbufs)
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'reward', 'discount', 'done'):
      if out[k] is None:
        continue
      assert _same(out[k].cpu().numpy(), ref[k]), (launch, k)
      assert _same(out[k].cpu().numpy(), alone[k].cpu().numpy()), (launch, k)
    assert torch.equal(game.fused.pos, serial.fused.pos)
    assert torch.equal(game.fused.backdrop, serial.fused.backdrop), launch
    assert _same(game.fused.ret.cpu().numpy(), serial.fused.ret.cpu().numpy())
    quits += int(ref['done'].sum())
    if launch in (2, 4):
      # play() in between: the serial kernel, on the state the frame-major launches left
      one = rng.randint(0, 4, size=batch).astype(np.int8)
      obs, _, _ = game.play(torch.from_numpy(one))
      obs2, _, _ = serial.play(torch.from_numpy(one))
      ref1 = og.rollout(one[None], reset_first=False)
      assert torch.equal(obs.layered_board, obs2.layered_board)
      assert _same(obs.layered_board.cpu().numpy(), ref1['obs'][0])
  assert quits > 0 or batch < 64


def test_frame_major_is_the_path_a_full_rollout_takes():
  game, _ = _game(64)
  acts = torch.randint(0, 4, (20, 64), dtype=torch.int8, device='cuda')
  bufs = game.fused.rollout_buffers(20)
  game.rollout(acts, out=bufs)
  for attempt in range(3):           # (the profiler now and then hands back an empty trace)
    names = _kernels_of(lambda: game.rollout(acts, out=bufs))
    if names:
      break
  short = sorted(n.split('(')[0].split('::')[-1].split('<')[0] for n in names)
  assert 'shape_render_split_kernel' in short and 'shape_update_split_kernel' in short and \
      'shape_rollout_kernel' not in short, names


_CHUNKED = r'''
import sys
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch
from campx_amd import _hip, gamespec
_hip.config_set('shape_chunk_kf', 1)        # chunks of 1 000 environment-frames
from games_under_test import SHAPE_GAMES
from oracle import cpu
for name, batch in (('hello_world', 64), ('shape_zoo3', 1000), ('shape_zoo4', 64)):
  game = SHAPE_GAMES[name](batch=batch, device='cuda')
  game.its_showtime()
  og = cpu.OracleGame.from_description(gamespec.describe(SHAPE_GAMES[name]()))
  rng = np.random.RandomState(3)
  for launch, T in enumerate([70, 33, 5]):
    actions = rng.choice(5, size=(T, batch), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions))
    assert out['trace'] is not None
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'reward', 'discount', 'done'):
      if out[k] is not None:
        a, b = out[k].cpu().numpy(), ref[k]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                              b.view(np.uint32) if b.dtype == np.float32 else b), (name, launch, k)
print('ok')
'''


def test_long_launches_run_as_chunks_of_frames():
  """A launch of more than `shape_chunk_kf` thousand environment-frames (a library setting, default 2 000) runs
  as chunks - update pass and render alternating, positions / trail words / returns carried from
  chunk to chunk - so that a chunk's offset trace and keyframes stay in the memory-side cache.
  With the bound set to 1 000 environment-frames: 64 environments x 70 frames in chunks of 12, 1 000
  environments in chunks of 4 (one key interval): against the oracle, byte for byte."""
  import os
  import subprocess
  import sys
  from conftest import REPO
  out = subprocess.run([sys.executable, '-c', _CHUNKED % dict(repo=REPO, tests=os.path.join(REPO, 'tests'))],
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  assert out.stdout.strip().endswith('ok')


def test_frame_major_needs_whole_chunks_and_every_frame():
  from campx_amd import shapes
  game, _ = _game(5)                        # 5 x 3 276 bytes per frame: not whole 16-byte chunks
  acts = torch.randint(0, 4, (10, 5), dtype=torch.int8, device='cuda')
  for attempt in range(3):
    names = _kernels_of(lambda: game.rollout(acts))
    if names:
      break
  assert len(names) == 1 and 'shape_rollout_kernel' in names[0], names
  game, _ = _game(64)
  assert game.fused.rollout_buffers(10, keep_obs=False)['trace'] is None
  assert game.fused.rollout_buffers(10, want_board=True)['trace'] is None
  assert game.fused.rollout_buffers(10, obs_dtype=torch.float16)['trace'] is None
  game, _ = _game(64, 'shape_zoo0')         # rows of 9 cells: not a game for the row-word render
  assert game.fused._tables is None and game.fused.rollout_buffers(10)['trace'] is None


def test_golden_rollout(golden):
  """The notebook's own classes on the reference engine -> same frames, incl. the
  trails sprites 1 and 2 leave in the backdrop and the quit at frame 25 of env 0."""
  gold = golden('hello_world')
  T, N = gold['actions'].shape
  game, (obs, reward, discount) = _game(N)
  assert reward is None and discount == 1.0
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
  assert _same(obs.layered_board.cpu().numpy(), gold['layered'][0])
  assert _same(obs.board.cpu().numpy(), gold['board'][0])
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert _same(out['obs'].cpu().numpy(), gold['layered'][1:])
  assert _same(out['board'].cpu().numpy(), gold['board'][1:])
  assert _same(out['reward'].cpu().numpy(), gold['reward'])
  assert _same(out['discount'].cpu().numpy(), gold['discount'])
  assert _same(out['done'].cpu().numpy(), gold['done'])
  assert gold['done'][25, 0] == 1 and np.isnan(gold['reward'][25, 0])


def test_play_frame_by_frame(golden):
  gold = golden('hello_world')
  T, N = gold['actions'].shape
  game, _ = _game(N)
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert _same(obs.layered_board.cpu().numpy(), gold['layered'][t + 1]), t
    assert _same(obs.board.cpu().numpy(), gold['board'][t + 1])
    for i, ch in enumerate(game.fused.chars):
      assert _same(obs.layers[ch].cpu().numpy(), gold['layered'][t + 1][:, i])
    assert _same(reward.cpu().numpy(), gold['reward'][t])
    assert _same(discount.cpu().numpy(), gold['discount'][t])
    assert _same(game.fused.done.cpu().numpy(), gold['done'][t])


@pytest.mark.parametrize('batch', [1, 3, 64, 1000])
def test_random_streams_vs_oracle(batch):
  """Ragged batches, state (offsets, backdrop, latch) carried across launches, quits."""
  rng = np.random.RandomState(batch)
  game, _ = _game(batch)
  og = cpu.OracleGame.from_description(gamespec.describe(hello_world.build()))
  # the running return: rewards since the latest rebuild; a frame nobody rewards (the quit
  # action: reward None = NaN) adds nothing to it
  want_ret, ended = np.zeros(batch, np.float32), np.zeros(batch, bool)
  for launch, T in enumerate([1, 9, 40]):
    actions = rng.choice(5, size=(T, batch), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'board', 'reward', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), ref[k]), (launch, k)
    for t in range(T):
      want_ret = np.where(ended, 0, want_ret).astype(np.float32)
      want_ret = want_ret + np.where(np.isnan(ref['reward'][t]), 0, ref['reward'][t]).astype(np.float32)
      ended = ref['done'][t] == 1
    assert _same(game.fused.ret.cpu().numpy(), want_ret), launch
  assert ref['done'].sum() > 0 or batch < 3
  assert not np.isnan(want_ret).any()


def test_larger_batch_and_invariants():
  batch, T = 8192, 50
  rng = np.random.RandomState(7)
  actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
  game, _ = _game(batch)
  out = game.rollout(torch.from_numpy(actions), reset_first=True)
  og = cpu.OracleGame.from_description(gamespec.describe(hello_world.build()))
  stride = 16
  ref = og.rollout(np.ascontiguousarray(actions[:, ::stride]), reset_first=True, want_board=False)
  assert _same(out['obs'][:, ::stride].cpu().numpy(), ref['obs'])
  assert _same(out['reward'][:, ::stride].cpu().numpy(), ref['reward'])
  assert _same(out['done'][:, ::stride].cpu().numpy(), ref['done'])
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1


def test_bad_action_ids_move_nothing_and_are_counted():
  game, _ = _game(64)
  game.fused.validate_actions = 'sync'
  before = game.fused._obs.clone()
  with pytest.raises(ValueError, match='64 action ids'):
    game.play(torch.full((64,), 7))
  game.fused.validate_actions = False
  obs, reward, discount = game.play(torch.full((64,), 7))
  assert torch.equal(obs.layered_board, before)
  assert torch.isnan(reward).all() and (discount == 1).all()


def _limit_game(H, W, batch=None, device=None):
  """Eight things (three big rolling drapes, four sprites, a static one) on an H x W board."""
  from campx_amd import rules
  from campx_amd.ascii_art import ascii_art_to_game, Partial
  art = [[' '] * W for _ in range(H)]
  for r in range(2, 14):
    for c in range(2, W - 2):
      art[r][c] = '@'
  for r in range(15, 24):
    for c in range(1, W - 1):
      art[r][c] = '%'
  for r in range(25, H - 1):
    for c in range(4, W - 4):
      art[r][c] = '&'
  for i, ch in enumerate('1234'):
    art[0][2 + 3 * i] = ch
  for c in range(W):
    art[H - 1][c] = '#'
  sprites = {ch: Partial(rules.SlidingSprite, i) for i, ch in enumerate('1234')}
  drapes = {'@': Partial(rules.RollingDrape, move_reward=0.5),
            '%': Partial(rules.RollingDrape, roll_axes=(1, 1, 0, 0), roll_shifts=(3, -5, 2, -2),
                         move_reward=0.25, quit_action=None),
            '&': Partial(rules.RollingDrape, move_reward=1.0, quit_action=None),
            '#': rules.FixedDrape}
  return ascii_art_to_game([''.join(r) for r in art], what_lies_beneath=' ', sprites=sprites,
                           drapes=drapes, z_order='12@3%#&4', update_schedule='4&#%3@21',
                           batch=batch, device=device)


@pytest.mark.parametrize('H,W', [(32, 32), (31, 33)])
def test_boards_at_the_tier_limit(H, W):
  """1 024 cells (the limit; four-cell path) and 1 023 (byte path), eight things, drapes of
  hundreds of cells (several passes of the paint loop), against the oracle."""
  batch = 37
  game = _limit_game(H, W, batch=batch, device='cuda')
  game.its_showtime()
  og = cpu.OracleGame.from_description(gamespec.describe(_limit_game(H, W)))
  rng = np.random.RandomState(1)
  for launch, T in enumerate((1, 70)):
    actions = rng.choice(5, size=(T, batch), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'board', 'reward', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), ref[k]), (launch, k)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('name,B', [('hello_world', 777), ('shape_zoo2', 64), ('shape_zoo4', 1000)])
def test_sixteen_bit_observations(name, B, dtype):
  """f16 / bf16 layered boards straight from the shape kernel (rollouts and play()): the
  int8 frames, converted."""
  build = SHAPE_GAMES[name]
  game = build(batch=B, device='cuda')
  game.its_showtime()
  rng = np.random.RandomState(31)
  actions = torch.from_numpy(rng.randint(0, 4, size=(25, B)).astype(np.int8))
  ref = game.rollout(actions, reset_first=True, want_board=True)
  out = game.rollout(actions, reset_first=True, want_board=True, obs_dtype=dtype)
  assert out['obs'].dtype == dtype
  assert torch.equal(out['obs'].to(torch.int8), ref['obs']) and torch.equal(out['board'], ref['board'])
  assert set(out['obs'].unique().tolist()) == {0.0, 1.0}
  game.fused.reset()
  game.fused.set_play_obs_dtype(dtype)
  first, _, _ = game.fused.reset()
  assert first.layered_board.dtype == dtype
  for t in range(6):
    obs, _, _ = game.play(actions[t])
    assert obs.layered_board.dtype == dtype
    assert torch.equal(obs.layered_board.to(torch.int8), ref['obs'][t]), t
    assert torch.equal(obs.layers[game.fused.chars[0]], obs.layered_board[:, 0])
