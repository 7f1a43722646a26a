"""HIP kernel (through the C ABI, via campx_amd.fused) vs goldens and the CPU oracle.

Bit-exact: observations, boards, done flags are integers; rewards/discounts are
float32 compared bitwise (NaN == NaN).
"""

import numpy as np
import pytest
import torch

from campx_amd import gamespec
from oracle import cpu
from games_under_test import FUSED_GAMES
from campx_amd.games import boat_race

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['table', 'interpreter'])
def update_pass_mode(request):
  """Every test runs with the (cell, action) transition table that
  campx_spec_compile() builds for one-mover games, and with the rule interpreter
  only."""
  from campx_amd import fused
  saved = fused.COMPILE_TABLE
  fused.COMPILE_TABLE = request.param == 'table'
  yield request.param
  fused.COMPILE_TABLE = saved


@pytest.fixture(autouse=True, params=['split', 'fused'])
def rollout_path(request):
  """... and with rollouts as two kernels (update pass -> trace -> render) or as the
  single fused kernel."""
  from campx_amd import fused
  saved = fused.SPLIT_ROLLOUT
  fused.SPLIT_ROLLOUT = request.param == 'split'
  yield request.param
  fused.SPLIT_ROLLOUT = saved


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def _fused(name, batch):
  from campx_amd import fused
  game = FUSED_GAMES[name](batch=batch, device='cuda')
  first = game.its_showtime()
  # every library game fits a table: (cell, action), (cell, cell, action) or, for three and
  # four movers, the global (cell, ..., action) table
  assert game.fused.uses_table == fused.COMPILE_TABLE
  return game, first


def _check_rollout(out, ref, keep_obs=True):
  assert _same(out['obs'].cpu().numpy(), ref['obs'])
  if out['board'] is not None:
    assert _same(out['board'].cpu().numpy(), ref['board'])
  assert _same(out['discount'].cpu().numpy(), ref['discount'])
  assert _same(out['done'].cpu().numpy(), ref['done'])
  if out['reward'] is None:
    assert np.isnan(ref['reward']).all()
  else:
    assert _same(out['reward'].cpu().numpy(), ref['reward'])
  if ref.get('perf') is not None:
    assert _same(out['perf'].cpu().numpy(), ref['perf'])
  # (a golden without 'perf' says nothing: sokoban's penalty is checked against its boards,
  # test_sokoban_side_effects_penalty_from_the_golden_boards)


@pytest.mark.parametrize('name', sorted(FUSED_GAMES))
def test_golden_trajectories(name, golden):
  """Same action streams the reference was run on -> same frames, bit for bit."""
  gold = golden(name)
  T, N = gold['actions'].shape
  game, (obs, reward, discount) = _fused(name, N)
  assert reward is None and discount == 1.0
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
  assert _same(obs.layered_board.cpu().numpy(), gold['layered'][0])
  assert _same(obs.board.cpu().numpy(), gold['board'][0])
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  ref = dict(obs=gold['layered'][1:], board=gold['board'][1:],
             reward=gold['reward'], discount=gold['discount'], done=gold['done'],
             perf=gold.get('perf'))
  _check_rollout(out, ref)


@pytest.mark.parametrize('name', sorted(FUSED_GAMES))
def test_play_matches_golden_frame_by_frame(name, golden):
  """Engine.play() (one launch per frame) including the layers dict views."""
  gold = golden(name)
  T, N = gold['actions'].shape
  T = min(T, 40)
  game, _ = _fused(name, N)
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert _same(obs.layered_board.cpu().numpy(), gold['layered'][t + 1])
    assert _same(obs.board.cpu().numpy(), gold['board'][t + 1])
    for i, ch in enumerate(game.fused.chars):
      assert _same(obs.layers[ch].cpu().numpy(), gold['layered'][t + 1][:, i])
    if reward is None:
      assert np.isnan(gold['reward'][t]).all()
    else:
      assert _same(reward.cpu().numpy(), gold['reward'][t])
    assert _same(discount.cpu().numpy(), gold['discount'][t])
    assert _same(game.fused.done.cpu().numpy(), gold['done'][t])
    if 'perf' in gold:
      assert _same(game.fused.perf.cpu().numpy(), gold['perf'][t])


@pytest.mark.parametrize('name', ['boat_race', 'wall_world', 'sokoban', 'demo3', 'sokoban_l2'])
@pytest.mark.parametrize('batch', [1, 63, 64, 65, 1000])
def test_random_streams_vs_oracle(name, batch):
  """Ragged batch sizes (tail waves, unaligned strides), state carried across launches."""
  rng = np.random.RandomState(batch * 7 + len(name))
  game, _ = _fused(name, batch)
  og = cpu.OracleGame.from_description(gamespec.describe(FUSED_GAMES[name]()))
  for launch, T in enumerate([1, 17, 50]):
    actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    _check_rollout(out, ref)


def test_one_hot_actions_and_validation():
  game, _ = _fused('boat_race', 128)
  game.fused.validate_actions = 'sync'          # raise in the offending call
  ids = torch.randint(0, 5, (128,))
  onehot = torch.nn.functional.one_hot(ids, 5).float()
  obs_a, r_a, _ = game.play(onehot)
  board_a, r_a = obs_a.board.clone(), r_a.clone()
  game2, _ = _fused('boat_race', 128)
  obs_b, r_b, _ = game2.play(ids)
  assert torch.equal(board_a, obs_b.board) and torch.equal(r_a, r_b)
  with pytest.raises(ValueError):
    game.play(torch.full((128,), 5))
  with pytest.raises(ValueError):
    game.play(torch.full((128,), 256))          # must not wrap to 0 when narrowed to int8
  game.play(ids)                                # the error state was cleared
  bad = onehot.clone()
  bad[3] = 0.5
  with pytest.raises(ValueError):
    game.play(bad)

@pytest.mark.gpu
def test_five_environments_and_a_list_of_five_zero_one_integers_is_ambiguous():
  """[0, 1, 0, 0, 0] with batch == 5: one one-hot action for all, or five ids?  Refused."""
  game, _, _, _ = boat_race.make_game(batch=5, device='cuda')
  with pytest.raises(ValueError, match='ambiguous'):
    game.play([0, 1, 0, 0, 0])
  _, reward, _ = game.play([0., 1., 0., 0., 0.])              # floats: a one-hot action
  assert reward.shape == (5,)
  game.play(torch.tensor([0, 1, 0, 0, 0]))                    # a tensor: five ids
  other, _, _, _ = boat_race.make_game(batch=6, device='cuda')
  other.play([0, 1, 0, 0, 0])                                 # any other batch: the one-hot list


@pytest.mark.gpu
def test_one_frame_rows_must_reach_the_padded_pitch_of_the_trace():
  """A caller's own contiguous [1, B] reward beside a padded trace (B % 16 != 0): the update
  kernels would store the row's last 16-element group whole - refused instead of written past
  the end (the advisor's round-3 finding)."""
  from campx_amd import _hip
  from campx_amd.games import sokoban
  B = 100
  game = sokoban.build(batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  out = f.rollout_buffers(1)
  if out['trace'] is None:
    pytest.skip('the single fused kernel keeps no trace (no padded pitch to disagree with)')
  assert out['trace'].stride(0) == 112
  acts = torch.zeros((1, B), dtype=torch.int8, device='cuda')
  f.rollout(acts, out=out)                                    # its own buffers: fine
  tight = dict(out)
  tight['reward'] = torch.empty((1, B), dtype=torch.float32, device='cuda')
  with pytest.raises(RuntimeError, match='must reach 112 elements'):
    f.rollout(acts, out=tight)



def test_lazy_validation_of_one_hot_rows():
  """Default mode, one-hot input: a row that is not exactly one-hot moves nothing (id 5)
  and is reported by the step kernel through the lazy flag - no synchronisation in play()."""
  game, _ = _fused('boat_race', 64)
  ids = torch.randint(0, 5, (64,))
  onehot = torch.nn.functional.one_hot(ids, 5).float().cuda()
  bad = onehot.clone()
  bad[5] = 0.0                                  # all zeros
  bad[9, :2] = 1.0                              # two ones
  pos_before = game.fused.pos.clone()
  game.play(bad)
  pos_after = game.fused.pos.cpu()
  assert torch.equal(pos_after[:, 5], pos_before.cpu()[:, 5])     # stayed
  assert torch.equal(pos_after[:, 9], pos_before.cpu()[:, 9])
  with pytest.raises(ValueError, match='2 action ids'):
    game.fused.check_actions()
  game.play(onehot)
  game.fused.check_actions()                    # clean again


def test_lazy_validation_flag_in_host_mapped_memory():
  """Default mode: the kernel that reads the ids raises a flag in pinned host memory;
  the host looks at it without synchronising, so the error surfaces at the latest in
  the first call after a synchronisation - and check_actions() forces it."""
  game, _ = _fused('boat_race', 256)
  assert game.fused.validate_actions is True
  good = torch.randint(0, 5, (256,), dtype=torch.int8, device='cuda')
  bad = good.clone()
  bad[17] = 9
  bad[200] = -3
  game.play(good)
  game.fused.check_actions()                    # nothing wrong so far
  try:
    game.play(bad)                              # may or may not have been seen yet
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match='2 action ids'):
      game.play(good)                           # flag is up: plain host read finds it
  except ValueError as e:
    assert '2 action ids' in str(e)
  game.play(good)                               # cleared
  game.fused.check_actions()
  # rollouts count through the same flag, on every kernel path
  acts = torch.randint(0, 5, (40, 256), dtype=torch.int8, device='cuda')
  acts[7, 3] = 77
  try:
    game.rollout(acts, reset_first=True)        # a launch in several chunks may already see it
    with pytest.raises(ValueError, match='1 action ids'):
      game.fused.check_actions()
  except ValueError as e:
    assert '1 action ids' in str(e)
  game.fused.validate_actions = False           # never looks
  game.rollout(acts, reset_first=True)
  game.fused.check_actions()


@pytest.mark.parametrize('name', ['boat_race', 'sokoban', 'sokoban_l2', 'demo4'])
def test_keep_obs_false_leaves_last_frame(name, golden):
  """Only the last frame's observation (and board) is kept - the two-kernel path renders
  just that one from the last row of the trace; the per-frame scalars are all there; and
  the state carries into the next launch."""
  gold = golden(name)
  T, N = gold['actions'].shape
  game, _ = _fused(name, N)
  half = T // 2
  out = game.rollout(torch.from_numpy(gold['actions'][:half]), keep_obs=False, want_board=True)
  assert out['obs'].shape == gold['layered'][0].shape
  assert _same(out['obs'].cpu().numpy(), gold['layered'][half])
  assert _same(out['board'].cpu().numpy(), gold['board'][half])
  assert _same(out['discount'].cpu().numpy(), gold['discount'][:half])
  assert _same(out['done'].cpu().numpy(), gold['done'][:half])
  out = game.rollout(torch.from_numpy(gold['actions'][half:]), keep_obs=False)
  assert _same(out['obs'].cpu().numpy(), gold['layered'][-1])
  if out['reward'] is not None:
    assert _same(out['reward'].cpu().numpy(), gold['reward'][half:])


def test_reset_first_starts_new_episode(golden):
  gold = golden('sokoban')
  game, _ = _fused('sokoban', gold['actions'].shape[1])
  acts = torch.from_numpy(gold['actions'])
  a = game.rollout(acts, reset_first=True)
  b = game.rollout(acts, reset_first=True)
  assert torch.equal(a['obs'], b['obs']) and torch.equal(a['reward'], b['reward'])
  # the running return restarted with the episode: it holds the rewards since the last
  # rebuild (the frame after the latest termination before the final frame)
  reward, done = b['reward'].cpu().numpy(), b['done'].cpu().numpy()
  want = np.zeros(reward.shape[1], np.float32)
  for t in range(reward.shape[0]):
    restart = done[t - 1] == 1 if t else np.ones(reward.shape[1], bool)
    want = np.where(restart, 0, want).astype(np.float32) + reward[t]
  assert _same(game.fused.ret.cpu().numpy(), want)


def test_sokoban_side_effects_penalty_from_the_golden_boards(golden):
  """The hidden (side-effects) performance of sokoban - -5 while a box stands next to a wall,
  -10 in a corner (games/sokoban.py wall_classes) - against a restatement that
  shares nothing with the kernels or the oracle: where the golden BOARDS (the reference
  engine's frames) show the boxes.  One, two and three boxes: pair and tuple tables."""
  from games_under_test import SOKOBAN_LEVEL, sokoban_penalty_from_boards
  for name, level in sorted(SOKOBAN_LEVEL.items()):
    gold = golden(name)
    want = sokoban_penalty_from_boards(gold, level)
    game, _ = _fused(name, gold['actions'].shape[1])
    out = game.rollout(torch.from_numpy(gold['actions']))
    assert out['perf'] is not None
    assert np.array_equal(out['perf'].cpu().numpy().astype(np.int32), want), name
    game, _ = _fused(name, gold['actions'].shape[1])
    for t in range(20):
      game.play(torch.from_numpy(gold['actions'][t]))
      assert np.array_equal(game.fused.perf.cpu().numpy().astype(np.int32), want[t]), (name, t)


def _oracle_frames(og, actions, frames):
  """Run the oracle over `actions`, keeping the full observation of the frames in
  `frames` only (it keeps either every frame or the last one of a call, so the episode
  is cut into calls that end at the sampled frames; its state carries across calls)."""
  parts, obs_at, start = [], {}, 0
  for stop in sorted(set(frames) | {actions.shape[0] - 1}):
    ref = og.rollout(actions[start:stop + 1], reset_first=(start == 0), keep_obs=False,
                     want_board=False)
    parts.append(ref)
    obs_at[stop] = ref['obs']
    start = stop + 1
  whole = {k: np.concatenate([p[k] for p in parts]) for k in ('reward', 'discount', 'done')}
  whole['perf'] = (np.concatenate([p['perf'] for p in parts])
                   if parts[0]['perf'] is not None else None)
  return whole, obs_at


def _check_full_size(game, og_factory, actions):
  """BASELINE-size parity: every frame's reward / discount / done / perf for every
  environment; the FULL observation of all environments at frames 0, 1, T/2, T-2, T-1;
  the full observation of a strided sample of 4096 environments at EVERY frame; and a
  size-independent invariant: every cell of every environment shows exactly one
  character in every frame."""
  T, batch = actions.shape
  out = game.rollout(torch.from_numpy(actions), reset_first=True)
  frames = sorted({0, 1, T // 2, T - 2, T - 1})
  ref, obs_at = _oracle_frames(og_factory(), actions, frames)
  assert _same(out['discount'].cpu().numpy(), ref['discount'])
  assert _same(out['done'].cpu().numpy(), ref['done'])
  assert _same(out['reward'].cpu().numpy(), ref['reward'])
  if ref['perf'] is not None:
    assert _same(out['perf'].cpu().numpy(), ref['perf'])
  for t in frames:
    assert _same(out['obs'][t].cpu().numpy(), obs_at[t]), 'frame %d' % t
  stride = max(1, batch // 4096)
  sample = og_factory().rollout(np.ascontiguousarray(actions[:, ::stride]), reset_first=True,
                                keep_obs=True, want_board=False)
  assert _same(out['obs'][:, ::stride].cpu().numpy(), sample['obs'])
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1
  # the trace is a compact trajectory in its own right: cell of the first mover
  if out['trace'] is not None:
    cells = (out['trace'][0] & 0x7f).long()
    layer = game.fused.spec.dyn_layer[0]
    where = out['obs'][:, :, layer].reshape(T, batch, -1).long().argmax(dim=2)
    shown = out['trace'][0] >> 7
    assert torch.equal(cells[shown == 1], where[shown == 1])


@pytest.mark.parametrize('name,batch,T', [('boat_race', 65536, 100),
                                          ('wall_world', 262144, 100),
                                          ('sokoban', 131072, 100),
                                          ('boat_race', 524288, 100)])
def test_full_size_vs_oracle(name, batch, T):
  """The three single-GPU BASELINE.json configurations at their full batch and the
  100-frame episode the bench times - and config 5's GLOBAL batch (524 288 = 8 x 65 536
  environments) on one GPU: what the eight shards compute together, here in one piece
  (its 52 MB trace plane also takes the launch through the chunked path)."""
  rng = np.random.RandomState(1234)
  actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
  game, _ = _fused(name, batch)
  desc = gamespec.describe(FUSED_GAMES[name]())
  _check_full_size(game, lambda: cpu.OracleGame.from_description(desc), actions)


def test_three_boxes_at_a_larger_batch_vs_oracle():
  """K = 4 (rule interpreter in both kernel paths) beyond the 32-environment golden."""
  batch, T = 16384, 100
  rng = np.random.RandomState(77)
  actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
  game, _ = _fused('sokoban_l2', batch)
  desc = gamespec.describe(FUSED_GAMES['sokoban_l2']())
  _check_full_size(game, lambda: cpu.OracleGame.from_description(desc), actions)


COURTYARD_ART = ['#########',
                 '#A  *   #',
                 '# ## #> #',
                 '#  *    #',
                 '#> #  * #',
                 '#     # #',
                 '#########']


def _courtyard_engine(batch):
  from campx_amd import rules
  from campx_amd.ascii_art import ascii_art_to_game, Partial
  return ascii_art_to_game(
      COURTYARD_ART, what_lies_beneath=' ',
      drapes={'A': Partial(rules.AgentDrape, blocking_chars='#', step_reward=-0.5,
                           reward_chars='*'),
              '>': Partial(rules.DirectionalHoverRewardDrape,
                           dctns=torch.tensor([0., 2., 0., 0., 0.]), base_reward=0.25),
              '#': rules.FixedDrape, '*': rules.FixedDrape},
      z_order='*>A#', update_schedule=[['A', '>'], ['*', '#']], batch=batch, device='cuda')


def _courtyard_description():
  """The same game written down by hand as the oracle's input - NOT read back from the
  engine with gamespec.describe(), so a mis-read z-order / schedule / mask in describe()
  (which the HIP lowering goes through) cannot hit both sides."""
  art = np.array([list(row) for row in COURTYARD_ART])
  mask = lambda ch: (art == ch).astype(np.uint8)
  entities = [
      gamespec.EntityDesc('A', 'agent', 0, mask('A'),
                          dict(op='agent', blocking='#', step_reward=-0.5, reward_chars='*')),
      gamespec.EntityDesc('>', 'dir_hover', 0, mask('>'),
                          dict(op='dir_hover', agents='A', dctns=[0., 2., 0., 0., 0.],
                               base_reward=0.25)),
      gamespec.EntityDesc('*', 'fixed', 1, mask('*'), {}),
      gamespec.EntityDesc('#', 'fixed', 1, mask('#'), {}),
  ]
  backdrop = np.full(art.shape, ord(' '), np.uint8)
  return gamespec.GameDescription(7, 9, [' ', '#', '*', '>', 'A'], backdrop, entities,
                                  ['*', '>', 'A', '#'])


def test_full_size_on_an_art_outside_the_goldens_with_a_hand_written_description():
  batch, T = 65536, 100
  rng = np.random.RandomState(99)
  actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
  game = _courtyard_engine(batch)
  game.its_showtime()
  desc = _courtyard_description()
  _check_full_size(game, lambda: cpu.OracleGame.from_description(desc), actions)


def test_two_mover_game_on_an_odd_sized_board():
  """5x5 two-mover game: n = 5*HW^2 is not a multiple of 8 (the pair-table scratch
  layout of campx_pair_table_build must not depend on that)."""
  from campx_amd import rules
  from campx_amd.ascii_art import ascii_art_to_game, Partial
  art = ['#####', '#A  #', '# X #', '#  G#', '#####']

  def build(batch=None, device=None):
    return ascii_art_to_game(
        art, what_lies_beneath=' ',
        drapes={'#': rules.FixedDrape,
                'A': Partial(rules.AgentDrape, blocking_chars='#X'),
                'X': Partial(rules.BoxDrape, agent_char='A', blocking_chars='#'),
                'G': Partial(rules.GoalDrape, agent_char='A', step_reward=-1, goal_reward=10)},
        update_schedule=[['X'], ['A', 'G', '#']], z_order='GXA#', batch=batch, device=device)
  batch = 320
  game = build(batch, 'cuda')
  game.its_showtime()
  rng = np.random.RandomState(4)
  og = cpu.OracleGame.from_description(gamespec.describe(build()))
  for launch, T in enumerate([3, 40]):
    actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    _check_rollout(out, og.rollout(actions, reset_first=(launch == 0)))


def test_return_gatherer_on_gpu_single_rank_rccl(tmp_path):
  """The RCCL all-gather of episode returns (bench.py's N>1 path) with one rank."""
  import subprocess
  import sys
  import os
  code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
import socket
with socket.socket() as _s:
    _s.bind(('127.0.0.1', 0)); _port = _s.getsockname()[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_port))
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from campx_amd.distributed import ReturnGatherer
from campx_amd.games import boat_race
game = boat_race.build(batch=4096, device=dev); game.its_showtime()
g = ReturnGatherer(4096, dev, dist)
acts = torch.randint(0, 5, (50, 4096), dtype=torch.int8, device=dev)
for episode in range(3):
    out = game.rollout(acts, reset_first=True, keep_obs=False)
    g.gather_async(game.fused.ret)
got = g.wait(); torch.cuda.synchronize()
assert torch.equal(got, out['reward'].sum(0)), 'gathered returns differ'
dist.destroy_process_group()
print('ok')
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
  assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize('name', ['boat_race', 'sokoban', 'sokoban_l2'])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_sixteen_bit_observations(name, dtype, golden):
  """obs_dtype=float16/bfloat16: exactly `layered_board.float()` of the golden frames
  (0.0 / 1.0 are exact in both formats)."""
  gold = golden(name)
  game, _ = _fused(name, gold['actions'].shape[1])
  out = game.rollout(torch.from_numpy(gold['actions']), obs_dtype=dtype)
  assert out['obs'].dtype == dtype
  want = torch.from_numpy(gold['layered'][1:].astype(np.float32))
  assert torch.equal(out['obs'].float().cpu(), want)
  assert _same(out['reward'].cpu().numpy(), gold['reward'])
  with pytest.raises(ValueError):
    game.rollout(torch.from_numpy(gold['actions']), obs_dtype=dtype, keep_obs=False)


@pytest.mark.parametrize('name,batch', [('boat_race', 5), ('boat_race', 1001), ('sokoban', 63),
                                        ('sokoban_l2', 77), ('wall_world', 3), ('boat_race', 1000),
                                        ('boat_race', 8), ('wall_world', 1), ('wall_world', 77)])
def test_sixteen_bit_observations_at_odd_batch_sizes(name, batch):
  """Frames that are not whole 16-byte chunks (batch * L*H*W odd multiples): the render
  kernel's chunks straddle frames; across two launches so that a launch boundary falls
  mid-chunk too.  And frames of 8 (mod 16) elements - boat race at B = 1 000 or 8: whole 16-byte
  stores of 16-bit elements, but every other frame starts in the middle of a 16-byte chunk of
  the image (tests/test_api_sequences.py found the first 8 elements of those frames wrong; the
  sokoban cases of this class passed by luck - zeros either way)."""
  rng = np.random.RandomState(batch)
  game, _ = _fused(name, batch)
  og = cpu.OracleGame.from_description(gamespec.describe(FUSED_GAMES[name]()))
  for launch, T in enumerate([7, 33]):
    actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), obs_dtype=torch.bfloat16)
    ref = og.rollout(actions, reset_first=(launch == 0))
    assert out['obs'].dtype == torch.bfloat16
    assert torch.equal(out['obs'].float().cpu(), torch.from_numpy(ref['obs'].astype(np.float32)))
    assert _same(out['done'].cpu().numpy(), ref['done'])


def _large_row_game(H, W, extra, boxes='', batch=None, device=None):
  from campx_amd import rules
  from campx_amd.ascii_art import ascii_art_to_game, Partial
  art = [[' '] * W for _ in range(H)]
  for c in range(W):
    art[0][c] = art[H - 1][c] = '#'
  for r in range(H):
    art[r][0] = art[r][W - 1] = '#'
  art[1][1] = 'A'
  for i, ch in enumerate(boxes):
    art[2][2 + 2 * i] = ch
  for i, ch in enumerate(extra):
    art[3 + i // (W - 4)][2 + i % (W - 4)] = ch
  drapes = {'#': rules.FixedDrape,
            'A': Partial(rules.AgentDrape, blocking_chars='#' + boxes, step_reward=-1.0,
                         reward_chars=extra[:1])}
  for ch in extra:
    drapes[ch] = rules.FixedDrape
  for ch in boxes:
    drapes[ch] = Partial(rules.BoxDrape, agent_char='A', blocking_chars='#' + boxes.replace(ch, ''))
  schedule = ([list(boxes)] if boxes else []) + [['A'] + list(extra) + ['#']]
  return ascii_art_to_game([''.join(r) for r in art], what_lies_beneath=' ', drapes=drapes,
                           z_order=extra + boxes + 'A#', update_schedule=schedule,
                           batch=batch, device=device)


@pytest.mark.parametrize('H,W,extra,boxes', [(10, 10, 'abcdefgh', ''), (8, 16, 'abcdefghijklm', ''),
                                             (8, 16, 'abcdefghijkl', 'X'), (10, 12, 'abcdefghi', 'XY')])
def test_large_rows_up_to_128_cells_and_16_characters(H, W, extra, boxes):
  """Rows of 1 100 - 2 048 bytes: the 64-environment LDS images of the one-frame and fused
  kernels need up to 146 KiB of the CU's 160 (hipFuncAttributeMaxDynamicSharedMemorySize);
  the two-kernel path does not care."""
  import functools
  build = functools.partial(_large_row_game, H, W, extra, boxes)
  batch = 130
  game = build(batch=batch, device='cuda')
  obs, reward, discount = game.its_showtime()
  og = cpu.OracleGame.from_description(gamespec.describe(build()))
  obs0, board0 = og.first_frame()
  assert _same(obs.layered_board.cpu().numpy()[7], obs0)
  rng = np.random.RandomState(H * W)
  actions = rng.randint(0, 5, size=(40, batch)).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions), want_board=True, reset_first=True)
  ref = og.rollout(actions, reset_first=True)
  _check_rollout(out, ref)
  more = rng.randint(0, 5, size=(6, batch)).astype(np.int8)
  ref = og.rollout(more)
  for t in range(6):
    obs, reward, discount = game.play(torch.from_numpy(more[t]))
    assert _same(obs.layered_board.cpu().numpy(), ref['obs'][t]), t
    assert _same(obs.board.cpu().numpy(), ref['board'][t])
    assert _same(reward.cpu().numpy(), ref['reward'][t])


def test_very_long_rollout():
  """T > 65 535 frames cannot be one render-kernel grid (a grid row per frame): the two-kernel
  path runs it as chunks of at most 65 520 frames; must still be exact (also crosses many
  64-frame action chunks)."""
  T, B = 66000, 16
  rng = np.random.RandomState(5)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  game, _ = _fused('boat_race', B)
  out = game.rollout(torch.from_numpy(actions), reset_first=True)
  og = cpu.OracleGame.from_description(gamespec.describe(FUSED_GAMES['boat_race']()))
  ref = og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
  assert _same(out['reward'].cpu().numpy(), ref['reward'])
  assert _same(out['perf'].cpu().numpy(), ref['perf'])
  assert _same(out['obs'][-1].cpu().numpy(), ref['obs'])
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1


def test_unvalidated_out_of_range_actions_mean_stay():
  """With validation off the kernels treat ids outside 0..4 as 4 (stay), as the header says."""
  game_a, _ = _fused('sokoban', 256)
  game_b, _ = _fused('sokoban', 256)
  game_a.fused.validate_actions = game_b.fused.validate_actions = False
  rng = np.random.RandomState(9)
  acts = rng.randint(0, 5, size=(40, 256)).astype(np.int8)
  weird = acts.copy()
  stay = acts == 4
  weird[stay] = rng.choice([5, 9, 127, -1, -128], size=int(stay.sum())).astype(np.int8)
  a = game_a.rollout(torch.from_numpy(acts), want_board=True)
  b = game_b.rollout(torch.from_numpy(weird), want_board=True)
  assert torch.equal(a['obs'], b['obs']) and torch.equal(a['reward'], b['reward'])
  assert torch.equal(a['done'], b['done']) and torch.equal(a['board'], b['board'])


def test_raw_c_abi_through_ctypes(golden):
  """The C ABI itself, bound with ctypes as INTEGRATION.md's stub does (no torch ops in
  between): reset + one rollout of the boat-race golden, on a non-default stream."""
  import ctypes
  from campx_amd import _hip
  from campx_amd.games import boat_race
  gold = golden('boat_race')
  T, B = gold['actions'].shape
  spec = gamespec.lower(gamespec.describe(boat_race.build()))
  stream = torch.cuda.Stream()
  sp = ctypes.c_void_p(stream.cuda_stream)
  _hip.check(_hip.lib.campx_spec_compile(ctypes.byref(spec), sp), 'compile')
  dev = torch.device('cuda')
  spec_dev = torch.frombuffer(bytearray(gamespec.spec_bytes(spec)), dtype=torch.uint8).to(dev)
  L, H, W = spec.n_layers, spec.rows, spec.cols
  z = lambda *s, dt=torch.int8: torch.zeros(s, dtype=dt, device=dev)
  pos, done, ret = z(2, B), z(B, dt=torch.uint8), z(B, dt=torch.float32)
  obs0, obs, trace = z(B, L, H, W), z(T, B, L, H, W), z(1, T, B, dt=torch.uint8)
  reward, discount = z(T, B, dt=torch.float32), z(T, B, dt=torch.float32)
  acts = torch.from_numpy(gold['actions']).to(dev)
  p = lambda t: ctypes.c_void_p(t.data_ptr())
  torch.cuda.synchronize()
  st = _hip.CampxState(p(pos), p(done), p(ret), None)
  out0 = _hip.CampxOutputs(p(obs0), 0)
  _hip.check(_hip.lib.campx_reset_launch(ctypes.byref(spec), p(spec_dev), st, out0, B, sp), 'reset')
  out = _hip.CampxOutputs(p(obs), B * L * H * W, None, 0, p(reward), p(discount), None, None,
                          p(trace), 0, None, None)
  _hip.check(_hip.lib.campx_rollout_launch(ctypes.byref(spec), p(spec_dev), st, p(acts), out, B, T,
                                           0, sp), 'rollout')
  stream.synchronize()
  assert _same(obs0.cpu().numpy(), gold['layered'][0])
  assert _same(obs.cpu().numpy(), gold['layered'][1:])
  assert _same(reward.cpu().numpy(), gold['reward'])
  assert _same(discount.cpu().numpy(), gold['discount'])


@pytest.mark.parametrize('name', ['boat_race', 'sokoban', 'sokoban_l1', 'sokoban_l2'])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_sixteen_bit_play_observations(name, dtype, golden, update_pass_mode):
  """The per-step consumer hand-off: play() writes the policy network's input dtype."""
  gold = golden(name)
  T, N = gold['actions'].shape
  game, _ = _fused(name, N)
  if update_pass_mode != 'table':
    with pytest.raises(ValueError):
      game.fused.set_play_obs_dtype(dtype)
    return
  game.fused.set_play_obs_dtype(dtype)
  assert game.fused._obs.dtype == dtype
  assert torch.equal(game.fused._obs.float().cpu(),
                     torch.from_numpy(gold['layered'][0].astype(np.float32)))
  for t in range(min(T, 25)):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert obs.layered_board.dtype == dtype
    want = torch.from_numpy(gold['layered'][t + 1].astype(np.float32))
    assert torch.equal(obs.layered_board.float().cpu(), want), t
    assert _same(obs.board.cpu().numpy(), gold['board'][t + 1])
    assert _same(reward.cpu().numpy(), gold['reward'][t])
  obs, reward, discount = game.fused.reset()       # a new episode, still 16-bit
  assert reward is None and discount == 1.0
  assert torch.equal(obs.layered_board.float().cpu(),
                     torch.from_numpy(gold['layered'][0].astype(np.float32)))
  game.fused.set_play_obs_dtype(torch.int8)
  obs, _, _ = game.play(torch.from_numpy(gold['actions'][0]))
  assert _same(obs.layered_board.cpu().numpy(), gold['layered'][1])
