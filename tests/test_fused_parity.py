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
  saved = fused.SPLIT_ROLLOUT, fused.FORCE_SPLIT
  fused.SPLIT_ROLLOUT = fused.FORCE_SPLIT = request.param == 'split'
  yield request.param
  fused.SPLIT_ROLLOUT, fused.FORCE_SPLIT = saved


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def _fused(name, batch):
  from campx_amd import fused
  game = FUSED_GAMES[name](batch=batch, device='cuda')
  first = game.its_showtime()
  tabulated = game.fused.n_dyn <= 2      # (cell, action) or (cell, cell, action) tables
  assert game.fused.uses_table == (fused.COMPILE_TABLE and tabulated)
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
  else:
    assert out['perf'] is None


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


@pytest.mark.parametrize('name', ['boat_race', 'wall_world', 'sokoban', 'demo3'])
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
  game.rollout(acts, reset_first=True)
  with pytest.raises(ValueError, match='1 action ids'):
    game.fused.check_actions()
  game.fused.validate_actions = False           # never looks
  game.rollout(acts, reset_first=True)
  game.fused.check_actions()


def test_keep_obs_false_leaves_last_frame(golden):
  gold = golden('boat_race')
  game, _ = _fused('boat_race', gold['actions'].shape[1])
  out = game.rollout(torch.from_numpy(gold['actions']), keep_obs=False)
  assert _same(out['obs'].cpu().numpy(), gold['layered'][-1])


def test_reset_first_starts_new_episode(golden):
  gold = golden('sokoban')
  game, _ = _fused('sokoban', gold['actions'].shape[1])
  acts = torch.from_numpy(gold['actions'])
  a = game.rollout(acts, reset_first=True)
  b = game.rollout(acts, reset_first=True)
  assert torch.equal(a['obs'], b['obs']) and torch.equal(a['reward'], b['reward'])
  # the running return restarted with the episode
  assert _same(game.fused.ret.cpu().numpy(), b['reward'].sum(0).cpu().numpy())


@pytest.mark.parametrize('name,batch,T', [('boat_race', 65536, 100),
                                          ('wall_world', 262144, 20),
                                          ('sokoban', 131072, 50)])
def test_full_size_vs_oracle(name, batch, T):
  """BASELINE.json batch sizes, every frame's reward/discount/done and a sample of
  full observations against the oracle; plus a size-independent invariant: every
  cell of every environment shows exactly one character."""
  rng = np.random.RandomState(1234)
  actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
  game, _ = _fused(name, batch)
  out = game.rollout(torch.from_numpy(actions), reset_first=True)
  og = cpu.OracleGame.from_description(gamespec.describe(FUSED_GAMES[name]()))
  ref = og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
  assert _same(out['discount'].cpu().numpy(), ref['discount'])
  assert _same(out['done'].cpu().numpy(), ref['done'])
  assert _same(out['reward'].cpu().numpy(), ref['reward'])
  if ref['perf'] is not None:
    assert _same(out['perf'].cpu().numpy(), ref['perf'])
  assert _same(out['obs'][-1].cpu().numpy(), ref['obs'])
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1


def test_return_gatherer_on_gpu_single_rank_rccl(tmp_path):
  """The RCCL all-gather of episode returns (bench.py's N>1 path) with one rank."""
  import subprocess
  import sys
  import os
  code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
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


def test_very_long_rollout_falls_back_to_the_fused_kernel():
  """T > 65 535 frames cannot be a render-kernel grid dimension: the launch takes the
  single-kernel path and must still be exact (also crosses many 64-frame action chunks)."""
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
