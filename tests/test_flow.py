"""One-launch rollouts of table games at small batches (csrc/k_update.hip
pipe_table_kernel<true>, and pipe_multi_kernel<K, ., true> for games of two to four movers:
update workgroups first, render workgroups of the SAME rollout behind
them, reading a tagged 16-bit copy of the trace as the update role writes it).  The default path
of `rollout()` up to 8 192 environments, so every parity test of such games already runs it; here:
that it IS the path taken (the profiler sees one kernel), the same bytes with it switched off
(the library setting flow=0), rollouts of changing length and state
carried over (the tagged copy is re-zeroed when T changes, tags wrap after 255 launches), and a
rollout captured into a HIP graph (must NOT take it: a replay would reuse the launch's tag)."""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from campx_amd import gamespec
from campx_amd.games import boat_race, sokoban, wall_world
from conftest import REPO
from oracle import cpu

pytestmark = pytest.mark.gpu


def _kernels_of(fn):
  from torch.profiler import ProfilerActivity, profile
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fn()
    torch.cuda.synchronize()
  return [e.key for e in prof.key_averages() if 'campx_impl' in e.key]


def test_small_batch_rollouts_take_one_launch():
  game = boat_race.build(batch=4096, device='cuda')
  game.its_showtime()
  acts = torch.randint(0, 5, (50, 4096), dtype=torch.int8, device='cuda')
  out = game.fused.rollout_buffers(50)
  game.rollout(acts, out=out)
  names = _kernels_of(lambda: game.rollout(acts, out=out))
  assert len(names) == 1 and 'pipe_table_kernel' in names[0], names
  # ... and past 8 192 environments, two
  big = boat_race.build(batch=16384, device='cuda')
  big.its_showtime()
  acts = torch.randint(0, 5, (50, 16384), dtype=torch.int8, device='cuda')
  out = big.fused.rollout_buffers(50)
  big.rollout(acts, out=out)
  names = _kernels_of(lambda: big.rollout(acts, out=out))
  assert sorted(n.split('<')[0].split('::')[-1] for n in names) == ['render_kernel', 'update_table_kernel'], names


def _sokoban(level):
  def build(**kw):
    return sokoban.build(level=level, **kw)
  return build


def test_small_batch_rollouts_of_two_to_four_movers_take_one_launch():
  """Round 5: the same one launch for sokoban's levels (two / three / four movers) - the four-mover
  game from 4 097 environments up only (below, its update role's chain of loads from the 212 MB
  tuple table is the longer part and runs faster alone)."""
  for level, B, want in ((0, 2048, 1), (0, 8192, 1), (0, 16384, 2), (1, 1024, 1), (2, 1024, 2), (2, 8192, 1)):
    game = sokoban.build(level=level, batch=B, device='cuda')
    game.its_showtime()
    acts = torch.randint(0, 5, (30, B), dtype=torch.int8, device='cuda')
    out = game.fused.rollout_buffers(30)
    game.rollout(acts, out=out)
    names = _kernels_of(lambda: game.rollout(acts, out=out))
    assert len(names) == want, (level, B, names)
    assert ('pipe_multi_kernel' in names[0]) == (want == 1), (level, B, names)
    assert game.fused._one_launch(30, out['trace'].stride(1)) == (want == 1)


@pytest.mark.parametrize('build,B', [(boat_race.build, 16), (boat_race.build, 1024),
                                     (boat_race.build, 5008), (boat_race.build, 8192),
                                     (wall_world.build, 2000), (_sokoban(0), 16), (_sokoban(0), 1000),
                                     (_sokoban(0), 8192), (_sokoban(1), 2064), (_sokoban(2), 5008)])
def test_many_launches_of_changing_length_match_the_oracle(build, B):
  game = build(batch=B, device='cuda')
  game.its_showtime()
  og = cpu.OracleGame.from_description(gamespec.describe(build()))
  rng = np.random.RandomState(B)
  for launch, T in enumerate([40, 40, 17, 40, 1, 64, 64]):
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=(launch == 3), reset_first=(launch == 0))
    ref = og.rollout(actions, reset_first=(launch == 0))
    assert np.array_equal(out['obs'].cpu().numpy(), ref['obs']), (B, launch)
    if launch == 3:
      assert np.array_equal(out['board'].cpu().numpy(), ref['board']), (B, launch)
    for k in ('reward', 'discount'):
      assert np.array_equal(out[k].cpu().numpy().view(np.uint32), ref[k].view(np.uint32)), (B, launch, k)
    assert np.array_equal(out['done'].cpu().numpy(), ref['done'])


def test_tags_wrap_after_255_launches():
  B, T = 256, 20
  game = boat_race.build(batch=B, device='cuda')
  game.its_showtime()
  og = cpu.OracleGame.from_description(gamespec.describe(boat_race.build()))
  rng = np.random.RandomState(1)
  out = game.fused.rollout_buffers(T)
  for launch in range(600):
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    game.rollout(torch.from_numpy(actions), out=out, reset_first=(launch == 0))
    ref = og.rollout(actions, reset_first=(launch == 0))
    if launch % 50 == 0 or 250 <= launch <= 262 or 505 <= launch <= 515:
      assert np.array_equal(out['obs'].cpu().numpy(), ref['obs']), launch
  assert np.array_equal(out['obs'].cpu().numpy(), ref['obs'])


def test_a_captured_rollout_takes_two_launches_and_replays_right():
  B, T = 1024, 30
  a, b = (boat_race.build(batch=B, device='cuda') for _ in range(2))
  for g in (a, b):
    g.its_showtime()
    g.fused.validate_actions = False
  acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
  out_a, out_b = a.fused.rollout_buffers(T), b.fused.rollout_buffers(T)
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    a.rollout(acts, out=out_a)                 # warm up outside the capture
  torch.cuda.current_stream().wait_stream(side)
  b.rollout(acts, out=out_b)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    a.rollout(acts, out=out_a)
  for _ in range(3):
    graph.replay()
    b.rollout(acts, out=out_b)
    torch.cuda.synchronize()
    assert torch.equal(out_a['obs'], out_b['obs'])
    assert torch.equal(out_a['reward'], out_b['reward'])
  assert torch.equal(a.fused.pos, b.fused.pos)


_CODE = r'''
import sys
sys.path.insert(0, %(repo)r)
import numpy as np, torch
from campx_amd import _hip, gamespec
from campx_amd.games import boat_race
from oracle import cpu
from torch.profiler import ProfilerActivity, profile
_hip.config_set('flow', 0)          # never the one-launch rollout: update pass, then render
B = 4096
game = boat_race.build(batch=B, device='cuda')
game.its_showtime()
og = cpu.OracleGame.from_description(gamespec.describe(boat_race.build()))
rng = np.random.RandomState(4)
for launch, T in enumerate([60, 33]):
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    out = game.rollout(torch.from_numpy(actions), reset_first=(launch == 0))
    torch.cuda.synchronize()
  names = [e.key for e in prof.key_averages() if 'campx_impl' in e.key]
  assert len(names) == 2 and not any('pipe_table' in n for n in names), names
  ref = og.rollout(actions, reset_first=(launch == 0))
  assert np.array_equal(out['obs'].cpu().numpy(), ref['obs'])
print('ok')
''' % dict(repo=REPO)


def test_switched_off_the_two_launches_give_the_same_bytes():
  out = subprocess.run([sys.executable, '-c', _CODE], capture_output=True, text=True,
                       timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  assert out.stdout.strip().endswith('ok')


@pytest.mark.parametrize('B,game', [(4096, 'boat_race'), (8192, 'boat_race'), (4096, 'sokoban'),
                                    (8192, 'sokoban_l1'), (8192, 'sokoban_l2')])
def test_back_to_back_launches_beside_a_busy_stream(B, game):
  """tools/flow_stress.py: launches queued without a pause, every one compared on the device with
  the two kernels of a twin engine, while another stream fills memory and rolls out a third game."""
  out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'flow_stress.py'), str(B), '3000', '100', '1',
                        game], capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  assert out.stdout.strip().endswith('ok B=%d launches=3000' % B)


# ------------------------------------------------ round 5: loud failure, caller-owned tag state

_GIVE_UP = r'''
import sys
sys.path.insert(0, %(repo)r)
import torch
from campx_amd import _hip
from campx_amd.games import boat_race, sokoban
_hip.config_set('flow_max_naps', %(naps)d)
_hip.config_set('flow_debug_delay', 3000)
B, T = 4096, 40
game = %(build)s(batch=B, device='cuda')
game.its_showtime()
game.fused.validate_actions = False          # the error word is looked at whatever this says
acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
out = game.fused.rollout_buffers(T)
assert game.fused._one_launch(T, out['trace'].stride(1))
try:
  game.rollout(acts, out=out)
  torch.cuda.synchronize()
  game.fused.check_ok()
except RuntimeError as e:
  assert 'gave up' in str(e) and 'WRONG' in str(e), str(e)
  # the word was taken: the next (healthy, two-launch: different T is still one launch) call is clean
  print('raised')
else:
  print('silent')
'''


@pytest.mark.parametrize('build', ['boat_race.build', 'sokoban.build'])
def test_a_render_wave_that_gives_up_says_so(build):
  """a render wave whose trace entries never get this launch's tag used to
  `break` and render stale bytes without a word.  Provoked here with the library's two test settings (flow_max_naps, flow_debug_delay) -
  the update role held back by a sleep, render waves allowed ONE second look - the launch must
  raise CAMPX_ERR_FLOW_TIMEOUT in the error word and `rollout()` / `check_ok()` a RuntimeError."""
  out = subprocess.run([sys.executable, '-c', _GIVE_UP % dict(repo=REPO, build=build, naps=1)],
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  assert out.stdout.strip().endswith('raised'), out.stdout[-500:]
  # ... and with the default patience the same delay is simply waited out: right frames, no error
  out = subprocess.run([sys.executable, '-c', _GIVE_UP % dict(repo=REPO, build=build, naps=1 << 20)],
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  assert out.stdout.strip().endswith('silent'), out.stdout[-500:]


def test_without_flow_state_or_error_flag_the_c_abi_runs_two_launches():
  """The one-launch path needs the caller's CampxFlowState and an error word (include/campx_hip.h):
  a CampxOutputs without either must not take it - and the library keeps no table of blocks."""
  import ctypes
  from campx_amd import _hip
  B, T = 1024, 24
  game = boat_race.build(batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  buf = f.rollout_buffers(T)
  acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
  scratch = torch.zeros((_hip.lib.campx_flow_scratch_bytes(B, T) + 3) // 4, dtype=torch.int32, device='cuda')
  state = _hip.CampxFlowState()
  flag = torch.zeros(1, dtype=torch.int32, device='cuda')
  p = lambda t: ctypes.c_void_p(t.data_ptr())

  def outputs(with_state, with_flag):
    o = _hip.CampxOutputs()
    o.obs, o.obs_t_stride = p(buf['obs']), B * f.n_layers * f.rows * f.cols
    o.reward, o.discount, o.done = p(buf['reward']), p(buf['discount']), p(buf['done'])
    o.trace, o.scalar_pitch = p(buf['trace']), buf['trace'].stride(1)
    o.overlap_ctl, o.overlap_ctl_bytes = p(scratch), scratch.numel() * 4
    if with_state:
      o.flow_state = ctypes.cast(ctypes.pointer(state), ctypes.c_void_p)
    if with_flag:
      o.error_flag = p(flag)
    return o

  st = _hip.CampxState(p(f.pos), p(f.done), p(f.ret), None)
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def launch(o):
    _hip.check(_hip.lib.campx_rollout_launch(ctypes.byref(f.spec), p(f._spec_dev), st, p(acts), o,
                                             B, T, 1, stream), 'campx_rollout_launch')
    torch.cuda.synchronize()

  for with_state, with_flag, want in ((False, True, 2), (True, False, 2), (True, True, 1)):
    o = outputs(with_state, with_flag)
    launch(o)
    names = _kernels_of(lambda: launch(o))
    assert len(names) == want, (with_state, with_flag, names)
  assert state.tag == 2 and (state.B, state.T, state.pitch) == (B, T, buf['trace'].stride(1))
  assert int(flag.item()) == 0


def test_one_block_with_changing_row_pitch_through_the_tag_wrap():
  """the re-zero test looked at (B, T) only; a C caller alternating `scalar_pitch` on
  one block left pad-column entries with old tags that the 255-tag wrap made current.  The state
  now records the pitch (and lives with the caller): 600 launches alternating two pitches on one
  block, every one compared with a two-launch twin."""
  import ctypes
  from campx_amd import _hip
  B, T = 1008, 12
  a, b = (boat_race.build(batch=B, device='cuda') for _ in range(2))
  for g in (a, b):
    g.its_showtime()
    g.fused.validate_actions = False
  f = a.fused
  p = lambda t: ctypes.c_void_p(t.data_ptr())
  pitches = (B, B + 16)
  scratch = torch.zeros((_hip.lib.campx_flow_scratch_bytes(B + 16, T) + 3) // 4, dtype=torch.int32, device='cuda')
  state = _hip.CampxFlowState()
  flag = torch.zeros(1, dtype=torch.int32, device='cuda')
  obs = torch.empty((T, B, f.n_layers, f.rows, f.cols), dtype=torch.int8, device='cuda')
  rows = {q: dict(reward=torch.empty((T, q), dtype=torch.float32, device='cuda'),
                  trace=torch.empty((1, T, q), dtype=torch.uint8, device='cuda')) for q in pitches}
  st = _hip.CampxState(p(f.pos), p(f.done), p(f.ret), None)
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  twin = b.fused.rollout_buffers(T)
  gen = torch.Generator(device='cpu').manual_seed(5)
  for launch in range(600):
    q = pitches[(launch // 3) & 1]                 # three launches per pitch, then the other
    acts = torch.randint(0, 5, (T, B), generator=gen, dtype=torch.int8).cuda()
    o = _hip.CampxOutputs()
    o.obs, o.obs_t_stride = p(obs), B * f.n_layers * f.rows * f.cols
    o.reward, o.trace, o.scalar_pitch = p(rows[q]['reward']), p(rows[q]['trace']), q
    o.overlap_ctl, o.overlap_ctl_bytes = p(scratch), scratch.numel() * 4
    o.flow_state = ctypes.cast(ctypes.pointer(state), ctypes.c_void_p)
    o.error_flag = p(flag)
    _hip.check(_hip.lib.campx_rollout_launch(ctypes.byref(f.spec), p(f._spec_dev), st, p(acts), o,
                                             B, T, int(launch == 0), stream), 'campx_rollout_launch')
    b.fused.rollout(acts, out=twin, reset_first=(launch == 0))
    if launch % 7 == 0 or 250 <= launch <= 262 or 505 <= launch <= 520:
      assert torch.equal(obs, twin['obs']), launch
      assert torch.equal(rows[q]['reward'][:, :B], twin['reward']), launch
  assert torch.equal(obs, twin['obs']) and int(flag.item()) == 0
  assert state.pitch in pitches and 1 <= state.tag <= 255
