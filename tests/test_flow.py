"""One-launch rollouts of one-mover games at small batches (csrc/k_update.hip
pipe_table_kernel<true>: update workgroups first, render workgroups of the SAME rollout behind
them, reading a tagged 16-bit copy of the trace as the update role writes it).  The default path
of `rollout()` up to 8 192 environments, so every parity test of such games already runs it; here:
that it IS the path taken (the profiler sees one kernel), the same bytes with it switched off
(CAMPX_NO_FLOW=1, a subprocess: the knob is read once), rollouts of changing length and state
carried over (the tagged copy is re-zeroed when T changes, tags wrap after 255 launches), and a
rollout captured into a HIP graph (must NOT take it: a replay would reuse the launch's tag)."""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from campx_amd import gamespec
from campx_amd.games import boat_race, wall_world
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


@pytest.mark.parametrize('build,B', [(boat_race.build, 16), (boat_race.build, 1024),
                                     (boat_race.build, 5008), (boat_race.build, 8192),
                                     (wall_world.build, 2000)])
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
from campx_amd import gamespec
from campx_amd.games import boat_race
from oracle import cpu
from torch.profiler import ProfilerActivity, profile
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
  env = dict(os.environ, CAMPX_NO_FLOW='1')
  out = subprocess.run([sys.executable, '-c', _CODE], env=env, capture_output=True, text=True,
                       timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  assert out.stdout.strip().endswith('ok')


@pytest.mark.parametrize('B', [4096, 8192])
def test_back_to_back_launches_beside_a_busy_stream(B):
  """tools/flow_stress.py: launches queued without a pause, every one compared on the device with
  the two kernels of a twin engine, while another stream fills memory and rolls out a third game."""
  out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'flow_stress.py'), str(B), '3000', '100', '1'],
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  assert out.stdout.strip().endswith('ok B=%d launches=3000' % B)
