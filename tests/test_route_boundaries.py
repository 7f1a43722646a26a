"""Batch sizes on both sides of every bound at which `rollout()` / `rollout_deferred()` change the
kernels they launch - one launch for both passes up to 8 192 environments (four movers: from 4 097),
the deferred shared launch up to 32 768 (one mover), 16 384 (two), 8 192 (three, four), the two-stream
form between those and 32 768, the 512-environment update workgroups - each side against the C
oracle, in order and deferred, with a rollout length that is not a multiple of the kernels' frame
groups and state carried over three calls."""

import numpy as np
import pytest
import torch

from campx_amd import gamespec
from campx_amd.games import boat_race, sokoban
from oracle import cpu

pytestmark = pytest.mark.gpu

CASES = []
for name, build, kw, bounds in (('boat_race', boat_race.build, {}, (8192, 32768)),
                                ('sokoban', sokoban.build, {}, (8192, 16384, 32768)),
                                ('sokoban_l1', sokoban.build, dict(level=1), (8192, 32768)),
                                ('sokoban_l2', sokoban.build, dict(level=2), (4096, 8192, 32768))):
  for b in bounds:
    for B in (b - 15, b, b + 1, b + 16):
      CASES.append(pytest.param(build, kw, B, id='{}-{}'.format(name, B)))


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


@pytest.mark.parametrize('build,kw,B', CASES)
def test_both_sides_of_a_bound(build, kw, B):
  T = 21
  rng = np.random.RandomState(B)
  streams = [rng.randint(0, 5, size=(T, B)).astype(np.int8) for _ in range(3)]
  og = cpu.OracleGame.from_description(gamespec.describe(build(**kw)))
  refs = [og.rollout(a, reset_first=(i == 0)) for i, a in enumerate(streams)]
  game = build(batch=B, device='cuda', **kw)
  game.its_showtime()
  for i, a in enumerate(streams):
    out = game.rollout(torch.from_numpy(a), reset_first=(i == 0))
    assert np.array_equal(out['obs'].cpu().numpy(), refs[i]['obs']), i
    for k in ('reward', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), refs[i][k]), (i, k)
  twin = build(batch=B, device='cuda', **kw)
  twin.its_showtime()
  first = twin.rollout_buffers(T)
  bufs = [first, twin.rollout_buffers(T, share=first)]
  for i, a in enumerate(streams):
    prev = twin.rollout_deferred(torch.from_numpy(a).cuda(), bufs[i & 1], reset_first=(i == 0))
    assert _same(bufs[i & 1]['reward'].cpu().numpy(), refs[i]['reward']), i
    if prev is not None:
      assert np.array_equal(prev['obs'].cpu().numpy(), refs[i - 1]['obs']), i
  assert np.array_equal(twin.flush()['obs'].cpu().numpy(), refs[-1]['obs'])
