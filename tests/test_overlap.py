"""The overlapped small-batch rollout (csrc/k_update.hip overlap_table_kernel: update pass and
observation render as two roles of one persistent launch, the render following the update pass
group by group through agent-scope progress flags).  An A/B path - measured slower than the two
launches in round 4 and off unless CAMPX_OVERLAP=1 - kept bit-exact here: the producer / consumer
hand-off is checked on every byte of every frame against the C oracle, under uneven load (batch
sizes that leave the last update workgroup partial, several launches with the state carried
over, the flat board rendered behind it)."""

import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

_CODE = r'''
import sys
sys.path.insert(0, %(repo)r)
import numpy as np, torch
from campx_amd import gamespec
from campx_amd.games import boat_race, wall_world
from oracle import cpu
for build, batches in ((boat_race.build, (16, 1024, 4096, 5008)), (wall_world.build, (64, 2000))):
  for B in batches:
    game = build(batch=B, device='cuda')
    game.its_showtime()
    assert game.fused._overlap_ctl is not None
    og = cpu.OracleGame.from_description(gamespec.describe(build()))
    rng = np.random.RandomState(B)
    for launch, T in enumerate([100, 33, 64]):
      actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
      out = game.rollout(torch.from_numpy(actions), want_board=(launch != 1))
      ref = og.rollout(actions, reset_first=(launch == 0))
      assert np.array_equal(out['obs'].cpu().numpy(), ref['obs']), (B, launch)
      if launch != 1:
        assert np.array_equal(out['board'].cpu().numpy(), ref['board']), (B, launch)
      for k in ('reward', 'discount'):
        assert np.array_equal(out[k].cpu().numpy().view(np.uint32), ref[k].view(np.uint32)), (B, launch, k)
      assert np.array_equal(out['done'].cpu().numpy(), ref['done'])
    ctl = game.fused._overlap_ctl.cpu().numpy()
    assert (ctl[:4 + (B + 255) // 256] == 0).all()        # the last workgroup out reset the block
print('ok')
''' % dict(repo=REPO)


def test_overlapped_rollouts_are_bit_exact():
  env = dict(os.environ, CAMPX_OVERLAP='1')
  out = subprocess.run([sys.executable, '-c', _CODE], env=env, capture_output=True, text=True,
                       timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  assert out.stdout.strip().endswith('ok')
