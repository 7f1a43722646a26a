"""Run by tests/test_update_workgroups.py in a child process; the library setting big_wgs=1 makes
the library pick its 512-environment update workgroups from B = 512 up (it reads the knob
once per process): ragged batches around that size against the oracle, bit for bit."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from campx_amd import _hip, gamespec     # noqa: E402
from oracle import cpu                    # noqa: E402
from games_under_test import FUSED_GAMES  # noqa: E402


def same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def main():
  _hip.config_set('big_wgs', 1)        # the 512-environment workgroups from one workgroup up
  rng = np.random.RandomState(7)
  for name in ('boat_race', 'wall_world', 'sokoban', 'demo3'):
    og = cpu.OracleGame.from_description(gamespec.describe(FUSED_GAMES[name]()))
    for batch in (512, 777, 1537):
      game = FUSED_GAMES[name](batch=batch, device='cuda')
      game.its_showtime()
      assert game.fused.uses_table
      for launch, T in enumerate((1, 17, 100)):
        actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
        out = game.rollout(torch.from_numpy(actions), want_board=True)
        ref = og.rollout(actions, reset_first=(launch == 0))
        for key in ('obs', 'board', 'discount', 'done'):
          assert same(out[key].cpu().numpy(), ref[key]), (name, batch, T, key)
        if out['reward'] is not None:
          assert same(out['reward'].cpu().numpy(), ref['reward']), (name, batch, T)
        if ref.get('perf') is not None:
          assert same(out['perf'].cpu().numpy(), ref['perf']), (name, batch, T)
      print('ok', name, batch, flush=True)


if __name__ == '__main__':
  main()
