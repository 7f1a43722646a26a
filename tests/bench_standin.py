"""CPU stand-in for the batched engine, for tests of bench.py's launcher only.

`bench.py --standin bench_standin:make` builds this instead of the HIP engine: the
CPU oracle does the rollouts, so that the multi-rank launch path of bench.py
(self-spawned torch.distributed.run children, gloo process group, sharded action
streams, ReturnLog all-gather, max-over-ranks timing, one JSON line from rank 0)
can be exercised without a GPU.  Test infrastructure: nothing under campx_amd/ or
bench.py's measured path imports it.
"""

import numpy as np
import torch

from campx_amd import games, gamespec
from oracle import cpu as oracle_cpu


class _Fused(object):
  def __init__(self, game_name, batch):
    desc = gamespec.describe(getattr(games, game_name).build())
    oracle_cpu.set_threads(1)
    self._og = oracle_cpu.OracleGame.from_description(desc)
    self.batch = batch
    self.n_dyn, self.uses_table, self.any_reward = 1, True, True
    self.validate_actions = True
    self.ret = torch.zeros(batch)

  def rollout_buffers(self, T, share=None):
    return {}

  def rollout(self, actions, out=None, reset_first=False, pipelined=False):
    ref = self._og.rollout(actions.numpy(), reset_first=reset_first, keep_obs=False,
                           want_board=False)
    reward = torch.from_numpy(ref['reward'])
    self.ret.copy_(reward.sum(0))       # what the kernel accumulates into `ret`
    out = {} if out is None else out
    out['reward'] = reward
    return out


class _Game(object):
  def __init__(self, game_name, batch):
    self.fused = _Fused(game_name, batch)

  def its_showtime(self):
    return None


def make(game_name, batch):
  return _Game(game_name, batch)
