#!/usr/bin/env python3
"""Random exploration of the boat race as DEFERRED rollouts: episodes whose actions do not wait
for observations (here: drawn at random up front), so the update pass of episode i+1 can share a
launch with the render pass of episode i.

What the reference does per environment and frame - `game.play(action)` (campx/engine.py:114-166)
returning an observation (campx/engine.py:286-324) - is one `rollout_deferred()` call per
100-frame episode of every environment here; the call hands back the PREVIOUS episode's buffers,
whose observations it has just completed, and `flush()` the last one's.

    python examples/random_rollouts_deferred.py        # needs an MI355X
"""

import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from campx_amd.games import boat_race                     # noqa: E402


def run(batch=16384, frames=100, episodes=50, device='cuda', consume=None):
  """`consume(out)`: called with each episode's complete buffers (obs [T, B, L, H, W] int8,
  reward / discount / done [T, B]); default: count the cells the boat shows on."""
  game = boat_race.build(batch=batch, device=device)
  game.its_showtime()
  first = game.rollout_buffers(frames)
  sets = [first, game.rollout_buffers(frames, share=first)]      # two sets, one observation buffer
  seen = []
  if consume is None:
    consume = lambda out: seen.append(out['obs'].sum(dtype=torch.int64))
  actions = torch.randint(0, 5, (episodes, frames, batch), dtype=torch.int8, device=device)
  returns = torch.zeros(batch, device=device)
  torch.cuda.synchronize(device)
  t0 = time.perf_counter()
  for e in range(episodes):
    done = game.rollout_deferred(actions[e], sets[e & 1], reset_first=True)
    returns += sets[e & 1]['reward'].sum(0)          # this episode's scalars are ready now ...
    if done is not None:
      consume(done)                                  # ... the previous one's observations too
  consume(game.flush())
  torch.cuda.synchronize(device)
  dt = time.perf_counter() - t0
  return dict(game=game, rate=batch * frames * episodes / dt, mean_return=float(returns.mean()) / episodes,
              seen=[int(s) for s in seen])


if __name__ == '__main__':
  got = run()
  print('%.3g env-steps/s, mean episode return %.2f' % (got['rate'], got['mean_return']))
