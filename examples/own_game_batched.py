#!/usr/bin/env python3
"""A game written the way CampX / PyColab games are written - plain Python `update()`
methods on `Drape` / `Sprite` subclasses, nothing from the rule library - run for B
environments on the MI355X.

The vault: a walker 'A', a key 'k', a door 'D' the key opens, a gem '$' behind the door,
on a 12x16 board.  Four things come and go, the board has more than 128 cells: the engine
tabulates the classes on the host (every reachable state, by running them) and the wide
tier's kernels walk the resulting state table (NOTES.md 1a, 3.9).  The classes are
ordinary: they also run, unchanged, on the single-environment generic tier and on the
reference's own engine.

    python examples/own_game_batched.py --batch 65536 --frames 100

Prints what was tabulated and the rollout rate; smoke-tested in tests/test_example.py.
"""

import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from campx import things  # noqa: E402  (`campx` is this repository's alias package)
from campx.ascii_art import ascii_art_to_game  # noqa: E402

DELTA = [(0, -1), (0, 1), (-1, 0), (1, 0), (0, 0)]      # left, right, up, down, stay

ART = ['################',
       '#A    k #     $#',
       '#       #      #',
       '#       #      #',
       '#       #      #',
       '#       #      #',
       '#       D      #',
       '#       #      #',
       '#       #      #',
       '#       #      #',
       '#       #      #',
       '################']


def action_id(actions):
  return int(np.argmax(np.asarray(actions.tolist() if torch.is_tensor(actions) else actions)))


class Walker(things.Drape):
  """One cell per frame; walls and the closed door stop it; -0.25 per frame."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = DELTA[action_id(actions)]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    if not all_things['#'].curtain[r + dr, c + dc] and not all_things['D'].curtain[r + dr, c + dc]:
      self.curtain.zero_()
      self.curtain[r + dr, c + dc] = 1
    the_plot.add_reward(-0.25)


class Key(things.Drape):
  """Picked up (+1) when the walker stands on it: its curtain is empty from then on."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is not None and (self.curtain * all_things['A'].curtain).sum():
      self.curtain.zero_()
      the_plot.add_reward(1.0)


class Door(things.Drape):
  """Opens (vanishes, +0.5) once the key is gone."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is not None and self.curtain.sum() and not all_things['k'].curtain.sum():
      self.curtain.zero_()
      the_plot.add_reward(0.5)


class Gem(things.Sprite):
  """Shows itself only while the door is open; reaching it pays +10 and ends the episode."""

  def __init__(self, corner, position, character):
    super(Gem, self).__init__(corner, position, character)
    self._visible = False

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    on_it = bool(all_things['A'].curtain[self.position.row, self.position.col])
    self._visible = (not all_things['D'].curtain.sum()) and not on_it
    if on_it:
      the_plot.add_reward(10.0)
      the_plot.terminate_episode()


def make_game(**where):
  return ascii_art_to_game(
      ART, what_lies_beneath=' ', sprites={'$': Gem},
      drapes={'A': Walker, 'k': Key, 'D': Door, '#': things.FixedDrape},
      z_order='k$DA#', update_schedule='AkD$#', **where)


def run(batch=65536, frames=100, launches=10, device='cuda'):
  t0 = time.perf_counter()
  game = make_game(batch=batch, device=device)
  first, reward, discount = game.its_showtime()          # tabulates, uploads, first frame
  setup = time.perf_counter() - t0
  f = game.fused
  print('{}: {} reachable states over {} frames of Python, {} things tracked ({}), set-up {:.1f} s'
        .format(type(f).__name__, f.traced.n_states, f.traced.n_plays, len(f.traced.movers),
                ''.join(f.traced.movers), setup))
  actions = torch.randint(0, 5, (frames, batch), dtype=torch.int8, device=device)
  bufs = f.rollout_buffers(frames)
  out = game.rollout(actions, out=bufs, reset_first=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(launches):
    out = game.rollout(actions, out=bufs, reset_first=True)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / launches
  rate = batch * frames / dt
  print('{} environments x {} frames per launch: {:.3f} ms, {:.3e} env-steps/s; '
        'mean return {:.2f}, episodes ended {}'.format(
            batch, frames, dt * 1e3, rate, float(out['reward'].nan_to_num().sum(0).mean()),
            int(out['done'].sum())))
  return dict(game=game, out=out, rate=rate)


if __name__ == '__main__':
  p = argparse.ArgumentParser()
  p.add_argument('--batch', type=int, default=65536)
  p.add_argument('--frames', type=int, default=100)
  args = p.parse_args()
  run(args.batch, args.frames)
