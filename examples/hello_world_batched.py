#!/usr/bin/env python3
"""The Hello World notebook's game, typed the way the notebook types it, for 32 768 environments.

examples/Hello World Example.ipynb (cells 3-4) defines `RollingDrape` (np.roll of a 95-cell
mask) and `SlidingSprite` (diagonal moves) with plain Python `update()` methods and integer
actions, and a zero-argument `make_game()`.  Nothing below tells the engine what those classes
do: `engine.set_default_batch()` makes `make_game()` build a batched engine, and
`its_showtime()` recognises the game (campx_amd/recognise.py: per-action offsets inferred on the
single-environment tier and verified on sampled walks) and hands it to the shape tier's kernel.

    python examples/hello_world_batched.py          # needs an MI355X
"""

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from campx import things                                  # noqa: E402  (the alias package)
from campx.ascii_art import ascii_art_to_game, Partial   # noqa: E402
from campx_amd import engine                              # noqa: E402
from campx_amd.games.hello_world import HELLO_ART         # noqa: E402  (the notebook's art)


class RollingDrape(things.Drape):
  _ROLL_AXES = [0, 0, 1, 1]
  _ROLL_SHIFTS = [-1, 1, -1, 1]

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    if actions == 4: the_plot.terminate_episode()
    if actions < 4:
      rolled = np.roll(self.curtain.numpy(), self._ROLL_SHIFTS[actions], self._ROLL_AXES[actions])
      self.curtain.set_(torch.from_numpy(rolled.copy()))
      the_plot.add_reward(1)


class SlidingSprite(things.Sprite):
  _DX = ([-1, 1, -1, 1], [-1, 1, -1, 1], [1, -1, 1, -1], [1, -1, 1, -1])
  _DY = ([-1, 1, 1, -1], [1, -1, -1, 1], [1, -1, -1, 1], [-1, 1, 1, -1])

  def __init__(self, corner, position, character, direction_set):
    super(SlidingSprite, self).__init__(corner, position, character)
    self._dx = self._DX[direction_set]
    self._dy = self._DY[direction_set]

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None or actions > 3: return
    self._position = self.Position((self._position.row + self._dy[actions]) % self.corner.row,
                                   (self._position.col + self._dx[actions]) % self.corner.col)


def make_game():
  return ascii_art_to_game(
      HELLO_ART, what_lies_beneath=' ',
      sprites={'1': Partial(SlidingSprite, 0), '2': Partial(SlidingSprite, 1),
               '3': Partial(SlidingSprite, 2), '4': Partial(SlidingSprite, 3)},
      drapes={'@': RollingDrape}, z_order='12@34')


def main():
  B, T = 32768, 100
  engine.set_default_batch(B, 'cuda')
  game = make_game()                      # the notebook's call, unchanged
  t0 = time.perf_counter()
  board, reward, discount = game.its_showtime()
  torch.cuda.synchronize()
  print('recognised and started in {:.2f} s: {}'.format(time.perf_counter() - t0, type(game.fused).__name__))
  board, reward, discount = game.play(0)                  # cell 6: "this moves all the 64s upward"
  print('play(0): reward', float(reward[0]), 'for every environment:', bool((reward == 1).all()))
  actions = torch.randint(0, 4, (T, B), dtype=torch.int8, device='cuda')
  game.fused.validate_actions = False
  out = game.fused.rollout_buffers(T)
  game.rollout(actions, out=out)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(10):
    game.rollout(actions, out=out)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 10
  row = out['obs'].shape[2] * out['obs'].shape[3] * out['obs'].shape[4]
  print('{} environments x {} frames: {:.2f} ms per launch, {:.2e} env-steps/s, {:.2f} TB/s of observations'
        .format(B, T, dt * 1e3, B * T / dt, B * T * row / dt / 1e12))


if __name__ == '__main__':
  main()
