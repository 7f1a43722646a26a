#!/usr/bin/env python3
"""A user's own Hello World for 32 768 environments: plain Python classes, recognised, not declared.

The game of examples/Hello World Example.ipynb (cells 3-4: the 95-cell '@' banner that rolls
under actions 0..3 and pays a point for it, four sprites that slide diagonally, action 4 quits;
integer actions; a zero-argument `make_game()`) written here independently of the notebook's
code - `Scroller` rolls with `torch.roll` by a (rows, cols) pair per action, `Bishop` derives
its four diagonal moves from the heading of action 0.  Nothing below tells the engine what
those classes do: `engine.set_default_batch()` makes `make_game()` build a batched engine, and
`its_showtime()` recognises the game (campx_amd/recognise.py: per-action offsets inferred on the
single-environment tier, then PROVED thing by thing - every reachable position x every action
with recording stand-ins for everything but the thing itself) and hands it to the shape tier's
kernel.  The recognised spec is byte-equal to the one the notebook's own cells recognise to
(tests/test_recognise.py, tests/golden/hello_world_spec.npz).

    python examples/hello_world_batched.py          # needs an MI355X
"""

import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from campx import things                                  # noqa: E402  (the alias package)
from campx.ascii_art import ascii_art_to_game, Partial   # noqa: E402
from campx_amd import engine                              # noqa: E402
from campx_amd.games.hello_world import HELLO_ART         # noqa: E402  (the notebook's art)

QUIT = 4


class Scroller(things.Drape):
  """The banner: actions 0..3 scroll it one cell up / down / left / right, wrapping, for a
  point; QUIT ends the episode."""

  SCROLL = {0: (-1, 0), 1: (1, 0), 2: (0, -1), 3: (0, 1)}     # action -> (rows, cols)

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    if actions == QUIT:
      the_plot.terminate_episode()
      return
    self.curtain.copy_(torch.roll(self.curtain, self.SCROLL[int(actions)], (0, 1)))
    the_plot.add_reward(1)


class Bishop(things.Sprite):
  """Slides diagonally: action 0 along `heading` (rows, cols), action 1 back, actions 2 and 3
  along the other diagonal (rows mirrored, and back)."""

  def __init__(self, corner, position, character, heading):
    super(Bishop, self).__init__(corner, position, character)
    dr, dc = heading
    self._moves = ((dr, dc), (-dr, -dc), (-dr, dc), (dr, -dc))

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None or actions == QUIT:
      return
    dr, dc = self._moves[int(actions)]
    rows, cols = self.corner
    self._position = self.Position((self.position.row + dr) % rows, (self.position.col + dc) % cols)


def make_game(bishop=Bishop, scroller=Scroller):
  return ascii_art_to_game(
      HELLO_ART, what_lies_beneath=' ',
      sprites={'1': Partial(bishop, (-1, -1)), '2': Partial(bishop, (1, -1)),
               '3': Partial(bishop, (1, 1)), '4': Partial(bishop, (-1, 1))},
      drapes={'@': scroller}, z_order='12@34')


def main():
  B, T = 32768, 100
  engine.set_default_batch(B, 'cuda')
  game = make_game()                      # a zero-argument make_game(), as the notebook calls its own
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
