#!/usr/bin/env python3
"""Collectibles and a floor that turns, for B environments on the MI355X - written as CampX /
PyColab games are written: plain Python `update()` methods, nothing from the rule library.

    'A'  a walker (one cell per frame, walls stop it; -0.125 per frame)
    'o'  coins: ONE drape of several cells; a coin the walker stands on leaves the curtain (+1)
    's'  a switch: each time the walker steps onto it the BACKDROP repaints the whole floor,
         day ' ' to night '.' and back (`Backdrop.update()`, campx/things.py:103-148); a night
         frame pays 0.25
    'E'  the exit ends the episode (+5)

The reference's Drape sets no one-cell limit and its Backdrop may repaint itself
(campx/things.py:161-262, 103-148).  A batched Engine tabulates the classes on the host by running
them over every reachable state: the walker is the one thing the kernels track cell by cell; the
floor and the coins that are left are thirteen pictures of the scenery - VARIANTS that the state
names, each environment's own laid by the render kernel.  Without the switch
(`make_game(floor=False)`: a plain Backdrop) the three coins are PIECES of the scenery: which of them
show is a 16-bit mask per state that the render kernel patches onto the one scenery row.  The
classes also run, unchanged, on the single-environment generic tier.

    python examples/coins_batched.py --batch 65536 --frames 100

Smoke-tested in tests/test_example.py.
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

ART = ['##########',
       '#A  o   .#',
       '# ##  #  #',
       '#o  s # o#',
       '#  #    E#',
       '##########']


def action_id(actions):
  return int(np.argmax(np.asarray(actions.tolist() if torch.is_tensor(actions) else actions)))


class Walker(things.Drape):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = DELTA[action_id(actions)]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    if not all_things['#'].curtain[r + dr, c + dc]:
      self.curtain.zero_()
      self.curtain[r + dr, c + dc] = 1
    the_plot.add_reward(-0.125)
    if (all_things['E'].curtain * self.curtain).sum():
      the_plot.add_reward(5.0)
      the_plot.terminate_episode()


class Coins(things.Drape):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    taken = self.curtain * all_things['A'].curtain
    if int(taken.sum()):
      self.curtain.set_(self.curtain - taken)
      the_plot.add_reward(float(taken.sum()))


class Floor(things.Backdrop):
  def update(self, actions, board, layers, things_, the_plot):
    if actions is None:
      return
    here = things_['A'].curtain              # (the Backdrop is updated first: where the last frame left the walker)
    before = the_plot.get('walker_was')
    the_plot['walker_was'] = here.clone()
    if before is not None and not bool((before == here).all()) and int((here * things_['s'].curtain).sum()):
      day, night = self.curtain == ord(' '), self.curtain == ord('.')
      self.curtain[day] = ord('.')
      self.curtain[night] = ord(' ')
    if int((self.curtain == ord('.')).sum()) > int((self.curtain == ord(' ')).sum()):
      the_plot.add_reward(0.25)


def make_game(floor=True, **where):
  if not floor:        # the coin field alone: no switch, a Backdrop that stays
    art = [row.replace('s', ' ').replace('.', ' ') for row in ART]
    return ascii_art_to_game(art, what_lies_beneath=' ',
                             drapes={'A': Walker, 'o': Coins, '#': things.FixedDrape, 'E': things.FixedDrape},
                             z_order='oEA#', update_schedule='Ao#E', **where)
  return ascii_art_to_game(ART, what_lies_beneath=' ', backdrop=Floor,
                           drapes={'A': Walker, 'o': Coins, '#': things.FixedDrape, 's': things.FixedDrape,
                                   'E': things.FixedDrape},
                           z_order='soEA#', update_schedule='Ao#sE', **where)


def run(batch=65536, frames=100, launches=10, device='cuda'):
  game = make_game(batch=batch, device=device)
  t0 = time.perf_counter()
  game.its_showtime()
  set_up = time.perf_counter() - t0
  traced = game.fused.traced
  actions = torch.randint(0, 5, (frames, batch), dtype=torch.int8, device=device)
  out = game.rollout_buffers(frames)
  game.rollout(actions, out=out, reset_first=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(launches):
    game.rollout(actions, out=out, reset_first=True)
  torch.cuda.synchronize()
  rate = batch * frames * launches / (time.perf_counter() - t0)
  return dict(game=game, out=out, rate=rate, set_up=set_up, states=traced.n_states, movers=traced.movers,
              variants=len(traced.variants))


if __name__ == '__main__':
  p = argparse.ArgumentParser()
  p.add_argument('--batch', type=int, default=65536)
  p.add_argument('--frames', type=int, default=100)
  args = p.parse_args()
  got = run(args.batch, args.frames)
  row = got['out']['obs'].shape[2] * got['out']['obs'].shape[3] * got['out']['obs'].shape[4]
  print('{} states tabulated in {:.1f} s: tracked {} + the floor ({} variants)'.format(
      got['states'], got['set_up'], ''.join(got['movers']), got['variants']))
  print('{:.3g} env-steps/s ({:.2f} TB/s of observations), mean return {:.2f}'.format(
      got['rate'], got['rate'] * row / 1e12, float(got['out']['reward'].sum(0).mean())))
