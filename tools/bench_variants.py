#!/usr/bin/env python3
"""What a scenery of several VARIANTS costs the render kernel (round 6): the tide game of
tests/random_pickups.py (a 6x8 board whose whole floor turns; two pictures) against the same
board with the plain Backdrop, both from their state tables, B environments, T = 100.

    python tools/bench_variants.py [batch]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch  # noqa: E402

import random_pickups  # noqa: E402
import traced_games as tg  # noqa: E402


def timed(game, B, T=100, n=30):
  game.its_showtime()
  f = game.fused
  f.validate_actions = False
  acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
  out = f.rollout_buffers(T)
  for _ in range(10):
    f.rollout(acts, out=out, reset_first=True)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    f.rollout(acts, out=out, reset_first=True)
  e1.record()
  e1.synchronize()
  ms = e0.elapsed_time(e1) / n
  row = f.n_layers * f.rows * f.cols
  return ms, row * B * T / (ms / 1e3) / 1e12, type(f).__name__, getattr(f, '_n_planes', None)


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
  d = random_pickups.definitions()[14]
  tide = random_pickups.builder(d)
  art = d['art']

  def plain(**where):      # the same board and things, a Backdrop that stays as it is
    drapes = {'A': tg.Forager, '#': tg.things.FixedDrape, 's': tg.things.FixedDrape, 'E': tg.things.FixedDrape}
    return tg.ascii_art_to_game(art, what_lies_beneath=' ', drapes=drapes, z_order='sEA#',
                                update_schedule='A#sE', **where)
  class Forced(object):          # the plain game through the STATE-table tier too: the same kernels but for kVar
    def __init__(self, engine):
      from campx_amd import tabulate, wide
      traced = tabulate.trace(plain())
      self.fused = wide.WideGame(engine, B, 'cuda', traced)

    def its_showtime(self):
      self.fused.showtime()

  for name, game in (('plain backdrop', plain(batch=B, device='cuda')),
                     ('plain, state table', Forced(plain())),
                     ('tide: 2 variants', tide(batch=B, device='cuda'))):
    ms, tbs, tier, planes = timed(game, B)
    print('%-18s B=%d  %.4f ms per 100-frame rollout  %.2f TB/s of observations  (%s, %s trace planes)' % (
        name, B, ms, tbs, tier, planes))


if __name__ == '__main__':
  main()
