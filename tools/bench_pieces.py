#!/usr/bin/env python3
"""What PIECES of the scenery cost the render kernel (round 6): the coin field of
examples/coins_batched.py (6x10, three coins in one drape, no switch) with its coins as pieces in a
16-bit mask per state, against the same board with coins that stay (a FixedDrape) - through the
one-cell tier and, forced, from its state table (the same kernels but for kMask) - and down the
other roads pieces can take (variants of the scenery; a tracked thing each).

    python tools/bench_pieces.py [batch]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'examples'))
sys.path.insert(0, os.path.join(REPO, 'tools'))

import coins_batched as ex  # noqa: E402
from bench_variants import timed  # noqa: E402
from campx import things  # noqa: E402
from campx.ascii_art import ascii_art_to_game  # noqa: E402
from campx_amd import gamespec, tabulate, wide  # noqa: E402


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
  art = [row.replace('s', ' ').replace('.', ' ') for row in ex.ART]

  def fixed(**where):      # the same board, coins that nobody can take
    return ascii_art_to_game(art, what_lies_beneath=' ',
                             drapes={'A': ex.Walker, 'o': things.FixedDrape, '#': things.FixedDrape,
                                     'E': things.FixedDrape}, z_order='oEA#', update_schedule='A#oE', **where)

  class Forced(object):
    def __init__(self, engine):
      self.fused = wide.WideGame(engine, B, 'cuda', tabulate.trace(fixed()))

    def its_showtime(self):
      self.fused.showtime()

  def road(pieces, variants):
    def make():
      gamespec.WIDE_MAX_PIECES, gamespec.WIDE_MAX_VARIANTS = pieces, variants
      return ex.make_game(floor=False, batch=B, device='cuda')
    return make

  rows = (('coins that stay', lambda: fixed(batch=B, device='cuda')),
          ('... from the state table', lambda: Forced(fixed())),
          ('pieces in a mask', road(16, 256)),
          ('variants of the scenery', road(0, 256)),
          ('a tracked thing each', road(0, 1)))
  for name, make in rows:
    game = make()
    ms, tbs, tier, planes = timed(game, B)
    spec = getattr(game.fused, 'spec', None)
    print('%-26s B=%d  %.4f ms per 100-frame rollout  %.2f TB/s of observations  (%s, %s trace planes, %d things, %d pieces, %d variants)' % (
        name, B, ms, tbs, tier, planes, game.fused.n_dyn, getattr(spec, 'n_pieces', 0), max(1, getattr(spec, 'n_variants', 1))))


if __name__ == '__main__':
  main()
