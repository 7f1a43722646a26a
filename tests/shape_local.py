"""A game of rigidly moving multi-cell things written HERE, with plain Python update() bodies -
nothing from `campx_amd.rules`, nothing that declares its offsets.  A batched engine has to
RECOGNISE it (campx_amd/recognise.py) to run it on the shape tier.

`bind(things)` makes the classes on the given entity bindings, so tests/golden/make_golden.py
can run the very same code on the REFERENCE engine (`parade.npz`).  Written against the
reference's entity API only (campx/things.py:161-392); actions are integers 0..4, like the
Hello World notebook's.
"""

import numpy as np
import torch

PARADE_ART = ['##############',
              '#  h         #',
              '#    WW      #',
              '#    W    b  #',
              '#    W       #',
              '#        ZZ  #',
              '#         Z  #',
              '#            #',
              '##############']


def bind(things):

  class Wave(things.Drape):
    """An L of four cells; every action moves it differently (cell by cell, in Python) and
    pays 0.25 - or, standing still on action 4, 0.5."""

    _STEP = [(-1, 0), (2, 0), (0, -1), (0, 3), (0, 0)]

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      dr, dc = self._STEP[int(actions)]
      old = self.curtain.numpy().copy()
      H, W = old.shape
      new = np.zeros_like(old)
      for r in range(H):
        for c in range(W):
          if old[r, c]:
            new[(r + dr) % H, (c + dc) % W] = 1
      self.curtain.set_(torch.from_numpy(new))
      the_plot.add_reward(0.5 if int(actions) == 4 else 0.25)

  class Zigzag(things.Drape):
    """Three cells that make knight's moves; action 3 costs 1.0, action 4 ends the episode."""

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      a = int(actions)
      if a == 4:
        the_plot.terminate_episode()
        return
      dr, dc = ((1, 2), (-1, -2), (2, -1), (-2, 1))[a]
      rows, cols = np.nonzero(self.curtain.numpy())
      new = torch.zeros_like(self.curtain)
      for r, c in zip(rows, cols):
        new[(r + dr) % new.shape[0], (c + dc) % new.shape[1]] = 1
      self.curtain.set_(new)
      if a == 3:
        the_plot.add_reward(-1.0)

  class Bouncer(things.Sprite):
    """Slides one cell to the right on every action but the last."""

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None or int(actions) == 4:
        return
      self._position = self.Position(self._position.row,
                                     (self._position.col + 1) % self.corner.col)

  class Hiker(things.Sprite):
    """Up on action 0, down on action 1.  Painted BEHIND every drape, so on the reference's
    renderer it writes itself into the backdrop: a trail (campx/rendering.py:128,150)."""

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None or int(actions) > 1:
        return
      step = -1 if int(actions) == 0 else 1
      self._position = self.Position((self._position.row + step) % self.corner.row,
                                     self._position.col)

  class Loner(things.Drape):
    """NOT a shape: it stops at walls (what it does depends on where it is)."""

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      rolled = torch.roll(self.curtain, 1, 1)
      if not (rolled * all_things['#'].curtain).sum():
        self.curtain.set_(rolled)

  class Clamper(things.Sprite):
    """NOT a shape either, and only an edge shows it: moves like a shape in the open, but
    stops at the board's last column instead of wrapping."""

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None or int(actions) != 1:
        return
      col = min(self._position.col + 1, self.corner.col - 1)
      self._position = self.Position(self._position.row, col)

  import types
  return types.SimpleNamespace(Wave=Wave, Zigzag=Zigzag, Bouncer=Bouncer, Hiker=Hiker,
                               Loner=Loner, Clamper=Clamper)


def build(to_game, things, **engine_kwargs):
  """The parade on the given bindings (`to_game` = ascii_art_to_game)."""
  C = bind(things)
  return to_game(PARADE_ART, what_lies_beneath=' ',
                 sprites={'b': C.Bouncer, 'h': C.Hiker},
                 drapes={'W': C.Wave, 'Z': C.Zigzag, '#': things.FixedDrape},
                 z_order='h#WbZ', update_schedule='ZbW#h', **engine_kwargs)


def parade(**where):
  from campx import things
  from campx.ascii_art import ascii_art_to_game
  return build(ascii_art_to_game, things, **where)


def clamps_at_the_edge(**where):
  """36 columns: no random walk of the recogniser gets a sprite from column 2 to the edge."""
  from campx import things
  from campx.ascii_art import ascii_art_to_game
  C = bind(things)
  art = [' ' * 36, '  c' + ' ' * 33, ' ' * 18 + 'WW' + ' ' * 16, ' ' * 36]
  return ascii_art_to_game(art, what_lies_beneath=' ', sprites={'c': C.Clamper},
                           drapes={'W': C.Wave}, z_order='Wc', update_schedule='cW', **where)


def not_a_shape(**where):
  from campx import things
  from campx.ascii_art import ascii_art_to_game
  C = bind(things)
  return ascii_art_to_game(['######', '#LL  #', '######'], what_lies_beneath=' ',
                           drapes={'L': C.Loner, '#': things.FixedDrape}, z_order='L#',
                           update_schedule='L#', **where)
