"""A seeded family of games of ARBITRARY Python classes (tests/traced_games.py: plain `update()`
bodies over numpy views, Python branches, Plot calls) on random boards: ice rinks (a skater that
slides many cells a frame, coins, an exit), toll roads (tiles that change the frame's discount,
an exit that ends the episode with discount 0.75), burrows (a mole that changes its place in the
z-order and is paid for being hidden) and vaults (a key and a door that leave the board, a gem
that shows itself only then).  What such games reach the device through is the tabulator, so
this family is the tabulator's - and the wide / one-cell table kernels' - against the REFERENCE's
engine, renderer and Plot (tests/golden/random_quests.npz, make_random_golden.py quests: the very
same file imported where `campx` is the reference)."""

import numpy as np

import traced_games as tg

N_GAMES = 12
SEED = 91020261


def _board(rng, lo, hi, walls):
  H, W = int(rng.randint(lo[0], hi[0])), int(rng.randint(lo[1], hi[1]))
  grid = np.full((H, W), ' ', dtype='<U1')
  grid[0, :] = grid[-1, :] = grid[:, 0] = grid[:, -1] = '#'
  inner = [(r, c) for r in range(1, H - 1) for c in range(1, W - 1)]
  for (r, c) in inner:
    if rng.rand() < walls:
      grid[r, c] = '#'
  free = [(r, c) for (r, c) in inner if grid[r, c] == ' ']
  rng.shuffle(free)
  return grid, free


def _place(grid, free, ch, n):
  for _ in range(n):
    if free:
      r, c = free.pop()
      grid[r, c] = ch


def _one(rng, kind):
  if kind == 'ice':
    grid, free = _board(rng, (5, 6), (9, 12), 0.08)
    _place(grid, free, 'A', 1)
    _place(grid, free, 'o', int(rng.randint(1, 6)))
    _place(grid, free, 'E', int(rng.randint(0, 2)))
  elif kind == 'toll':
    grid, free = _board(rng, (4, 5), (8, 11), 0.1)
    _place(grid, free, 'A', 1)
    _place(grid, free, '$', int(rng.randint(0, 4)))
    _place(grid, free, '%', int(rng.randint(0, 3)))
    _place(grid, free, 'E', int(rng.randint(0, 2)))
  elif kind == 'burrow':
    grid, free = _board(rng, (4, 6), (7, 11), 0.05)
    _place(grid, free, 'A', 1)
    _place(grid, free, 'd', int(rng.randint(1, 3)))
    _place(grid, free, 'u', int(rng.randint(1, 3)))
    _place(grid, free, '$', int(rng.randint(0, 2)))
    _place(grid, free, '=', int(rng.randint(2, 8)))
  else:
    grid, free = _board(rng, (4, 7), (6, 11), 0.05)
    _place(grid, free, 'A', 1)
    _place(grid, free, 'k', 1)
    _place(grid, free, 'D', 1)
    _place(grid, free, '$', 1)
  return dict(kind=kind, art=[''.join(row) for row in grid])


def definitions():
  rng = np.random.RandomState(SEED)
  return [_one(rng, kind) for kind in ('ice', 'toll', 'burrow', 'vault') * (N_GAMES // 4)]


def builder(d):
  """`d` built from tests/traced_games.py's classes on whatever `campx` that module imported."""
  art, things, to_game = d['art'], tg.things, tg.ascii_art_to_game

  def make(**where):
    if d['kind'] == 'ice':
      return to_game(art, what_lies_beneath=' ',
                     drapes={'A': tg.IceSkater, '#': things.FixedDrape, 'o': things.FixedDrape,
                             'E': things.FixedDrape}, z_order='oEA#', update_schedule='A#oE', **where)
    if d['kind'] == 'toll':
      return to_game(art, what_lies_beneath=' ',
                     drapes={'A': tg.TollWalker, '#': things.FixedDrape, '$': things.FixedDrape,
                             '%': things.FixedDrape, 'E': things.FixedDrape},
                     z_order='$%EA#', update_schedule='A#$%E', **where)
    if d['kind'] == 'burrow':
      return to_game(art, what_lies_beneath=' ',
                     drapes={'A': tg.Mole, '#': things.FixedDrape, '=': things.FixedDrape,
                             'd': things.FixedDrape, 'u': things.FixedDrape, '$': things.FixedDrape},
                     z_order='du$=A#', update_schedule='A#=du$', **where)
    return to_game(art, what_lies_beneath=' ', sprites={'$': tg.Gem},
                   drapes={'A': tg.VaultWalker, 'k': tg.Key, 'D': tg.Door, '#': things.FixedDrape},
                   z_order='k$DA#', update_schedule='AkD$#', **where)
  return make
