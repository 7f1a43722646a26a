"""A seeded family of games with drapes of SEVERAL cells that come and go (round 6): fields of
coins taken one by one, coins that all come back when the last is gone, ice that breaks behind
the walker - tests/traced_games.py's `Coins`, `ReturningCoins`, `ThinIce` with its `Forager` - on
random boards, two to seven such cells a game, some with an exit that ends the episode; and
(`lamps`) a BACKDROP that changes: `Lamps`, a `Backdrop.update()` that flips floor lamps between
':' and '*' as the walker steps on them (campx/things.py:103-148); and (`tide`, `seasons`) one that
changes ALL OVER - `Tide` / `Seasons`: a switch turns the whole floor, two or three pictures of
dozens of cells, which the tabulator tracks as VARIANTS of the scenery.  The
reference's `Drape` sets no one-cell limit (campx/things.py:161-262); until round 6 the batched
tiers did, and these games ran on the generic tier only.  Now the tabulator describes such cells
as pieces of the scenery (tests/test_random_pickups.py says how they reach the kernels), and the
state-table kernels run them - against the REFERENCE's
engine, renderer and Plot (tests/golden/random_pickups.npz, make_random_golden.py pickups: this
very file imported where `campx` is the reference)."""

import numpy as np

import traced_games as tg

N_GAMES = 16
SEED = 61020261


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


def _one(rng, kind, n_cells):
  grid, free = _board(rng, (4, 6), (7, 10), 0.06)
  _place(grid, free, 'A', 1)
  if kind == 'lamps':
    _place(grid, free, ':', n_cells - 1)
    _place(grid, free, '*', 1)                   # (one is on from the start: both characters in the palette)
  elif kind in ('tide', 'seasons'):
    _place(grid, free, 's', n_cells)             # switches; and one floor cell of every other picture's
    for ch in ('.' if kind == 'tide' else '.:'):  # character, so that the palette has them all
      _place(grid, free, ch, 1)
  else:
    _place(grid, free, {'coins': 'o', 'returning': 'o', 'ice': '~'}[kind], n_cells)
  if kind != 'returning' and rng.rand() < 0.6:
    _place(grid, free, 'E', 1)
  return dict(kind=kind, art=[''.join(row) for row in grid])


def definitions():
  rng = np.random.RandomState(SEED)
  plan = [('coins', 2), ('returning', 2), ('ice', 4), ('coins', 7), ('returning', 3), ('ice', 6),
          ('coins', 4), ('returning', 2), ('ice', 3), ('lamps', 2), ('lamps', 4), ('lamps', 3),
          ('tide', 2), ('seasons', 1), ('tide', 3)]
  games = [_one(rng, kind, n) for kind, n in plan]
  # ... and one by hand: NINE tiles of ice in two rows - more pieces than a byte has bits (the
  # state-table tier hands pieces to the render kernel as a 16-bit mask), 1 097 reachable states
  games.append(dict(kind='ice', art=['#######', '#A~~~~#', '#~~~~~#', '#######']))
  return games


def builder(d):
  """`d` built from tests/traced_games.py's classes on whatever `campx` that module imported."""
  art, things, to_game = d['art'], tg.things, tg.ascii_art_to_game
  has_exit = any('E' in row for row in art)

  def make(**where):
    drapes = {'A': tg.Forager, '#': things.FixedDrape}
    if has_exit:
      drapes['E'] = things.FixedDrape
    if d['kind'] in ('tide', 'seasons'):
      drapes['s'] = things.FixedDrape
      return to_game(art, what_lies_beneath=' ', drapes=drapes,
                     backdrop=tg.Tide if d['kind'] == 'tide' else tg.Seasons,
                     z_order='s' + 'E' * has_exit + 'A#', update_schedule='A#s' + 'E' * has_exit, **where)
    if d['kind'] == 'lamps':
      return to_game(art, what_lies_beneath=' ', drapes=drapes, backdrop=tg.Lamps,
                     z_order='E' * has_exit + 'A#', update_schedule='A#' + 'E' * has_exit, **where)
    if d['kind'] == 'ice':
      drapes['~'] = tg.ThinIce
      z, schedule = 'E' * has_exit + 'A~#', 'A~#' + 'E' * has_exit
    else:
      drapes['o'] = tg.Coins if d['kind'] == 'coins' else tg.ReturningCoins
      z, schedule = 'E' * has_exit + 'oA#', 'Ao#' + 'E' * has_exit
    return to_game(art, what_lies_beneath=' ', drapes=drapes, z_order=z, update_schedule=schedule, **where)
  return make
