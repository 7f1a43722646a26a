"""A seeded family of games of the Hello World kind (examples/Hello World Example.ipynb cells 3-4):
one or two rolling drapes of random shapes, one to five diagonally sliding sprites with random
direction sets, a random backdrop of '#' patches, on boards from 5x9 to 14x40, in random z-orders
(sprites painted before the first drape write into the backdrop - trails, campx/rendering.py:128,150)
and update schedules.

`definitions()` is a pure function of the seed: tests/golden/make_random_golden.py builds every
game from the NOTEBOOK's own RollingDrape / SlidingSprite (cell 3, exec'd from the .ipynb where it
lies) on the reference's engine and stores what it did (tests/golden/random_hellos.npz);
tests/test_random_hellos.py builds the same definitions with this repo's classes."""

import numpy as np

N_GAMES = 20
SEED = 5020261003


def _blob(rng, H, W, h, w, fill):
  """A random shape inside an h x w box placed somewhere on the board: [(r, c), ...]."""
  r0, c0 = int(rng.randint(0, H - h + 1)), int(rng.randint(0, W - w + 1))
  cells = [(r0 + r, c0 + c) for r in range(h) for c in range(w) if rng.rand() < fill]
  return cells or [(r0, c0)]


def _one(rng):
  H = int(rng.randint(5, 15))
  W = int(rng.choice([9, 12, 15, 16, 17, 20, 24, 31, 32, 36, 40]))
  grid = np.full((H, W), ' ', dtype='<U1')
  for _ in range(int(rng.randint(0, 4))):                       # backdrop patches
    for (r, c) in _blob(rng, H, W, min(H, 3), min(W, 6), 0.6):
      grid[r, c] = '#'
  drapes = '@%'[:int(rng.randint(1, 3))]
  for ch in drapes:
    for (r, c) in _blob(rng, H, W, min(H, int(rng.randint(2, 6))), min(W, int(rng.randint(3, 12))), 0.55):
      if grid[r, c] in ' #':
        grid[r, c] = ch
  drapes = ''.join(ch for ch in drapes if (grid == ch).any())
  sprites = {}
  for ch in '12345'[:int(rng.randint(1, 6))]:
    free = np.argwhere(grid == ' ')
    r, c = free[int(rng.randint(len(free)))]
    grid[r, c] = ch
    sprites[ch] = int(rng.randint(0, 4))
  order = list(drapes + ''.join(sprites))
  rng.shuffle(order)
  schedule = list(order)
  rng.shuffle(schedule)
  return dict(art=[''.join(row) for row in grid], drapes=drapes, sprites=sprites,
              z_order=''.join(order), schedule=''.join(schedule))


def definitions():
  rng = np.random.RandomState(SEED % (2 ** 32))
  return [_one(rng) for _ in range(N_GAMES)]


def build(d, to_game, partial, rolling, sliding, **engine_kwargs):
  """The game of definition `d` from the given classes (the notebook's, or this repo's)."""
  return to_game(d['art'], what_lies_beneath=' ',
                 sprites={ch: partial(sliding, k) for ch, k in d['sprites'].items()},
                 drapes={ch: rolling for ch in d['drapes']},
                 z_order=d['z_order'], update_schedule=d['schedule'], **engine_kwargs)


def library_builder(d, rebound=False):
  """`d` with this repo's rule classes on this repo's engine; `rebound`: bound afresh, so that
  the engine takes them for a user's own classes (recognised for the shape tier, not lowered)."""
  from campx import things
  from campx.ascii_art import ascii_art_to_game, Partial
  from campx_amd import rules
  R = rules.bind(things) if rebound else rules

  def make(**where):
    return build(d, ascii_art_to_game, Partial, R.RollingDrape, R.SlidingSprite, **where)
  return make
