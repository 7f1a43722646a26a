"""A seeded family of race-track games of the boat race's kind (examples/boat_race.py:15-113): an
agent drape, walls, directional hover-reward tiles - on random boards, with random tile placements
and reward vectors, random z-orders (tiles in front of the agent or behind it: the tiles' rendered
layers, which is what both classes read through the Plot, boat_race.py:55-59,80-91, differ), random
update schedules (the agent before or after the tiles, one update group or several: one repaint
per group, campx/engine.py:168-208) and tiles that also block.

`definitions()` is a pure function of the seed: tests/golden/make_random_golden.py builds every
game with the REFERENCE's own AgentDrape / DirectionalHoverRewardDrape on the reference's engine
and stores what it did (tests/golden/random_tracks.npz, which also holds each game's art and
parameters so that a drifted generator is noticed); tests/test_random_tracks.py builds the same
definitions with this repo's classes.  Nothing here is taken from the reference but the two
constructor signatures.
"""

import numpy as np

N_GAMES = 32
N_BIG = 8            # more games, on boards above 128 cells: the batched engine runs those from their state table
SEED = 20261003
TILES = '^>v<'
BONUSES = (0.0, 1.0, 3.0, -2.0, 0.5)


def _one(rng, big=False):
  H, W = (int(rng.randint(4, 9)), int(rng.randint(4, 11))) if not big else \
      (int(rng.randint(10, 17)), int(rng.randint(14, 25)))
  grid = np.full((H, W), ' ', dtype='<U1')
  grid[0, :] = grid[-1, :] = grid[:, 0] = grid[:, -1] = '#'
  inner = [(r, c) for r in range(1, H - 1) for c in range(1, W - 1)]
  for (r, c) in inner:
    if rng.rand() < 0.12:
      grid[r, c] = '#'
  free = [(r, c) for (r, c) in inner if grid[r, c] == ' ']
  rng.shuffle(free)
  r, c = free.pop()
  grid[r, c] = 'A'
  n_tiles = int(rng.randint(1, 5))
  tiles = ''.join(sorted(rng.choice(list(TILES), size=n_tiles, replace=False), key=TILES.index))
  dctns = {}
  for ch in tiles:
    for _ in range(int(rng.randint(1, 4))):
      if free:
        r, c = free.pop()
        grid[r, c] = ch
    dctns[ch] = [float(BONUSES[int(i)]) for i in rng.randint(0, len(BONUSES), size=5)]
  tiles = ''.join(ch for ch in tiles if (grid == ch).any())
  order = list(tiles + 'A#')
  rng.shuffle(order)
  schedule = list(tiles + 'A#')
  rng.shuffle(schedule)
  groups = int(rng.choice([1, 1, 2, 3]))
  cuts = sorted(rng.choice(np.arange(1, len(schedule)), size=min(groups - 1, len(schedule) - 1),
                           replace=False).tolist()) if groups > 1 else []
  grouped = [schedule[a:b] for a, b in zip([0] + cuts, cuts + [len(schedule)])]
  blocking = '#'
  if tiles and rng.rand() < 0.3:
    blocking += str(rng.choice(list(tiles)))
  return dict(art=[''.join(row) for row in grid], tiles=tiles, dctns={ch: dctns[ch] for ch in tiles},
              z_order=''.join(order), schedule=grouped, blocking=blocking)


def definitions():
  rng = np.random.RandomState(SEED)
  games = [_one(rng) for _ in range(N_GAMES)]
  return games + [_one(rng, big=True) for _ in range(N_BIG)]


def build(d, to_game, partial, agent, hover, fixed, tensor, **engine_kwargs):
  """The game of definition `d` from the given classes (the reference's, or this repo's)."""
  drapes = {'A': partial(agent, blocking_chars=d['blocking']), '#': fixed}
  for ch in d['tiles']:
    drapes[ch] = partial(hover, dctns=tensor(d['dctns'][ch]))
  schedule = d['schedule'] if len(d['schedule']) > 1 else ''.join(d['schedule'][0])
  return to_game(d['art'], what_lies_beneath=' ', drapes=drapes, z_order=d['z_order'],
                 update_schedule=schedule, **engine_kwargs)


def library_builder(d, rebound=False):
  """`d` with this repo's rule classes on this repo's engine (batch / device as keywords).
  `rebound`: the same classes bound afresh (`rules.bind`): to the engine they are then arbitrary
  Python classes - a user's own - and reach the device through the tabulator, not the rule lowering."""
  import torch
  from campx import things
  from campx.ascii_art import ascii_art_to_game, Partial
  from campx_amd import rules
  R = rules.bind(things) if rebound else rules

  def make(**where):
    return build(d, ascii_art_to_game, Partial, R.AgentDrape, R.DirectionalHoverRewardDrape,
                 R.FixedDrape, torch.FloatTensor, **where)
  return make
