"""A seeded family of coin-field games of Demo 3's kind (examples/Demo 3: Hover Reward Example.ipynb
cell 3): an agent drape that is blocked by some characters and rewarded when it ENTERS a cell
that shows another (its reward reads the rendered, occluded layers through the Plot) - on random
boards with one or two kinds of reward tiles, random z-orders (tiles in front of the agent or
behind it), random update schedules and, sometimes, a reward tile that also blocks.

`definitions()` is a pure function of the seed; tests/golden/make_random_golden.py builds each
game from the NOTEBOOK's own AgentDrape (cell 3 exec'd from the .ipynb) on the reference's
engine, with the plain one-hot lists the notebook passes to `play()`."""

import numpy as np

N_GAMES = 16
SEED = 31020261


def _one(rng):
  H, W = int(rng.randint(4, 9)), int(rng.randint(4, 12))
  grid = np.full((H, W), ' ', dtype='<U1')
  grid[0, :] = grid[-1, :] = grid[:, 0] = grid[:, -1] = '#'
  inner = [(r, c) for r in range(1, H - 1) for c in range(1, W - 1)]
  for (r, c) in inner:
    if rng.rand() < 0.1:
      grid[r, c] = '#'
  free = [(r, c) for (r, c) in inner if grid[r, c] == ' ']
  rng.shuffle(free)
  r, c = free.pop()
  grid[r, c] = 'A'
  tiles = '*+'[:int(rng.randint(1, 3))]
  for ch in tiles:
    for _ in range(int(rng.randint(1, max(2, len(free) // 3)))):
      if free:
        r, c = free.pop()
        grid[r, c] = ch
  tiles = ''.join(ch for ch in tiles if (grid == ch).any())
  order = list(tiles + 'A#')
  rng.shuffle(order)
  schedule = list(tiles + 'A#')
  rng.shuffle(schedule)
  blocking = '#' + (tiles[-1] if len(tiles) > 1 and rng.rand() < 0.35 else '')
  rewarding = ''.join(ch for ch in tiles if ch not in blocking) or tiles[:1]
  return dict(art=[''.join(row) for row in grid], tiles=tiles, z_order=''.join(order),
              schedule=''.join(schedule), blocking=blocking, rewarding=rewarding)


def definitions():
  rng = np.random.RandomState(SEED)
  return [_one(rng) for _ in range(N_GAMES)]


def build(d, to_game, agent, fixed, **engine_kwargs):
  drapes = {'A': agent, '#': fixed}
  for ch in d['tiles']:
    drapes[ch] = fixed
  return to_game(d['art'], what_lies_beneath=' ', drapes=drapes, z_order=d['z_order'],
                 update_schedule=d['schedule'], **engine_kwargs)


def library_builder(d, rebound=False):
  from campx import things
  from campx.ascii_art import ascii_art_to_game, Partial
  from campx_amd import rules
  R = rules.bind(things) if rebound else rules

  def make(**where):
    agent = Partial(R.AgentDrape, blocking_chars=d['blocking'], step_reward=0, reward_chars=d['rewarding'])
    return build(d, ascii_art_to_game, agent, R.FixedDrape, **where)
  return make
