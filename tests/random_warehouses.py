"""A seeded family of side-effects-sokoban levels (campx_amd/games/sokoban.py's rules: an agent,
one to three boxes that block each other, a goal that ends the episode)
on random boards of 5x5 to 8x10 with random interior walls, box and goal positions, box order in the
z-order and in their update group.

`definitions()` is a pure function of the seed.  The box and goal rules are this repo's (the
reference has no sokoban source); what tests/golden/make_random_golden.py pins is everything they
run ON: the reference's engine, renderer and Plot, with the reference's own AgentDrape
(examples/boat_race.py:28-59) as the agent - two update groups, one repaint per group
(campx/engine.py:168-208), boxes read through the rendered layers.  tests/test_random_warehouses.py
holds the generic tier, the rule lowering (pair / tuple tables, built on the device) and the
multi-mover kernels to those frames."""

import numpy as np

N_GAMES = 16
SEED = 77020261


def _one(rng):
  while True:
    H, W = int(rng.randint(5, 9)), int(rng.randint(5, 11))
    grid = np.full((H, W), ' ', dtype='<U1')
    grid[0, :] = grid[-1, :] = grid[:, 0] = grid[:, -1] = '#'
    inner = [(r, c) for r in range(1, H - 1) for c in range(1, W - 1)]
    for (r, c) in inner:
      if rng.rand() < 0.08:
        grid[r, c] = '#'
    free = [(r, c) for (r, c) in inner if grid[r, c] == ' ']
    n_boxes = int(rng.randint(1, 4))
    if len(free) < n_boxes + 6:
      continue
    rng.shuffle(free)
    for ch in ['A', 'G'] + list('XYZ'[:n_boxes]):
      r, c = free.pop()
      if ch == 'X':                                  # the first box beside the agent where there is room
        ar, ac = [int(v) for v in np.argwhere(grid == 'A')[0]]
        beside = [p for p in free + [(r, c)] if abs(p[0] - ar) + abs(p[1] - ac) == 1]
        if beside:
          free.append((r, c))
          r, c = beside[int(rng.randint(len(beside)))]
          free.remove((r, c))
      grid[r, c] = ch
    boxes = list('XYZ'[:n_boxes])
    rng.shuffle(boxes)
    group = list(boxes)
    rng.shuffle(group)
    rest = ['A', 'G', '#']
    rng.shuffle(rest)
    return dict(art=[''.join(row) for row in grid], boxes=''.join(sorted(boxes)),
                z_order='G' + ''.join(boxes) + 'A#', schedule=[group, rest])


def definitions():
  rng = np.random.RandomState(SEED)
  return [_one(rng) for _ in range(N_GAMES)]


def build(d, to_game, partial, agent, box, goal, fixed, **engine_kwargs):
  boxes = d['boxes']
  drapes = {'#': fixed, 'A': partial(agent, blocking_chars='#' + boxes),
            'G': partial(goal, agent_char='A', step_reward=-1, goal_reward=50)}
  for ch in boxes:
    drapes[ch] = partial(box, agent_char='A', blocking_chars='#' + ''.join(b for b in boxes if b != ch))
  return to_game(d['art'], what_lies_beneath=' ', drapes=drapes, update_schedule=d['schedule'],
                 z_order=d['z_order'], **engine_kwargs)


def library_builder(d):
  from campx.ascii_art import ascii_art_to_game, Partial
  from campx_amd import rules

  def make(**where):
    return build(d, ascii_art_to_game, Partial, rules.AgentDrape, rules.BoxDrape, rules.GoalDrape,
                 rules.FixedDrape, **where)
  return make
