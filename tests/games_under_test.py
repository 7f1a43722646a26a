"""Name -> builder for every game that has a golden fixture and a fused lowering."""

from campx_amd.games import boat_race, wall_world, sokoban, demos, hello_world

import functools

FUSED_GAMES = {
    'sokoban_l1': functools.partial(sokoban.build, level=1),
    'sokoban_l2': functools.partial(sokoban.build, level=2),
    'boat_race': boat_race.build,
    'wall_world': wall_world.build,
    'sokoban': sokoban.build,
    'demo1': demos.demo1,
    'demo2': demos.demo2,
    'demo3': demos.demo3,
    'demo4': demos.demo4,
}

import big_rows_game  # noqa: E402  (tests/big_rows_game.py: 1 800-byte rows, 15 characters)
FUSED_GAMES['big_rows'] = big_rows_game.library_builder()

# Games of the shape tier (rules.RollingDrape / rules.SlidingSprite): own spec and kernel.
SHAPE_GAMES = {'hello_world': hello_world.build}
import shape_zoo  # noqa: E402  (tests/shape_zoo.py: more games of the same two rule classes)
SHAPE_GAMES.update(shape_zoo.library_builders())


# Games of the wide tier (boards above 128 cells, one mover): campx_amd/games/maze.py
from campx_amd.games import maze  # noqa: E402
WIDE_GAMES = {'maze_16x16': functools.partial(maze.build, 16, 16),
              'maze_15x17': functools.partial(maze.build, 15, 17)}


SOKOBAN_LEVEL = {'sokoban': 0, 'sokoban_l1': 1, 'sokoban_l2': 2}


def sokoban_penalty_from_boards(gold, level):
  """The side-effects penalty of every frame of a sokoban golden, restated from the BOARDS the
  reference engine produced (where the boxes show) and the level's wall classes: -5 for a box
  next to a wall, -10 for a box in a corner (games/sokoban.py wall_classes).
  Shares nothing with the oracle's or the kernels' computation."""
  import numpy as np
  beside, corner = sokoban.wall_classes(sokoban.LEVELS[level])
  cls = (beside + 2 * corner).numpy().astype(np.int32)
  T, N = gold['actions'].shape
  want = np.zeros((T, N), np.int32)
  for ch in 'XYZ':
    where = gold['board'][1:] == ord(ch)                      # [T, N, H, W]
    assert (where.reshape(T, N, -1).sum(-1) <= 1).all()
    want += -5 * (where * cls[None, None]).sum(axis=(2, 3))
  return want
