"""Mazes on boards above 128 cells (16x16 by default): the wide tier's library game.

Build-authored; PyColab-sized (the reference's `Engine(rows, cols)` sets no limit,
campx/engine.py:31, its own games stop at 5x5).  The agent is the rule library's
`AgentDrape` - Demo 3's semantics: blocked by '#', +1 on entering a '*' tile
(`Demo 3` cell 3) - and the goal is sokoban's `GoalDrape` (-1 per frame, +50 and the end
of the episode on 'G').  '#', '*', 'o' are static drapes; z-order and schedule as in the
wall world.

`maze_art(rows, cols)` draws the board deterministically: a border, a wall across every
fourth row with two doors, '*' and 'o' tiles sprinkled by a fixed formula, 'A' top left
and 'G' bottom right.
"""

from .. import rules
from ..ascii_art import ascii_art_to_game, Partial


def maze_art(rows=16, cols=16):
  if rows < 6 or cols < 6:
    raise ValueError('maze_art: at least 6x6')
  art = [[' '] * cols for _ in range(rows)]
  for r in range(rows):
    for c in range(cols):
      if r in (0, rows - 1) or c in (0, cols - 1):
        art[r][c] = '#'
      elif r % 4 == 0 and r < rows - 2:
        doors = (1 + (3 * r) % (cols - 2), 1 + (7 * r + cols // 2) % (cols - 2))
        art[r][c] = ' ' if c in doors else '#'
      elif (7 * r + 3 * c) % 11 == 0:
        art[r][c] = '*'
      elif (5 * r + c) % 13 == 0:
        art[r][c] = 'o'
  art[1][1] = 'A'
  art[rows - 2][cols - 2] = 'G'
  return [''.join(row) for row in art]


def build_with(to_game, Partial, agent_cls, goal_cls, fixed_cls, rows=16, cols=16, **engine_kwargs):
  """The game on any engine binding (tests/golden/make_golden.py runs it on the
  reference's)."""
  return to_game(
      maze_art(rows, cols), what_lies_beneath=' ',
      drapes={'A': Partial(agent_cls, blocking_chars='#', step_reward=0, reward_chars='*'),
              'G': Partial(goal_cls, agent_char='A', step_reward=-1, goal_reward=50),
              '#': fixed_cls, '*': fixed_cls, 'o': fixed_cls},
      z_order='*oGA#', update_schedule='AG*o#', **engine_kwargs)


def build(rows=16, cols=16, batch=None, device=None):
  return build_with(ascii_art_to_game, Partial, rules.AgentDrape, rules.GoalDrape,
                    rules.FixedDrape, rows, cols, batch=batch, device=device)


def make_game(rows=16, cols=16, batch=None, device=None):
  game = build(rows, cols, batch, device)
  board, reward, discount = game.its_showtime()
  return game, board, reward, discount
