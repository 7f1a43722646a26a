"""A 10 x 12 board with 15 characters (rows of 1 800 bytes) and three movers: sokoban's rules
(agent, two boxes, goal) among nine kinds of static decoration.  Large rows and many layers are
what the reference's own games never reach; `build()` takes the engine bindings as arguments so
that tests/golden/make_golden.py can run the very same game on the REFERENCE engine."""

import functools

ART = ['############',
       '#A  a b c  #',
       '# X        #',
       '#   d e f  #',
       '#     Y    #',
       '#  g h i   #',
       '#          #',
       '#  aa  ii  #',
       '#        G #',
       '############']
BOXES = 'XY'
DECOR = 'abcdefghi'


def build(to_game, Partial, agent_cls, box_cls, goal_cls, fixed_cls, **engine_kwargs):
  drapes = {'#': fixed_cls,
            'A': Partial(agent_cls, blocking_chars='#' + BOXES),
            'G': Partial(goal_cls, agent_char='A', step_reward=-1, goal_reward=50)}
  for ch in DECOR:
    drapes[ch] = fixed_cls
  for ch in BOXES:
    drapes[ch] = Partial(box_cls, agent_char='A', blocking_chars='#' + BOXES.replace(ch, ''))
  return to_game(ART, what_lies_beneath=' ', drapes=drapes,
                 update_schedule=[list(BOXES), ['A', 'G', '#'] + list(DECOR)],
                 z_order=DECOR + 'G' + BOXES + 'A#', **engine_kwargs)


def library_builder():
  from campx_amd import rules
  from campx_amd.ascii_art import ascii_art_to_game, Partial
  return functools.partial(build, ascii_art_to_game, Partial, rules.AgentDrape, rules.BoxDrape,
                           rules.GoalDrape, rules.FixedDrape)
