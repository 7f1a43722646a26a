"""Shape-tier games beyond Hello World, built from the same two rule classes
(rules.RollingDrape / rules.SlidingSprite) to reach what Hello World does not: more than
four things (the second offset word), boards whose cell count is odd / 8k + 4 / 8k / below 8,
a drape first in z-order (no trails) and several sprites before the first drape (trails),
drapes that cover hundreds of cells, a static FixedDrape among the things, the rewards of
several drapes summed in one frame (0.1 + 0.7 + 1.3), a quit action that also rolls and
pays.

`build(name, ...)` takes the engine bindings as arguments, so tests/golden/make_golden.py
can build the very same games on the REFERENCE engine (its ascii_art_to_game, its things)
and store what the reference does with them as golden fixtures."""

import functools

ZOO = {
    # 7x9 = 63 cells (not a multiple of 4); the drape is behind everything: no trails
    'zoo0': dict(
        art=['         ',
             ' @@   1  ',
             ' @       ',
             '    %% 2 ',
             '    %    ',
             '  3      ',
             '         '],
        sprites={'1': 0, '2': 3, '3': 1},
        drapes={'@': dict(move_reward=0.1),
                '%': dict(roll_axes=(1, 1, 0, 0), roll_shifts=(2, -3, 1, -1), move_reward=0.7,
                          quit_action=None)},
        z_order='@1%23', update_schedule='3%21@'),
    # 6x10 = 60 cells (8k + 4); eight things; sprites 1 and 2 leave trails; '%' quits on
    # action 2, which also rolls it and pays
    'zoo1': dict(
        art=['1    @@   ',
             '  2  @  5 ',
             ' %%       ',
             ' %   3 && ',
             '        & ',
             '   4      '],
        sprites={'1': 0, '2': 1, '3': 2, '4': 3, '5': 0},
        drapes={'@': dict(move_reward=0.1, quit_action=None),
                '%': dict(move_reward=0.7, quit_action=2),
                '&': dict(roll_axes=(1, 0, 1, 0), roll_shifts=(-1, -1, 1, 1), move_reward=1.3,
                          quit_action=None)},
        z_order='12@3%4&5', update_schedule='&5%4@321'),
    # 2x3 = 6 cells (below 8)
    'zoo2': dict(
        art=['@ 1',
             '   '],
        sprites={'1': 2},
        drapes={'@': dict()},
        z_order='1@', update_schedule='@1'),
    # 16x24 = 384 cells (8k); a static '#' thing; a drape of 150 cells; trails of 1 and 2
    'zoo3': dict(
        art=['########################',
             '#  1                   #',
             '#     @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#  2  @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#     @@@@@@@@@@@@@@@  #',
             '#                   3  #',
             '#                      #',
             '#                      #',
             '########################'],
        sprites={'1': 1, '2': 2, '3': 3},
        drapes={'@': dict(move_reward=2)},
        fixed='#',
        z_order='12#@3', update_schedule='@#123'),
    # Hello World's own art (13x36 = 468 cells, 3 276-byte observations), the letters a
    # static thing, the drape painted FIRST: no trails - the two-kernel path at the size the
    # shape tier is measured at
    'zoo4': dict(
        art=['                                    ',
             '  #   #  ### #    #     ###         ',
             '  #   # #    #    #    #   #        ',
             '  ##### #### #    #    #   #        ',
             '  #   # #    #    #    #   #        ',
             '  #   #  ###  ###  ###  ###         ',
             '                                    ',
             '     @   @  @@@   @@@  @    @@@@  1 ',
             '     @   @ @   @ @   @ @    @   @ 2 ',
             '     @ @ @ @   @ @@@@  @    @   @ 3 ',
             '     @ @ @ @   @ @   @ @    @   @   ',
             '      @@@   @@@  @   @  @@@ @@@@  4 ',
             '                                    '],
        sprites={'1': 0, '2': 1, '3': 2, '4': 3},
        drapes={'@': dict()},
        fixed='#',
        z_order='@#1234', update_schedule='1234@#'),
}


def build(name, to_game, Partial, rules, fixed_cls, **engine_kwargs):
  """The game `name` on the given bindings: `to_game` = ascii_art_to_game, `rules` =
  an object with RollingDrape / SlidingSprite, `fixed_cls` = FixedDrape."""
  z = ZOO[name]
  sprites = {ch: Partial(rules.SlidingSprite, d) for ch, d in z['sprites'].items()}
  drapes = {ch: Partial(rules.RollingDrape, **kw) for ch, kw in z['drapes'].items()}
  if z.get('fixed'):
    drapes[z['fixed']] = fixed_cls
  return to_game(z['art'], what_lies_beneath=' ', sprites=sprites, drapes=drapes,
                 z_order=z['z_order'], update_schedule=z['update_schedule'], **engine_kwargs)


def library_builders():
  """name -> builder(batch=None, device=None) on this repo's engine."""
  from campx_amd import rules
  from campx_amd.ascii_art import ascii_art_to_game, Partial
  return {'shape_' + name: functools.partial(build, name, ascii_art_to_game, Partial, rules,
                                             rules.FixedDrape)
          for name in ZOO}
