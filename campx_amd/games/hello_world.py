"""Hello World (examples/Hello World Example.ipynb cells 3-4) with library rules.

Four diagonal `SlidingSprite`s and the rolling '@' drape over a static '#' backdrop;
actions are the notebook's integers: 0..3 move, 4 quits.  z-order '12@34' puts sprites
1 and 2 behind the drape - which, on the reference's renderer, makes them leave
trails in the backdrop; every tier here reproduces that.
"""

from .. import rules
from ..ascii_art import ascii_art_to_game, Partial

HELLO_ART = ['                                    ',
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
             '                                    ']


def build(batch=None, device=None):
  return ascii_art_to_game(
      HELLO_ART, what_lies_beneath=' ',
      sprites={'1': Partial(rules.SlidingSprite, 0), '2': Partial(rules.SlidingSprite, 1),
               '3': Partial(rules.SlidingSprite, 2), '4': Partial(rules.SlidingSprite, 3)},
      drapes={'@': rules.RollingDrape}, z_order='12@34', batch=batch, device=device)


def make_game(batch=None, device=None):
  game = build(batch, device)
  board, reward, discount = game.its_showtime()
  return game, board, reward, discount
