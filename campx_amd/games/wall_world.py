"""10x10 wall world with four drapes (BASELINE config 3).

Build-authored scale-up of the reference's Demo 2 (`Demo 2` cells 2-4): the
agent is Demo 2's `AgentDrape` (blocked by '#', +1 per acted step); '#', '*' and
'o' are static drapes.  Board, z-order and schedule are SURVEY.md appendix A.6.
"""

from .. import rules
from ..ascii_art import ascii_art_to_game, Partial

GAME_ART = ['##########',
            '#A  *   o#',
            '# ## ## ##',
            '#  *     #',
            '## # ### #',
            '#    o   #',
            '# ###  # #',
            '#  *   # #',
            '#o    *  #',
            '##########']


def build(batch=None, device=None):
  return ascii_art_to_game(
      GAME_ART, what_lies_beneath=' ',
      drapes={'A': Partial(rules.AgentDrape, blocking_chars='#', step_reward=1),
              '#': rules.FixedDrape,
              '*': rules.FixedDrape,
              'o': rules.FixedDrape},
      z_order='*oA#', update_schedule='A*o#', batch=batch, device=device)


def make_game(batch=None, device=None):
  game = build(batch, device)
  board, reward, discount = game.its_showtime()
  return game, board, reward, discount
