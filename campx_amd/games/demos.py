"""The reference's Demo 1-4 notebook games expressed with library rules.

Art is the notebooks' (`Demo 1` cell 2 etc.); each builder cites the cell whose
`make_game` it restates.
"""

import torch

from .. import rules
from ..ascii_art import ascii_art_to_game, Partial

STAR_ART = ['#####',
            '#A* #',
            '#*#*#',
            '# * #',
            '#####']

ARROW_ART = ['#####',
             '#A> #',
             '#^#v#',
             '# < #',
             '#####']


def demo1(batch=None, device=None):
  """Free-roaming agent, +1 per step, walls are backdrop (`Demo 1` cell 4)."""
  return ascii_art_to_game(
      STAR_ART, what_lies_beneath=' ',
      drapes={'A': Partial(rules.AgentDrape, blocking_chars='', step_reward=1)},
      z_order='A', batch=batch, device=device)


def demo2(batch=None, device=None):
  """Agent blocked by the '#' drape, +1 per step (`Demo 2` cell 4)."""
  return ascii_art_to_game(
      STAR_ART, what_lies_beneath=' ',
      drapes={'A': Partial(rules.AgentDrape, blocking_chars='#', step_reward=1),
              '#': rules.FixedDrape},
      z_order='A#', batch=batch, device=device)


def demo3(batch=None, device=None):
  """+1 on entering a '*' tile (`Demo 3` cell 4)."""
  return ascii_art_to_game(
      STAR_ART, what_lies_beneath=' ',
      drapes={'A': Partial(rules.AgentDrape, blocking_chars='#',
                           step_reward=0, reward_chars='*'),
              '#': rules.FixedDrape,
              '*': rules.FixedDrape},
      z_order='*A#', batch=batch, device=device)


def demo4(batch=None, device=None):
  """+1 on entering an arrow tile along its direction (`Demo 4` cell 4)."""
  unit = {'^': [0, 0, 1, 0, 0], '>': [0, 1, 0, 0, 0],
          'v': [0, 0, 0, 1, 0], '<': [1, 0, 0, 0, 0]}
  drapes = {'A': rules.AgentDrape, '#': rules.FixedDrape}
  for ch, d in unit.items():
    drapes[ch] = Partial(rules.DirectionalHoverRewardDrape,
                         dctns=torch.tensor(d, dtype=torch.float32),
                         base_reward=0)
  return ascii_art_to_game(ARROW_ART, what_lies_beneath=' ', drapes=drapes,
                           z_order='^>v<A#', update_schedule='A^>v<#',
                           batch=batch, device=device)
