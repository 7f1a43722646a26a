"""Games written the way the reference writes its own (examples/boat_race.py:40-57): Drapes whose
`update()` is arithmetic on `[H, W]` tensors - `+ - * >=`, `torch.cat`, `sum`, `set_` - with no
Python branch on a tensor.  Nothing from `campx_amd.rules`; a batched engine has to tabulate them,
and a two-crate warehouse has far more states than one frame of Python per (state, action) can
walk (campx_amd/tabulate.py MAX_PLAYS): campx_amd/tabulate_batched.py runs these very methods on
lane tensors, many states per call.
"""

import torch

WAREHOUSE_ART = ['############',
                 '#P   #     #',
                 '# X  #  G  #',
                 '#    #     #',
                 '#          #',
                 '### #### ###',
                 '#          #',
                 '#  G #  Y  #',
                 '#    #     #',
                 '#    #     #',
                 '#    #     #',
                 '############']


def _shifted(b):
  """b moved one cell left, right, up, down, and not at all (cyclic: the walls do the rest)."""
  return [torch.cat([b[:, 1:], b[:, :1]], dim=1), torch.cat([b[:, -1:], b[:, :-1]], dim=1),
          torch.cat([b[1:], b[:1]], dim=0), torch.cat([b[-1:], b[:-1]], dim=0), b]


def _moved(act, b):
  s = _shifted(b)
  return (act[0] * s[0]) + (act[1] * s[1]) + (act[2] * s[2]) + (act[3] * s[3]) + (act[4] * s[4])


def bind(things):

  class Crate(things.Drape):
    """Pushed by the porter when he walks into it, unless a wall or another crate is behind."""

    def __init__(self, curtain, character, porter='P', solid='#', others=''):
      super(Crate, self).__init__(curtain, character)
      self.porter, self.solid, self.others = porter, solid, others

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      act = actions.byte()
      me = self.curtain
      pushed = (_moved(act, layers[self.porter]) * me).sum()          # 1: the porter walks into me
      there = _moved(act, me)
      behind = (there * layers[self.solid]).sum()
      for ch in self.others:
        behind = behind + (there * all_things[ch].curtain).sum()
      go = (pushed * (1 - (behind >= 1).long())).byte()
      self.curtain.set_((go * there) + ((1 - go) * me))

  class Porter(things.Drape):
    """Walks; stopped by walls and by crates that did not give way.  -0.25 a frame, +0.5 per
    crate that stands on a goal cell after the frame."""

    def __init__(self, curtain, character, goal, solid='#', crates='XY'):
      super(Porter, self).__init__(curtain, character)
      self.goal, self.solid, self.crates = goal, solid, crates

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      act = actions.byte()
      me = self.curtain
      there = _moved(act, me)
      hit = (there * layers[self.solid]).sum()
      reward = -0.25
      for ch in self.crates:
        hit = hit + (there * all_things[ch].curtain).sum()
        reward = reward + 0.5 * (all_things[ch].curtain * self.goal).sum().float()
      free = (1 - (hit >= 1).long()).byte()
      self.curtain.set_((free * there) + ((1 - free) * me))
      the_plot.add_reward(reward)

  import types
  return types.SimpleNamespace(Crate=Crate, Porter=Porter)


def build(to_game, things, partial, art=None, **engine_kwargs):
  art = list(art or WAREHOUSE_ART)
  goal = torch.tensor([[1 if c == 'G' else 0 for c in row] for row in art], dtype=torch.uint8)
  art = [row.replace('G', ' ') for row in art]
  C = bind(things)
  return to_game(art, what_lies_beneath=' ',
                 drapes={'X': partial(C.Crate, others='Y'), 'Y': partial(C.Crate, others='X'),
                         'P': partial(C.Porter, goal), '#': things.FixedDrape},
                 z_order='XYP#', update_schedule='XYP#', **engine_kwargs)


def warehouse(art=None, **where):
  from campx import things
  from campx.ascii_art import ascii_art_to_game, Partial
  return build(ascii_art_to_game, things, Partial, art=art, **where)


SMALL_ART = ['######',
             '#P   #',
             '# X  #',
             '# GY #',
             '######']


def small_warehouse(**where):
  """The same classes on a board the one-frame-per-play walker can finish (a few thousand frames: 12 free cells, some 900 states)."""
  return warehouse(art=SMALL_ART, **where)
