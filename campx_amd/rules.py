"""Library rule classes: the game-logic vocabulary the fused HIP tier understands.

The reference has no rule library: every example re-types its `AgentDrape` /
reward drape as a Python class whose `update()` is a handful of tensor ops.  The
classes below are that same vocabulary, parameterised,
with two faces:

* `update()` is ordinary generic-tier Python, op-for-op equivalent to the
  reference example class it generalises (cited per class), so it runs on the
  single-environment engine - and, through `bind(<reference things module>)`,
  on the reference's own engine, which is how `tests/golden/make_golden.py`
  pins it.
* `fused_rule()` returns a small declarative description that
  `campx_amd.fused` lowers into the GameSpec consumed by the HIP kernel.

Action convention for the one-cell rules (examples/boat_race.py:26): a 5-vector
one-hot `[left, right, up, down, stay]`; "left" is column-1 and "up" is row-1,
both cyclic (boat_race.py:42-45).  The Hello World rules (`RollingDrape`,
`SlidingSprite`) take the notebook's integer actions 0..4.
"""

import types

import torch

from . import things as _things


def _shifted(mask, action_weights):
  """Blend of the four cyclic one-cell shifts of `mask` plus `mask` itself.

  `action_weights[i]` multiplies shift i in the order left, right, up, down,
  stay - the same blend as boat_race.py:42-49, written with `torch.roll`.
  """
  w = action_weights
  return ((w[0] * torch.roll(mask, -1, 1)) + (w[1] * torch.roll(mask, 1, 1)) +
          (w[2] * torch.roll(mask, -1, 0)) + (w[3] * torch.roll(mask, 1, 0)) +
          (w[4] * mask))


def _integral(actions):
  """One-hot actions as integers (tensor `.byte()` as boat_race.py:40, or list)."""
  return actions.byte() if torch.is_tensor(actions) else actions


def bind(things):
  """Build the rule classes on top of a given `things` module.

  Called once below for `campx_amd.things`; the golden generator calls it with
  the reference's `campx.things` so the very same `update()` bodies run on the
  reference engine.
  """

  class AgentDrape(things.Drape):
    """One-cell agent moved by the action, optionally blocked / rewarded.

    Generalises the reference's agent classes:
      * Demo 1 (`Demo 1` cell 3): `blocking_chars=''`, `step_reward=1`.
      * Demo 2 (`Demo 2` cell 3): `blocking_chars='#'`, `step_reward=1`.
      * Demo 3 (`Demo 3` cell 3): `+ reward_chars='*'`, `step_reward=0`: reward
        when the agent *enters* a cell showing one of those characters.
      * Boat race (examples/boat_race.py:28-59): `blocking_chars='#'`, no reward.

    A move into a cell whose *rendered* character (as of the latest repaint)
    is blocking reverts the agent to its rendered position (boat_race.py:52-56).
    """

    def __init__(self, curtain, character, blocking_chars='#',
                 step_reward=None, reward_chars=''):
      super(AgentDrape, self).__init__(curtain, character)
      self.blocking_chars = blocking_chars
      self.step_reward = step_reward
      self.reward_chars = reward_chars

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      del board, backdrop, all_things
      mine = 'prev_pos_' + self.character
      if actions is not None:
        b = _shifted(self.curtain, _integral(actions))
        for c in self.blocking_chars:
          if mine in the_plot:
            gate = (b * (1 - layers[c])).sum()          # 1 = free, 0 = blocked
            b = (gate * b) + (the_plot[mine] * (1 - gate))
        self.curtain.set_(b)
        if self.step_reward is not None or self.reward_chars:
          reward = 0 if self.step_reward is None else self.step_reward
          for c in self.reward_chars:
            if 'prev_pos_' + c in the_plot:
              reward += (b * the_plot['prev_pos_' + c]).sum()
          the_plot.add_reward(reward)
      # Live references: always "as of the latest render" (SURVEY A.3 Q1).
      the_plot[mine] = layers[self.character]
      for c in self.reward_chars:
        the_plot['prev_pos_' + c] = layers[c]

    def fused_rule(self):
      return dict(op='agent', blocking=self.blocking_chars,
                  step_reward=self.step_reward, reward_chars=self.reward_chars)

  class DirectionalHoverRewardDrape(things.Drape):
    """Static tiles paying `base_reward + dctns[action]` when the agent enters.

    Restates examples/boat_race.py:61-91 (`base_reward=-0.25`, the default) and
    `Demo 4` cell 3 (`base_reward=0`).  The gate multiplies the agent's *new*
    curtain with this drape's *rendered, occluded* layer, so it fires only on
    the step the agent arrives.  The reference looks the agent up
    as `all_things['A']` whatever `agent_chars` says; here `agent_chars` is
    honoured, which is identical for the default 'A'.
    """

    def __init__(self, curtain, character, agent_chars='A', dctns=None,
                 base_reward=-0.25):
      super(DirectionalHoverRewardDrape, self).__init__(curtain, character)
      self.agent_chars = agent_chars
      self.d = dctns
      self.base_reward = base_reward

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      del board, backdrop
      mine = 'prev_pos_' + self.character
      if actions is not None:
        reward = self.base_reward
        for ac in self.agent_chars:
          if mine in the_plot:
            on_tile = (all_things[ac].curtain * the_plot[mine]).sum()
            reward += on_tile * (self.d * actions).sum()
        the_plot.add_reward(reward)
      the_plot[mine] = layers[self.character]

    def fused_rule(self):
      return dict(op='dir_hover', agents=self.agent_chars,
                  dctns=[float(x) for x in self.d],
                  base_reward=float(self.base_reward))

  class BoxDrape(things.Drape):
    """A one-cell box the agent pushes (sokoban).  Not in the reference.

    Build-authored rule: the box moves one cell in the
    action's direction when the agent's rendered position, shifted by the
    action, lands on the box and the cell beyond shows no blocking character.
    It must sit in an update group *before* the agent's so that the repaint
    between the groups shows the agent where the box now is (engine.py:208).
    """

    def __init__(self, curtain, character, agent_char='A', blocking_chars='#'):
      super(BoxDrape, self).__init__(curtain, character)
      self.agent_char = agent_char
      self.blocking_chars = blocking_chars

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      del board, backdrop, all_things, the_plot
      if actions is None:
        return
      act = _integral(actions)
      pushed_from = _shifted(layers[self.agent_char], act)
      move = (pushed_from * self.curtain).sum()
      beyond = _shifted(self.curtain, act)
      for c in self.blocking_chars:
        move = move * (beyond * (1 - layers[c])).sum()
      self.curtain.set_((move * beyond) + ((1 - move) * self.curtain))

    def fused_rule(self):
      return dict(op='box', agent=self.agent_char,
                  blocking=self.blocking_chars)

  class GoalDrape(things.Drape):
    """Static goal tiles: per-step reward, bonus + termination on arrival.

    Build-authored.  Uses the agent's current curtain and
    this drape's own (un-occluded) curtain, so it must be updated after the
    agent in the same frame.
    """

    def __init__(self, curtain, character, agent_char='A', step_reward=-1,
                 goal_reward=50):
      super(GoalDrape, self).__init__(curtain, character)
      self.agent_char = agent_char
      self.step_reward = step_reward
      self.goal_reward = goal_reward

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      del board, layers, backdrop
      if actions is None:
        return
      arrived = (all_things[self.agent_char].curtain * self.curtain).sum()
      the_plot.add_reward(self.step_reward + arrived * self.goal_reward)
      if arrived:
        the_plot.terminate_episode()

    def fused_rule(self):
      return dict(op='goal', agent=self.agent_char,
                  step_reward=float(self.step_reward),
                  goal_reward=float(self.goal_reward))

  class RollingDrape(things.Drape):
    """A multi-cell drape whose whole mask rolls cyclically with the action.

    Restates `RollingDrape` of the reference's Hello World notebook
    (examples/Hello World Example.ipynb cell 3): the action is an integer; actions
    0..3 roll the mask by `roll_shifts[a]` along axis `roll_axes[a]` and pay
    `move_reward`; `quit_action` ends the episode (and pays nothing); anything else
    is ignored.  (The notebook round-trips through numpy's `np.roll`; `torch.roll` is
    the same permutation.)
    """

    def __init__(self, curtain, character, roll_axes=(0, 0, 1, 1),
                 roll_shifts=(-1, 1, -1, 1), quit_action=4, move_reward=1):
      super(RollingDrape, self).__init__(curtain, character)
      self.roll_axes = tuple(roll_axes)
      self.roll_shifts = tuple(roll_shifts)
      self.quit_action = quit_action
      self.move_reward = move_reward

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      del board, layers, backdrop, all_things
      if actions is None:
        return
      a = int(actions)
      if a == self.quit_action:
        the_plot.terminate_episode()
      if 0 <= a < len(self.roll_axes):
        self.curtain.set_(torch.roll(self.curtain, self.roll_shifts[a],
                                     self.roll_axes[a]))
        the_plot.add_reward(self.move_reward)

    def fused_rule(self):
      n = len(self.roll_axes)
      drow = [self.roll_shifts[a] if self.roll_axes[a] == 0 else 0 for a in range(n)]
      dcol = [self.roll_shifts[a] if self.roll_axes[a] == 1 else 0 for a in range(n)]
      return dict(op='shape', drow=drow, dcol=dcol,
                  rewards=[float(self.move_reward)] * n,
                  quit_action=self.quit_action)

  class SlidingSprite(things.Sprite):
    """A sprite that slides diagonally, cyclically (Hello World notebook cell 3).

    `direction_set` in 0..3 picks one of four mappings from actions 0..3 to diagonal
    steps; other actions are ignored.
    """

    _DX = ([-1, 1, -1, 1], [-1, 1, -1, 1], [1, -1, 1, -1], [1, -1, 1, -1])
    _DY = ([-1, 1, 1, -1], [1, -1, -1, 1], [1, -1, -1, 1], [-1, 1, 1, -1])

    def __init__(self, corner, position, character, direction_set):
      super(SlidingSprite, self).__init__(corner, position, character)
      self.direction_set = direction_set
      self._dx = self._DX[direction_set]
      self._dy = self._DY[direction_set]

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      del board, layers, backdrop, all_things, the_plot
      if actions is None or int(actions) > 3:
        return
      a = int(actions)
      self._position = self.Position(
          (self._position.row + self._dy[a]) % self.corner.row,
          (self._position.col + self._dx[a]) % self.corner.col)

    def fused_rule(self):
      return dict(op='shape', drow=list(self._dy), dcol=list(self._dx),
                  rewards=[None] * 4, quit_action=None)

  return types.SimpleNamespace(
      RollingDrape=RollingDrape,
      SlidingSprite=SlidingSprite,
      AgentDrape=AgentDrape,
      DirectionalHoverRewardDrape=DirectionalHoverRewardDrape,
      BoxDrape=BoxDrape,
      GoalDrape=GoalDrape,
      FixedDrape=things.FixedDrape)


_bound = bind(_things)
AgentDrape = _bound.AgentDrape
DirectionalHoverRewardDrape = _bound.DirectionalHoverRewardDrape
BoxDrape = _bound.BoxDrape
GoalDrape = _bound.GoalDrape
FixedDrape = _bound.FixedDrape
RollingDrape = _bound.RollingDrape
SlidingSprite = _bound.SlidingSprite

FUSED_RULE_CLASSES = (AgentDrape, DirectionalHoverRewardDrape, BoxDrape,
                      GoalDrape)
# Rigidly translated things that interact with nothing: lowered to the shape tier
# (gamespec.lower_shapes, csrc shape_rollout_kernel) instead of the one-cell model.
SHAPE_RULE_CLASSES = (RollingDrape, SlidingSprite)
