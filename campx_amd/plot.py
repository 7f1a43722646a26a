"""The Plot: a per-game blackboard shared by the engine and all entities.

Mirrors `campx/plot.py:29-317` of the reference.  A `Plot` is a `dict` (games
store their own keys in it, e.g. `the_plot['prev_pos_A']`) that also carries the
requests entities make of the engine during one frame:

* `add_reward(r)`           rewards are summed in call order as `r + total`
                            (plot.py:208-211); nobody calling it means `None`.
* `terminate_episode(d)`    latches game-over, reports discount `d` (plot.py:179-184).
* `change_z_order(a, b)`    queue "paint a just in front of b" (plot.py:154-159).
* `change_default_discount` sets the discount reported for this frame.  NOTE the
                            reference re-creates its directives with discount 1.0
                            after every frame (plot.py:297, 100), so unlike
                            upstream PyColab the new default does not persist; that
                            behaviour is kept.
* `log(msg)`                append to the plot's message list (the reference
                            delegates to pycolab.protocols.logging, plot.py:230).
"""


class _Directives(object):
  """What the entities asked the engine to do this frame (plot.py:65-100)."""

  __slots__ = ('z_updates', 'summed_reward', 'game_over', 'discount')

  def __init__(self):
    self.z_updates = []
    self.summed_reward = None
    self.game_over = False
    self.discount = 1.0


def _require_single_character(character):
  try:
    ord(character)
  except TypeError:
    raise ValueError(
        '{} was used as an argument in a call to change_z_order, but only '
        'single ASCII characters are valid arguments'.format(repr(character)))


def _require_unit_interval(discount):
  if not 0.0 <= discount <= 1.0:
    raise ValueError('Pcontinue must be in range [0,1]')


class Plot(dict):
  """Blackboard + engine directives for one game."""

  LOG_KEY = 'log_messages'

  def __init__(self):
    super(Plot, self).__init__()
    self._frame = -1            # becomes 0 on the priming frame (plot.py:109)
    self._update_group = None
    self._engine_directives = _Directives()

  # -- requests entities can make ------------------------------------------

  def change_z_order(self, move_this, in_front_of_that):
    _require_single_character(move_this)
    if in_front_of_that is not None:
      _require_single_character(in_front_of_that)
    self._engine_directives.z_updates.append((move_this, in_front_of_that))

  def terminate_episode(self, discount=0.0):
    _require_unit_interval(discount)
    self._engine_directives.game_over = True
    self._engine_directives.discount = discount

  def add_reward(self, reward):
    total = self._engine_directives.summed_reward
    self._engine_directives.summed_reward = (
        reward if total is None else reward + total)

  def change_default_discount(self, discount):
    _require_unit_interval(discount)
    self._engine_directives.discount = discount

  def log(self, message):
    self.setdefault(self.LOG_KEY, []).append(message)

  def consume_log(self):
    """Return and clear the messages `log()` collected (build addition)."""
    return self.pop(self.LOG_KEY, [])

  # -- statistics the engine publishes --------------------------------------

  @property
  def frame(self):
    return self._frame

  @frame.setter
  def frame(self, val):
    assert val == self._frame + 1   # frames advance one at a time (plot.py:279)
    self._frame = val

  def _advance_frame(self):
    """The engine's `the_plot.frame += 1` (campx/engine.py:182).  Kept apart from the property
    so that the tabulator's probe can tell a GAME reading the frame number from the engine
    advancing it (campx_amd/tabulate.py)."""
    self._frame += 1

  @property
  def update_group(self):
    return self._update_group

  @update_group.setter
  def update_group(self, group):
    self._update_group = group

  @property
  def default_discount(self):
    return self._engine_directives.discount

  # -- engine side -----------------------------------------------------------

  def _get_engine_directives(self):
    return self._engine_directives

  def _clear_engine_directives(self):
    self._engine_directives = _Directives()
