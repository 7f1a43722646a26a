"""The game engine: set-up API, `its_showtime()` and the `play()` loop.

Mirror of the reference's engine surface (`campx/engine.py:29-544`) with two
execution tiers behind one class:

generic tier  (`batch=None`, the default)
    One environment.  Entities are arbitrary Python `Sprite`/`Drape`/`Backdrop`
    subclasses whose `update()` is called every frame; rendering is
    `rendering.BaseObservationRenderer`.  This is the reference's own execution
    model (engine.py:114-324) and is what unmodified `examples/boat_race.py` and
    the Demo notebooks run on.  CPU torch tensors, no native code.

fused tier  (`batch=B`, or a process-wide default: `set_default_batch`, CAMPX_BATCH)
    B independent environments advanced by the HIP kernels (csrc/k_*.hip) over an
    int8 [B, L, H, W] layered board; `its_showtime()` compiles the game for them:
    * entities from `campx_amd.rules` are lowered to a GameSpec whose state tables
      the rule interpreter kernel builds on the device (`gamespec`, `fused`);
    * ANY OTHER Python classes - the unmodified `examples/boat_race.py` classes,
      the Demo notebooks' classes, a user's own - are tabulated on the host by
      running their `update()` on the generic tier over every reachable state
      (`tabulate`), and the same table + render kernels run the result;
    * games the cell-indexed tables cannot take - boards above 128 cells (up to 1 024),
      more than four tracked values (up to eight things that show) - run from their
      STATE table, one row per reachable state (`wide`, csrc/k_wide.hip);
    * Hello-World-style games of rigidly moving multi-cell things take the shape tier
      (`shapes`, csrc/k_shape.hip).
    There is no CPU fallback: without the HIP library or a GPU this tier raises.

Error behaviour kept from the reference: `RuntimeError` for `play()` before
`its_showtime()` or after game over (engine.py:146-151), for set-up calls during
play (engine.py:327-330) and for characters claimed twice (engine.py:343-350);
`ValueError`/`TypeError` for malformed set-up arguments (engine.py:47-53,
332-341, 421-426, 470-472).

Out of scope: the un-occluded renderer
(`occlusion_in_layers=False`, broken in the reference) and the PySyft
`send()`/`share()` transport.
"""

import collections
import os

import torch

from . import plot
from . import rendering
from . import things

# Process-wide default for `Engine(batch=None)`: lets set-up code written for the
# reference - a zero-argument `make_game()` (examples/boat_race.py:93-115), a notebook
# cell - build a batched engine without being edited.
_DEFAULT_BATCH = [None, None]
_UNSET = object()


def set_default_batch(batch, device=None):
  """Make every `Engine` constructed without `batch=` a batched one (None: back to the
  reference's single-environment generic tier).  The environment variables CAMPX_BATCH /
  CAMPX_DEVICE set the same default at import time."""
  if batch is not None and int(batch) < 1:
    raise ValueError('batch must be >= 1')
  _DEFAULT_BATCH[0] = None if batch is None else int(batch)
  _DEFAULT_BATCH[1] = device


def get_default_batch():
  return tuple(_DEFAULT_BATCH)


if os.environ.get('CAMPX_BATCH'):
  set_default_batch(int(os.environ['CAMPX_BATCH']), os.environ.get('CAMPX_DEVICE') or None)


class Engine(object):
  """A grid-world game: entities, their update order, their z-order."""

  def __init__(self, rows, cols, occlusion_in_layers=True, batch=_UNSET,
               device=None):
    if batch is _UNSET:      # not given: the process-wide default (normally None)
      batch = _DEFAULT_BATCH[0]
      device = _DEFAULT_BATCH[1] if device is None else device
    if not occlusion_in_layers:
      raise NotImplementedError(
          'occlusion_in_layers=False is not supported: the reference renderer '
          'for it (rendering.py:227-353) does not run, so there is nothing to '
          'be compatible with.')
    self._rows = rows
    self._cols = cols
    self._occlusion_in_layers = occlusion_in_layers
    self._backdrop = None
    self._sprites_and_drapes = collections.OrderedDict()   # also the z-order
    self._update_groups = collections.defaultdict(list)
    self._showtime = False
    self._game_over = False
    self._the_plot = plot.Plot()
    self._board = None
    self._renderer = None
    self._hidden_performance = None
    self._hidden_penalty = None
    # Fused tier.
    self._batch = batch
    self._device = device
    self._fused = None
    self._action_set = None

  # ---------------------------------------------------------------- set-up

  @property
  def rows(self):
    return self._rows

  @property
  def cols(self):
    return self._cols

  @property
  def batch(self):
    return self._batch

  @property
  def the_plot(self):
    return self._the_plot

  @property
  def things(self):
    return self._sprites_and_drapes

  @property
  def backdrop(self):
    return self._backdrop

  @property
  def z_order(self):
    return list(self._sprites_and_drapes.keys())

  @property
  def game_over(self):
    return self._game_over

  @property
  def hidden_performance(self):
    return self._hidden_performance

  def set_hidden_performance(self, agent_char, cycle):
    """Declare a hidden performance measure (build addition; set-up time only).

    `cycle` is a sequence of [H, W] 0/1 masks m_0 .. m_{n-1} (disjoint).  The
    performance of a frame is +1 when `agent_char` goes from a cell of m_i to a
    cell of m_{i+1} (cyclically), -1 for m_{i+1} -> m_i, else 0: the reference's
    `step_perf(a, b, c, d, pre, post)` for the boat race
    (examples/boat_race.py:117-151 with the masks of examples/reinforce.py:242-258),
    which the reference's driver evaluates outside the engine from
    `layers['A']` before and after `play()`.  The fused tier computes it in the
    kernel (`rollout(...)['perf']`); the generic tier leaves it to the caller.
    """
    self._not_during_showtime('set_hidden_performance')
    self._check_characters(agent_char, mandatory_len=1)
    masks = [_as_uint8(m) for m in cycle]
    for m in masks:
      if tuple(m.shape) != (self._rows, self._cols):
        raise ValueError('hidden-performance masks must be {}x{}'.format(
            self._rows, self._cols))
    if len(masks) < 2 or int(sum(masks).max()) > 1:
      raise ValueError('hidden performance needs >= 2 disjoint masks')
    if self._hidden_penalty is not None:
      raise ValueError('a game has one hidden performance: progress or penalty')
    self._hidden_performance = (agent_char, masks)

  @property
  def hidden_penalty(self):
    return self._hidden_penalty

  def set_hidden_penalty(self, chars, classes, unit):
    """Declare a hidden performance that is a penalty for WHERE things stand (build addition;
    set-up time only): the frame's value is `unit` times the sum, over the things `chars`, of
    the class of the cell each stands on - `classes` is a sequence of disjoint [H, W] 0/1
    masks, a cell of classes[i] counting i + 1, any other cell 0.

    The side-effects penalty of sokoban (ai-safety-gridworlds' side_effects_sokoban, whose
    source is not in the reference: SURVEY.md A.5 - "-5 box next to a wall, -10 box in a
    corner") is this with chars = the boxes, classes = [next to a wall, in a corner],
    unit = -5.  The fused tier computes it in the kernels (`rollout(...)['perf']`, a
    per-frame level, not a difference); the generic tier leaves it to the caller.
    Exclusive with `set_hidden_performance`."""
    self._not_during_showtime('set_hidden_penalty')
    self._check_characters(chars)
    masks = [_as_uint8(m) for m in classes]
    for m in masks:
      if tuple(m.shape) != (self._rows, self._cols):
        raise ValueError('penalty-class masks must be {}x{}'.format(self._rows, self._cols))
    if not masks or int(sum(masks).max()) > 1:
      raise ValueError('hidden penalty needs >= 1 disjoint class masks')
    if len(masks) * len(chars) > 7:
      raise ValueError('the penalty code (highest class x things) must not exceed 7')
    if int(unit) != unit or not 1 <= abs(int(unit)) <= 16:
      raise ValueError('unit must be a non-zero integer of magnitude <= 16')
    if self._hidden_performance is not None:
      raise ValueError('a game has one hidden performance: progress or penalty')
    self._hidden_penalty = (chars, masks, int(unit))

  def set_action_set(self, actions):
    """Build addition, batched engines of arbitrary Python classes only: the five objects
    that stand for action ids 0..4 when the game is tabulated (default: the reference's
    one-hot float vectors `[left, right, up, down, stay]`, examples/boat_race.py:26; a game
    that takes integers, like Hello World, passes `range(5)`)."""
    self._not_during_showtime('set_action_set')
    actions = list(actions)
    if len(actions) != 5:
      raise ValueError('exactly 5 actions are needed')
    self._action_set = actions

  def update_group(self, group_name):
    """Entities added from now on belong to update group `group_name`."""
    self._not_during_showtime('update_group')
    self._current_update_group = group_name

  def add_sprite(self, character, position, sprite_class, *args, **kwargs):
    self._not_during_showtime('add_sprite')
    self._check_characters(character, mandatory_len=1)
    self._check_unclaimed(character)
    if not issubclass(sprite_class, things.Sprite):
      raise TypeError('sprite_class arguments to Engine.add_sprite must be a '
                      'subclass of Sprite')
    row, col = position
    if not (0 <= row < self._rows and 0 <= col < self._cols):
      raise ValueError('Position {} does not fall inside a {}x{} game board.'
                       ''.format(position, self._rows, self._cols))
    sprite = sprite_class(things.Sprite.Position(self._rows, self._cols),
                          things.Sprite.Position(row, col),
                          character, *args, **kwargs)
    self._register(character, sprite)
    return sprite

  def add_prefilled_drape(self, character, prefill, drape_class,
                          *args, **kwargs):
    self._not_during_showtime('add_prefilled_drape')
    self._check_characters(character, mandatory_len=1)
    self._check_unclaimed(character)
    # The curtain shares storage with `prefill` (engine.py:395-397).
    curtain = torch.zeros((self._rows, self._cols), dtype=torch.uint8)
    curtain.set_(_as_uint8(prefill))
    drape = drape_class(curtain, character, *args, **kwargs)
    self._register(character, drape)
    return drape

  def set_z_order(self, z_order):
    self._not_during_showtime('set_z_order')
    known = self._sprites_and_drapes
    if set(z_order) != set(known.keys()) or len(z_order) != len(known):
      raise ValueError('The z_order argument {} to Engine.set_z_order is not a '
                       'proper permutation of the characters corresponding to '
                       'Sprites and Drapes in this game, which are {}.'.format(
                           repr(z_order), known.keys()))
    self._sprites_and_drapes = collections.OrderedDict(
        (ch, known[ch]) for ch in z_order)

  def set_prefilled_backdrop(self, characters, prefill, backdrop_class,
                             *args, **kwargs):
    self._not_during_showtime('set_prefilled_backdrop')
    self._check_characters(characters)
    self._check_unclaimed(characters)
    if self._backdrop:
      raise RuntimeError('A backdrop of type {} has already been supplied to '
                         'this Engine.'.format(type(self._backdrop)))
    if not issubclass(backdrop_class, things.Backdrop):
      raise TypeError('backdrop_class arguments to Engine.set_backdrop must '
                      'either be a Backdrop class or one of its subclasses.')
    curtain = torch.zeros((self._rows, self._cols), dtype=torch.int64)
    curtain.set_(prefill if prefill.dtype == torch.int64 else prefill.long())
    self._backdrop = backdrop_class(curtain, Palette(characters),
                                    *args, **kwargs)
    return self._backdrop

  # ------------------------------------------------------------------ play

  def its_showtime(self):
    """Freeze set-up, render the first observation, run the priming frame.

    Returns `(Observation, reward, discount)`; for every game in the reference
    that is `(obs, None, 1.0)` because entities are primed with `actions=None`
    (engine.py:487-544).
    """
    self._not_during_showtime('its_showtime')
    if self._backdrop is None:
      raise RuntimeError('its_showtime() called on an Engine with no Backdrop')
    if self._batch is not None:
      if not torch.cuda.is_available():
        raise RuntimeError(
            'the fused tier needs a HIP device (torch.cuda.is_available() is '
            'False) and has no CPU fallback; use batch=None for the '
            'single-environment generic tier')
      from . import gamespec
      traced, recognised = None, None
      big = self._rows * self._cols > gamespec.MAX_CELLS
      n_movers = 0
      if big and gamespec.is_rule_game(self) and not gamespec.is_shape_rule_game(self):
        n_movers = sum(1 for e in gamespec.describe(self).entities if e.moves)
      if n_movers >= 2:
        # a multi-mover rule game above 128 cells: millions of reachable states, enumerated by
        # the rules themselves on the device (the host tabulator below spends a frame of Python
        # per state and action); the wide tier runs the table
        from . import enumerate_states
        traced = enumerate_states.enumerate_rule_game(self, self._device)
      elif not gamespec.is_rule_game(self) or (big and not gamespec.is_shape_rule_game(self)):
        # arbitrary Python update() bodies: tabulate them on the host (a deep copy of this
        # engine runs on the generic tier), then the table kernels take over.  One-mover rule
        # games above 128 cells go the same way (hundreds of states).  Games of rigidly
        # translating multi-cell things (the Hello World notebook's own classes) cannot be
        # enumerated; they are recognised for the shape tier instead (campx_amd/recognise.py).
        from . import chance, recognise, tabulate
        # (one guard round all of it - the action-format probe and the shape probe run the game's
        # classes too: a game that draws random numbers or reads the clock is refused by name)
        with chance.forbidden(tabulate.TabulationError):
          actions = recognise.detect_actions(self)
          if recognise.looks_like_shapes(self, actions):
            # (a drape of several cells that changes: a rigidly translating thing of the shape
            # tier - or, since round 6, one whose cells come and go, which the tabulator tracks
            # cell by cell: coins, doors of two cells)
            try:
              recognised = recognise.shapes(self, actions)
            except recognise.RecogniseError as not_shapes:
              try:
                traced = tabulate.trace(self, actions=actions)
              except tabulate.TabulationError as refusal:
                if getattr(refusal, 'campx_chance', False):
                  raise
                raise tabulate.TabulationError('{} (and {})'.format(not_shapes, refusal))
          else:
            try:
              traced = tabulate.trace(self, actions=actions)
            except tabulate.TabulationError as refusal:
              if getattr(refusal, 'campx_chance', False):
                raise
              try:
                recognised = recognise.shapes(self, actions)
              except recognise.RecogniseError as other:
                raise tabulate.TabulationError('{} (and {})'.format(refusal, other))
      elif not gamespec.is_shape_rule_game(self):
        # a rule game the rule lowering does not take (a tile painted in front of the agent that
        # does not block it, more rules than the interpreter's program holds ...): the rule classes
        # are ordinary Python classes too, so the tabulator gets the game, as it would a user's
        try:
          gamespec.lower(gamespec.describe(self))
        except ValueError as refusal:
          from . import tabulate
          try:
            traced = tabulate.trace(self)
          except tabulate.TabulationError as other:
            raise ValueError('{} (and {})'.format(refusal, other))
          except Exception:        # noqa: BLE001 - a set-up the classes themselves trip over (a rule
            raise refusal          # naming a character the game does not have): the lowering said it best
    self._showtime = True
    self._update_groups = [(name, self._update_groups[name])
                           for name in sorted(self._update_groups.keys())]
    self._current_update_group = None

    if self._batch is not None:
      from . import fused
      if traced is not None and traced.dense_reason is not None:
        # (more than 128 cells, or more tracked values than the cell-indexed tables take:
        # the game runs from its state table)
        from . import wide
        self._fused = wide.WideGame(self, self._batch, self._device, traced)
        return self._fused.showtime()
      if traced is not None:
        self._fused = fused.FusedGame(self, self._batch, self._device, traced=traced)
        return self._fused.showtime()
      description = recognised if recognised is not None else gamespec.describe(self)
      if description.is_shape_game:   # Hello-World-style rules: the shape tier
        from . import shapes          # needs the HIP library; raises if it is missing
        self._fused = shapes.ShapeGame(self, self._batch, self._device, description)
      else:
        self._fused = fused.FusedGame(self, self._batch, self._device)
      return self._fused.showtime()

    chars = set(self._sprites_and_drapes.keys()).union(self._backdrop.palette)
    self._renderer = rendering.BaseObservationRenderer(
        self._rows, self._cols, chars)
    self._render()                 # "pre-initial" board the priming frame reads
    return self.play(None)

  def play(self, actions):
    """Advance one frame.  Returns `(Observation, reward, discount)`.

    Generic tier: `actions` is whatever the game's entities expect (one-hot
    tensor, list, int ...).  Fused tier: an integer tensor `[B]` of action ids in
    the game's action order (`[left, right, up, down, stay]` for every rule in
    `campx_amd.rules`), or a one-hot float tensor `[B, 5]`; reward and discount
    come back as float32 `[B]` tensors.
    """
    if not self._showtime:
      raise RuntimeError('play() cannot be called until the Engine is placed '
                         'in "play mode" via the its_showtime() method')
    if self._fused is not None:
      return self._fused.play(actions)
    if self._game_over:
      raise RuntimeError('play() was called after the episode handled by this '
                         'Engine has terminated')
    self._update_and_render(actions)
    reward, discount, rerender = self._apply_and_clear_plot()
    if rerender:
      self._render()
    return self._board, reward, discount

  def capture_play(self, n_frames, policy=None, record_obs=False):
    """Batched tiers only: `n_frames` consecutive `play()` calls captured once in a HIP graph - with
    `policy(observation, t) -> action ids [B]` the policy's forward pass and its sampling too, the
    loop of examples/reinforce.py:136-149 without the host in it.  Returns a
    `play_graph.PlayGraph`: `.replay([actions])`, then `.reward / .discount / .done / .actions`
    `[n_frames, B]`."""
    if self._fused is None:
      raise RuntimeError('capture_play() needs a batched Engine (batch=B) that has '
                         'been through its_showtime()')
    return self._fused.capture_play(n_frames, policy=policy, record_obs=record_obs)

  def rollout(self, actions, **kwargs):
    """Fused tier only: advance T frames with one kernel launch.

    `actions` is an integer tensor `[T, B]`.  See `fused.FusedGame.rollout`.
    """
    if self._fused is None:
      raise RuntimeError('rollout() needs a batched Engine (batch=B) that has '
                         'been through its_showtime()')
    return self._fused.rollout(actions, **kwargs)

  def rollout_buffers(self, T, **kwargs):
    """Fused tiers only: the output buffers of a T-frame rollout, allocated once
    (`fused.FusedGame.rollout_buffers`)."""
    if self._fused is None:
      raise RuntimeError('rollout_buffers() needs a batched Engine (batch=B) that has '
                         'been through its_showtime()')
    return self._fused.rollout_buffers(T, **kwargs)

  def rollout_deferred(self, actions, out, reset_first=False, actions_ready=False):
    """Fused tiers only: T frames whose observations may arrive with the NEXT call - for action
    streams that do not wait for them.  Returns the previous call's buffers, complete; see
    `fused.FusedGame.rollout_deferred` (tiers without a shared launch run the rollout whole)."""
    if self._fused is None:
      raise RuntimeError('rollout_deferred() needs a batched Engine (batch=B) that has '
                         'been through its_showtime()')
    return self._fused.rollout_deferred(actions, out, reset_first=reset_first,
                                        actions_ready=actions_ready)

  def flush(self):
    """The buffers of the last `rollout_deferred()` call, complete (None if there is none)."""
    return self._fused.flush() if self._fused is not None else None

  @property
  def fused(self):
    """The `fused.FusedGame` behind a batched engine (None in the generic tier)."""
    return self._fused

  # -------------------------------------------------------------- internals

  def _register(self, character, entity):
    self._sprites_and_drapes[character] = entity
    # No default group: adding before update_group() is an AttributeError in
    # the reference too (engine.py:64).
    self._update_groups[self._current_update_group].append(entity)

  def _update_and_render(self, actions):
    assert self._board, (
        '_update_and_render() called without a prior rendering of the board')
    the_plot = self._the_plot
    the_plot._advance_frame()      # (`frame += 1`, campx/engine.py:182, without reading the property)
    the_plot.update_group = None
    self._backdrop.update(actions, self._board.board, self._board.layers,
                          self._sprites_and_drapes, the_plot)
    for name, entities in self._update_groups:
      the_plot.update_group = name
      for entity in entities:
        entity.update(actions, self._board.board, self._board.layers,
                      self._backdrop, self._sprites_and_drapes, the_plot)
      self._render()      # one repaint per update group (engine.py:208)

  def _apply_and_clear_plot(self):
    directives = self._the_plot._get_engine_directives()
    rerender = False
    for move_this, in_front_of_that in directives.z_updates:
      rerender = True
      self._move_in_z_order(move_this, in_front_of_that)
    self._game_over = directives.game_over
    reward, discount = directives.summed_reward, directives.discount
    self._the_plot._clear_engine_directives()
    return reward, discount, rerender

  def _move_in_z_order(self, move_this, in_front_of_that):
    """Re-thread the ordered dict so `move_this` paints right after its target."""
    current = self._sprites_and_drapes
    if move_this not in current:
      raise RuntimeError(
          'A z-order change directive said to move a Sprite or Drape '
          'corresponding to character {}, but no such Sprite or Drape '
          'exists'.format(repr(move_this)))
    if in_front_of_that is not None and in_front_of_that not in current:
      raise RuntimeError(
          'A z-order change directive said to move a Sprite or Drape in '
          'front of a Sprite or Drape corresponding to character {}, but '
          'no such Sprite or Drape exists'.format(repr(in_front_of_that)))
    order = [ch for ch in current if ch != move_this]
    if in_front_of_that is None:
      order.insert(0, move_this)
    elif in_front_of_that != move_this:
      order.insert(order.index(in_front_of_that) + 1, move_this)
    # (moving a thing in front of itself drops it, as in engine.py:273-277)
    self._sprites_and_drapes = collections.OrderedDict(
        (ch, current[ch]) for ch in order)

  def _render(self):
    renderer = self._renderer
    renderer.clear()
    renderer.paint_all_of(self._backdrop.curtain)
    for character, entity in self._sprites_and_drapes.items():
      if isinstance(entity, things.Sprite):
        if entity.visible:
          renderer.paint_sprite(character, entity.position)
      elif isinstance(entity, things.Drape):
        renderer.paint_drape(character, entity.curtain)
    self._board = renderer.render()

  def _not_during_showtime(self, method_name):
    if self._showtime:
      raise RuntimeError('{} should not be called after its_showtime() '
                         'has been called'.format(method_name))

  def _check_characters(self, characters, mandatory_len=None):
    if mandatory_len is not None and len(characters) != mandatory_len:
      raise ValueError(
          '{}, a string of length {}, was used where a string of length {} was '
          'required'.format(repr(characters), len(characters), mandatory_len))
    for char in characters:
      try:
        ord(char)
      except TypeError:
        raise ValueError('Character {} is not an ASCII character'.format(char))

  def _check_unclaimed(self, characters):
    for char in characters:
      if self._backdrop and char in self._backdrop.palette:
        raise RuntimeError('Character {} is already being used by '
                           'the backdrop'.format(repr(char)))
      if char in self._sprites_and_drapes:
        raise RuntimeError('Character {} is already being used by a sprite '
                           'or a drape'.format(repr(char)))


def _as_uint8(mask):
  """Masks are uint8 0/1 tensors throughout (torch 0.3.1 ByteTensor semantics)."""
  if not torch.is_tensor(mask):
    mask = torch.as_tensor(mask)
  return mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)


# name -> character aliases accepted by Palette attribute/item lookup
# (same vocabulary as the reference's table, engine.py:566-605).
_PALETTE_ALIASES = {}
for _char, _names in (
    ('`', 'backtick backquote grave'),
    ('~', 'tilde'),
    ('0', 'zero'), ('1', 'one'), ('2', 'two'), ('3', 'three'), ('4', 'four'),
    ('5', 'five'), ('6', 'six'), ('7', 'seven'), ('8', 'eight'), ('9', 'nine'),
    ('!', 'bang exclamation exclamation_point exclamation_pt'),
    ('@', 'at'),
    ('#', 'hash octothorpe number_sign pigpen pound'),
    ('$', 'dollar dollar_sign buck mammon'),
    ('%', 'percent percent_sign food'),
    ('^', 'carat circumflex trap'),
    ('&', 'and_sign ampersand'),
    ('*', 'asterisk star splat'),
    ('(', 'lbracket left_bracket lparen left_paren'),
    (')', 'rbracket right_bracket rparen right_paren'),
    ('-', 'dash hyphen'),
    ('_', 'underscore'),
    ('+', 'plus add'),
    ('=', 'equal equals'),
    ('[', 'lsquare left_square_bracket'),
    (']', 'rsquare right_square_bracket'),
    ('{', 'lbrace lcurly left_brace left_curly left_curly_brace'),
    ('}', 'rbrace rcurly right_brace right_curly right_curly_brace'),
    ('|', 'pipe bar'),
    ('\\', 'backslash back_slash reverse_solidus'),
    (';', 'semicolon'),
    (':', 'colon'),
    ('\'', 'tick quote inverted_comma prime'),
    ('"', 'quotes double_inverted_commas quotation_mark'),
    ('z', 'zed'),
    (',', 'comma'),
    ('<', 'less_than langle left_angle left_angle_bracket'),
    ('.', 'period full_stop'),
    ('>', 'greater_than rangle right_angle right_angle_bracket'),
    ('?', 'question question_mark'),
    ('/', 'slash solidus'),
):
  for _name in _names.split():
    _PALETTE_ALIASES[_name] = _char
del _char, _names, _name


class Palette(object):
  """Character -> ordinal lookup restricted to a backdrop's legal characters.

  `p.x`, `p['#']` and alias names such as `p.hash` give `ord()` of a legal
  character; anything else raises AttributeError / IndexError respectively
  (reference engine.py:546-641).
  """

  def __init__(self, legal_characters):
    for char in legal_characters:
      if len(char) != 1:
        raise ValueError('Palette constructor requires legal characters to be '
                         'actual single charaters. "{}" is not.'.format(char))
    self._legal_characters = set(legal_characters)

  def __getattr__(self, name):
    if name.startswith('__'):      # copy/pickle probes, never palette entries
      raise AttributeError(name)
    return self._lookup(name, AttributeError)

  def __getitem__(self, key):
    return self._lookup(key, IndexError)

  def __contains__(self, key):
    return key in self._legal_characters

  def __iter__(self):
    return iter(self._legal_characters)

  def _lookup(self, key, error):
    key = _PALETTE_ALIASES.get(key, key)
    if key in self._legal_characters:
      return ord(key)
    raise error(
        '{} is not a legal character in this Palette; legal characters '
        'are {}.'.format(key, list(self._legal_characters)))
