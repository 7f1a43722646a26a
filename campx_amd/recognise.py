"""Recognise a game of arbitrary Python classes as a SHAPE game (the Hello World kind).

The shape tier (`shapes.ShapeGame`, csrc/k_shape.hip) runs games whose things are rigid
shapes that translate cyclically by a per-action offset and interact with nothing: the
reference's examples/Hello World Example.ipynb (cell 3 `RollingDrape`, `SlidingSprite`; cell 4
`make_game()`, z-order '12@34').  Such a game has rows*cols positions PER THING, so its joint
states cannot be enumerated the way `tabulate.trace()` does; `campx_amd.rules` holds re-typed
classes that declare their offsets (`fused_rule()`), and until round 4 only those reached the
device.

This module gives the notebook's OWN classes - any user classes of that kind - the same
`gamespec.GameDescription` by watching them on the generic tier (`engine.Engine` with
`batch=None`, the reference's execution model, campx/engine.py:114-324).  Since round 5 the
check is a PROOF over the game's own code, not a sample of walks:

1. THE MODEL.  From the state `its_showtime()` leaves, each of the five actions is played once
   on a deep copy; for every thing the curtain (a drape) or position (a sprite,
   campx/things.py:294) after the frame must be the one before it rolled by some (rows, cols)
   offset, and the reward it added (campx/plot.py:186-211; at most one `add_reward` per thing
   and frame) and whether it ended the episode (discount 0.0 only) are noted per (thing,
   action) - the probe's Plot records WHO called.  The frame's total reward (`r + total` in
   update order) and discount depend on the action alone, so these five frames pin them for
   good - given step 2.

2. EVERY THING IS INDEPENDENT, EVERYWHERE IT CAN GET.  Each thing (and the Backdrop) is then
   taken alone: its `update()` is called directly, for every position the thing can reach
   under its own offsets (rows * cols of them for Hello World's things) times the five
   actions - 468 x 5 x 5 calls for Hello World - with RECORDING stand-ins for everything
   `update()` is handed besides the action (campx/engine.py:200-204): `board`, every
   `layers[ch]`, `backdrop.curtain` and the curtains of the other `things` are tensors that
   note any operation on them, other entries of `things` and other attributes of the backdrop
   note being fetched, the Plot notes reads and writes of its entries, `frame` reads,
   z-order and default-discount requests.  Any such access is a refusal: the thing's behaviour
   could depend on (or change) something besides itself.  What is left is a function of (the
   thing's own state, the action), and it is checked on every point of its domain: the new
   curtain / position is the old one moved by the action's offset, the reward and termination
   are the model's, visibility and every other attribute of the entity stay as they were.  By
   induction over frames the model then holds for every action sequence, whatever the other
   things do.  (One input is beyond this: module globals, closures, clocks and random number
   generators that `update()` consults behind the engine's back.  As in `tabulate.py` those
   are only spot-checked - by the walks of step 3.)

3. CROSS-CHECK of this module's own model against the engine (rendering with the trails that
   sprites painted before the first drape leave in the backdrop, campx/rendering.py:128,150;
   reward summation; game over): all 25 two-action openings, every action repeated once round
   the board, and `WALKS` random walks of `WALK_FRAMES` frames, every thing's curtain /
   position, reward bit for bit, discount, game-over and rendered board compared frame by
   frame.  The same recording stand-ins are in force during these frames.

`RecogniseError` (a ValueError) names the first thing, position and action that contradicts
the model.  Host logic only; no GPU.
"""

import copy

import numpy as np
import torch

from . import gamespec
from . import tabulate
from . import things as _things

N_ACTIONS = gamespec.N_ACTIONS
WALKS = 8
WALK_FRAMES = 48


class RecogniseError(ValueError):
  pass


def _fail(msg):
  raise RecogniseError('not a game of rigidly translating things (shape tier): ' + msg)


# ---------------------------------------------------------------- who called the Plot

_CALLS = []          # (character of the thing being updated, 'reward' | 'end', value)
_NOW = [None]
_RECORDING = [False]  # the engine's own update() calls get recording stand-ins too
BACKDROP = 'the Backdrop'
_WATCHED = {}
_WATCHED_PLOTS = {}


def _watched_class(cls):
  """`cls` with an `update()` that notes whose turn it is."""
  if cls in _WATCHED:
    return _WATCHED[cls]
  if getattr(cls, '_campx_watched', False):
    return cls

  class Watched(cls):
    _campx_watched = True

    def update(self, *args, **kwargs):
      _NOW[0] = getattr(self, 'character', BACKDROP)
      try:
        if _RECORDING[0] and len(args) == 5 and not kwargs and args[0] is not None:
          # (the Backdrop's call, campx/engine.py:190-192)
          actions, board, layers, all_things, the_plot = args
          args = (actions, _watched_tensor(board, 'board'),
                  {ch: _watched_tensor(t, 'layers[{!r}]'.format(ch)) for ch, t in layers.items()},
                  _WatchedThings(all_things, None), the_plot)
        elif _RECORDING[0] and len(args) == 6 and not kwargs and args[0] is not None:
          # (the engine's call, campx/engine.py:202-204: board, layers, backdrop, things
          # replaced by their recording twins)
          actions, board, layers, backdrop, all_things, the_plot = args
          args = (actions, _watched_tensor(board, 'board'),
                  {ch: _watched_tensor(t, 'layers[{!r}]'.format(ch)) for ch, t in layers.items()},
                  _WatchedBackdrop(backdrop), _WatchedThings(all_things, self.character),
                  the_plot)
        return cls.update(self, *args, **kwargs)
      finally:
        _NOW[0] = None

  Watched.__name__, Watched.__qualname__ = cls.__name__, cls.__qualname__
  _WATCHED[cls] = Watched
  return Watched


def _watched_plot(base):
  if base in _WATCHED_PLOTS:
    return _WATCHED_PLOTS[base]

  class WatchedPlot(base):

    def add_reward(self, reward):
      _CALLS.append((_NOW[0], 'reward', reward))
      return base.add_reward(self, reward)

    def terminate_episode(self, *args, **kwargs):
      out = base.terminate_episode(self, *args, **kwargs)
      _CALLS.append((_NOW[0], 'end', self._engine_directives.discount))
      return out

    def change_default_discount(self, *args, **kwargs):
      _CALLS.append((_NOW[0], 'discount', None))
      return base.change_default_discount(self, *args, **kwargs)

    def change_z_order(self, *args, **kwargs):
      _CALLS.append((_NOW[0], 'z-order', None))
      return base.change_z_order(self, *args, **kwargs)

    # The Plot is a dict for the game's own entries (campx/plot.py:29): inside an update() any
    # access to one is noted - an entry may alias the renderer's live layers
    # (`the_plot['prev_pos_A'] = layers['A']`, examples/boat_race.py:59), and a written one is
    # state outside the thing.  (`log()` appends to its own key: write-only, left alone.)
    def _entry(self, key, how):
      if _NOW[0] is not None and key != self.LOG_KEY:
        _READS.append((_NOW[0], 'the_plot[{!r}] ({})'.format(key, how)))

    def __getitem__(self, key):
      self._entry(key, 'read')
      return base.__getitem__(self, key)

    def get(self, key, default=None):
      self._entry(key, 'read')
      return base.get(self, key, default)

    def __contains__(self, key):
      self._entry(key, 'read')
      return base.__contains__(self, key)

    def __setitem__(self, key, value):
      self._entry(key, 'written')
      return base.__setitem__(self, key, value)

    def __delitem__(self, key):
      self._entry(key, 'written')
      return base.__delitem__(self, key)

    def setdefault(self, key, default=None):
      self._entry(key, 'written')
      return base.setdefault(self, key, default)

    def pop(self, key, *default):
      self._entry(key, 'written')
      return base.pop(self, key, *default)

    def update(self, *args, **kwargs):
      self._entry('...', 'written')
      return base.update(self, *args, **kwargs)

    def clear(self):
      self._entry('...', 'written')
      return base.clear(self)

    def keys(self):
      self._entry('...', 'read')
      return base.keys(self)

    def values(self):
      self._entry('...', 'read')
      return base.values(self)

    def items(self):
      self._entry('...', 'read')
      return base.items(self)

    def __iter__(self):
      self._entry('...', 'read')
      return base.__iter__(self)

  _WATCHED_PLOTS[base] = WatchedPlot
  return WatchedPlot


# ---------------------------------------------------------------- what update() is handed

_READS = []          # (character of the thing being updated, what it touched)
_META = frozenset(('shape', 'size', 'dim', 'ndim', 'ndimension', 'dtype', 'device', 'numel',
                   'nelement', 'stride', 'is_contiguous', 'is_floating_point', 'layout',
                   'requires_grad', 'is_cuda', 'is_sparse', 'element_size', 'type', 'names'))


def _func_name(func):
  owner = getattr(func, '__self__', None)          # a property getter: Tensor.shape.__get__
  if getattr(func, '__name__', '') in ('__get__', '__set__') and hasattr(owner, '__name__'):
    return owner.__name__
  return getattr(func, '__name__', repr(func))


class _WatchedTensor(torch.Tensor):
  """A tensor that notes every operation it takes part in (reading its metadata aside); the
  operation itself runs on plain tensors and returns plain tensors."""

  @classmethod
  def __torch_function__(cls, func, types, args=(), kwargs=None):
    kwargs = kwargs or {}
    name = _func_name(func)
    if name not in _META:
      seen = []

      def look(x):
        if isinstance(x, _WatchedTensor):
          seen.append(x)
        elif isinstance(x, (list, tuple)):
          for y in x:
            look(y)
      look(args)
      look(list(kwargs.values()))
      for x in seen:
        _READS.append((_NOW[0], '{} ({})'.format(getattr(x, '_campx_label', 'a tensor'), name)))
    with torch._C.DisableTorchFunctionSubclass():
      return func(*args, **kwargs)


def _watched_tensor(t, label):
  w = torch.Tensor._make_subclass(_WatchedTensor, t.detach())
  w._campx_label = label
  return w


class _WatchedThings(dict):
  """`things` as update() gets it (campx/engine.py:203): the thing's own entry is itself;
  fetching another one is noted."""

  def __init__(self, real, own):
    dict.__init__(self, real)
    self._own = own

  def _note(self, key):
    if key != self._own:
      _READS.append((_NOW[0], 'things[{!r}]'.format(key)))

  def __getitem__(self, key):
    self._note(key)
    return dict.__getitem__(self, key)

  def get(self, key, default=None):
    self._note(key)
    return dict.get(self, key, default)

  def values(self):
    _READS.append((_NOW[0], 'things.values()'))
    return dict.values(self)

  def items(self):
    _READS.append((_NOW[0], 'things.items()'))
    return dict.items(self)


class _WatchedBackdrop(object):
  """The Backdrop as a thing's update() gets it: the curtain is a watched tensor, the palette a
  constant, any other attribute noted."""

  def __init__(self, real):
    object.__setattr__(self, '_real', real)
    object.__setattr__(self, '_curtain_w', _watched_tensor(real.curtain, 'backdrop.curtain'))

  @property
  def curtain(self):
    return self._curtain_w

  @property
  def palette(self):
    return self._real.palette

  def __getattr__(self, name):
    _READS.append((_NOW[0], 'backdrop.' + name))
    return getattr(self._real, name)

  def __setattr__(self, name, value):
    _READS.append((_NOW[0], 'backdrop.{} (written)'.format(name)))
    setattr(self._real, name, value)


def _stand_ins(eng, own):
  """(board, layers, backdrop, things) for one `update()` call of the thing `own` (None: the
  Backdrop) on the engine `eng`, all recording."""
  obs = eng._board
  board = _watched_tensor(obs.board, 'board')
  layers = {ch: _watched_tensor(t, 'layers[{!r}]'.format(ch)) for ch, t in obs.layers.items()}
  return board, layers, _WatchedBackdrop(eng.backdrop), _WatchedThings(eng.things, own)


# ---------------------------------------------------------------- actions

def detect_actions(engine):
  """The five objects `play()` takes for action ids 0..4: `engine._action_set` when the game
  set one, else the first of the reference's conventions the game's own classes accept for a
  whole frame of every action - one-hot float vectors (examples/boat_race.py:26), then plain
  integers (Hello World: `game.play(0)`, notebook cell 6)."""
  if engine._action_set is not None:
    return list(engine._action_set)
  errors = []
  for name, actions in (('one-hot float vectors', tabulate.default_actions()),
                        ('integers', list(range(N_ACTIONS)))):
    probe = tabulate.clone_engine(engine)
    probe._batch, probe._device, probe._fused = None, None, None
    try:
      probe.its_showtime()
      for a in range(N_ACTIONS):
        trial = tabulate.clone_engine(probe)
        trial.play(copy.deepcopy(actions[a]))
      return actions
    except Exception as e:      # noqa: BLE001 - whatever the user's classes raise
      errors.append('{}: {}: {}'.format(name, type(e).__name__, str(e)[:80]))
  raise RecogniseError('the game\'s classes accept neither action format (' + '; '.join(errors) +
                       '); use Engine.set_action_set()')


# ---------------------------------------------------------------- the model

class _Model(object):
  """Offsets per thing, the backdrop with its trails; `step(a)` / `board()` as the shape
  kernels will compute them."""

  def __init__(self, H, W, order, schedule, masks, is_sprite, visible, backdrop):
    self.H, self.W = H, W
    self.order, self.schedule = order, schedule          # z-order / update order (characters)
    self.masks, self.is_sprite, self.visible = masks, is_sprite, visible
    self.drow = {ch: [0] * N_ACTIONS for ch in order}
    self.dcol = {ch: [0] * N_ACTIONS for ch in order}
    self.reward = {ch: [None] * N_ACTIONS for ch in order}
    self.ends = {ch: [False] * N_ACTIONS for ch in order}
    drapes = [i for i, ch in enumerate(order) if not is_sprite[ch]]
    self.first_drape = drapes[0] if drapes else None
    self.backdrop0 = backdrop
    self.reset()

  def reset(self):
    self.off = {ch: (0, 0) for ch in self.order}
    self.backdrop = self.backdrop0.copy()
    self.over = False

  def mask(self, ch):
    dr, dc = self.off[ch]
    return np.roll(self.masks[ch], (dr, dc), (0, 1))

  def board(self):
    """campx/engine.py:306-324 on campx/rendering.py:104-178: sprites in front of nothing but
    the backdrop are painted INTO it."""
    for ch in self.order[:self.first_drape]:
      if self.visible[ch]:
        self.backdrop[self.mask(ch) != 0] = ord(ch)
    board = self.backdrop.copy()
    for ch in self.order[self.first_drape:]:
      if self.is_sprite[ch] and not self.visible[ch]:
        continue
      board[self.mask(ch) != 0] = ord(ch)
    return board

  def step(self, a):
    total = None
    for ch in self.schedule:
      dr, dc = self.off[ch]
      self.off[ch] = ((dr + self.drow[ch][a]) % self.H, (dc + self.dcol[ch][a]) % self.W)
      r = self.reward[ch][a]
      if r is not None:
        total = r if total is None else np.float32(r + total)      # plot.py:211, in float32
      if self.ends[ch][a]:
        self.over = True
    return (np.float32(np.nan) if total is None else np.float32(total),
            0.0 if self.over else 1.0)


def _offsets_between(before, after, H, W):
  """Every (dr, dc) with roll(before) == after, smallest move first."""
  found = []
  cells = np.argwhere(before != 0)
  target = np.argwhere(after != 0)
  if len(cells) != len(target) or len(cells) == 0:
    return found
  r0, c0 = cells[0]
  for r1, c1 in target:                                  # the first cell lands on one of these
    dr, dc = int(r1 - r0) % H, int(c1 - c0) % W
    if np.array_equal(np.roll(before, (dr, dc), (0, 1)), after):
      found.append((dr, dc))
  def size(o):
    return min(o[0], H - o[0]) + min(o[1], W - o[1])
  return sorted(set(found), key=lambda o: (size(o), o))


def _thing_mask(ent, H, W):
  if isinstance(ent, _things.Sprite):
    mask = np.zeros((H, W), np.uint8)
    mask[ent.position.row % H, ent.position.col % W] = 1
    return mask
  return ent.curtain.detach().cpu().numpy().astype(np.uint8)


def _entity_extras(ent, kind):
  """Image of everything an entity holds besides its curtain / position / visibility."""
  out = []
  for name, value in sorted(vars(ent).items()):
    if name in tabulate._CORE_ATTRS[kind]:
      continue
    try:
      out.append((name, tabulate._plain(value, name, 0, frozenset())))
    except tabulate._Unimageable as e:
      _fail('attribute {} of {!r} is not plain data'.format(e, getattr(ent, 'character', 'the Backdrop')))
  return tuple(out)


def _prove_independent(probe, model, actions, reads0):
  """Step 2 of the module docstring.  `probe` is the watched engine right after
  `its_showtime()`; `model` holds the per-(thing, action) offsets, rewards and terminations
  inferred from one frame of every action.  Calls every thing's `update()` directly - on a
  copy of the engine - for every position its own offsets can take it to x every action,
  handing it recording stand-ins; raises `RecogniseError` on the first access to anything but
  itself, or the first (position, action) where it does not do what the model says."""
  H, W = model.H, model.W
  eng = tabulate.clone_engine(probe)
  plot = eng._the_plot

  def expected_calls(ch, a):
    want = []
    if model.reward[ch][a] is not None:
      want.append(('reward', np.array([model.reward[ch][a]], np.float32).view(np.uint32)[0]))
    if model.ends[ch][a]:
      want.append(('end', 0.0))
    return sorted(want)

  def one_call(ent, ch, a, where, backdrop_call=False):
    del _CALLS[:]
    del _READS[:]
    board, layers, backdrop, all_things = _stand_ins(eng, ch)
    action = copy.deepcopy(actions[a])
    _RECORDING[0] = False        # (the stand-ins are handed over here)
    if backdrop_call:
      ent.update(action, board, layers, all_things, plot)       # campx/engine.py:190-192
    else:
      ent.update(action, board, layers, backdrop, all_things, plot)
    plot._clear_engine_directives()
    if tabulate.FRAME_READS[0] != reads0:
      _fail(where + ' reads the_plot.frame')
    if _READS:
      _fail('{} looks at / touches {}: a thing of a shape game depends on nothing but itself '
            'and the action'.format(where, _READS[0][1]))
    got = []
    for who, what, value in _CALLS:
      if what == 'reward':
        got.append(('reward', np.array([tabulate.reward_f32(value)], np.float32).view(np.uint32)[0]))
      elif what == 'end':
        got.append(('end', float(value)))
      else:
        _fail('{} asks the Plot for a {} change'.format(where, what))
    return sorted(got)

  # the Backdrop: one state, five actions; it may do nothing at all
  bd = eng.backdrop
  if getattr(type(bd), '_campx_watched', False):
    art = bd.curtain.detach().clone()
    extras = _entity_extras(bd, 'backdrop')
    for a in range(N_ACTIONS):
      where = BACKDROP + ', action {}:'.format(a)
      if one_call(bd, None, a, where, backdrop_call=True):
        _fail(where + ' it talks to the Plot')
      if not torch.equal(bd.curtain, art) or _entity_extras(bd, 'backdrop') != extras:
        _fail(where + ' it changes')

  for ch in model.schedule:
    ent = eng.things[ch]
    sprite = model.is_sprite[ch]
    kind = 'sprite' if sprite else 'drape'
    start = model.masks[ch]
    extras = _entity_extras(ent, kind)
    offs = [(model.drow[ch][a], model.dcol[ch][a]) for a in range(N_ACTIONS)]
    r0 = c0 = 0
    if sprite:
      r0, c0 = int(ent.position.row), int(ent.position.col)
      if not (0 <= r0 < H and 0 <= c0 < W):
        _fail('sprite {!r} stands outside the board'.format(ch))
    seen = {(0, 0)}
    frontier = [(0, 0)]
    while frontier:
      pos = frontier.pop()
      for a in range(N_ACTIONS):
        if sprite:
          ent._position = ent.Position((r0 + pos[0]) % H, (c0 + pos[1]) % W)
        else:
          ent._curtain = torch.from_numpy(np.roll(start, pos, (0, 1)).copy())
        where = '{!r} moved by (rows {}, cols {}) from its start, action {}:'.format(
            ch, pos[0], pos[1], a)
        got = one_call(ent, ch, a, where)
        if got != expected_calls(ch, a):
          _fail(where + ' its reward / termination is not the one action {} has at its '
                'start'.format(a))
        nxt = ((pos[0] + offs[a][0]) % H, (pos[1] + offs[a][1]) % W)
        if sprite:
          at = ent.position
          ok = (int(at.row), int(at.col)) == ((r0 + nxt[0]) % H, (c0 + nxt[1]) % W) \
              and bool(ent.visible) == model.visible[ch]
        else:
          now = ent.curtain.detach().cpu().numpy()
          ok = now.shape == start.shape and np.array_equal(now, np.roll(start, nxt, (0, 1)))
        if not ok:
          _fail(where + ' it is not moved by the action\'s offset (rows {}, cols {}): what it '
                'does depends on where it is'.format(offs[a][0], offs[a][1]))
        if _entity_extras(ent, kind) != extras:
          _fail(where + ' state outside its curtain / position changed (an attribute of the '
                'entity)')
        if nxt not in seen:
          seen.add(nxt)
          frontier.append(nxt)


def looks_like_shapes(engine, actions):
  """One frame of every action from the start: True when the cell-indexed / state tables
  cannot be the right home for this game - a DRAPE that moves covers more than one cell, or a visible sprite is painted before the first drape (it writes into the backdrop,
  campx/rendering.py:128,150) - so that a batched Engine asks `shapes()` first instead of
  walking `tabulate.trace()` into its refusal."""
  H, W = engine.rows, engine.cols
  probe = tabulate.clone_engine(engine)
  probe._batch, probe._device, probe._fused = None, None, None
  try:
    probe.its_showtime()
    order = list(probe.things.values())
    drapes = [i for i, ent in enumerate(order) if not isinstance(ent, _things.Sprite)]
    if drapes and any(isinstance(ent, _things.Sprite) and ent.visible for ent in order[:drapes[0]]):
      return True
    before = {ent.character: _thing_mask(ent, H, W) for ent in order}
    for a in range(N_ACTIONS):
      eng = tabulate.clone_engine(probe)
      eng.play(copy.deepcopy(actions[a]))
      for ch, ent in eng.things.items():
        if isinstance(ent, _things.Sprite):
          continue
        after = _thing_mask(ent, H, W)
        # (a drape that empties - a coin collected - is the tabulator's "absent" thing)
        if not np.array_equal(after, before[ch]) and (before[ch].sum() > 1 or after.sum() > 1):
          return True
  except Exception:       # noqa: BLE001 - let the tabulator report what is wrong with the game
    return False
  return False


def shapes(engine, actions=None):
  """A set-up (not started) `Engine` of arbitrary classes -> the `gamespec.GameDescription`
  `gamespec.describe()` gives for the same game written with `rules.RollingDrape` /
  `rules.SlidingSprite` / `FixedDrape`, or `RecogniseError`.  `engine` is not touched.  A game
  that draws random numbers or reads the clock is refused (`tabulate.TabulationError`:
  campx_amd/chance.py - stand-ins for the sources of chance while the classes are run)."""
  from . import chance
  with chance.forbidden(tabulate.TabulationError):
    return _shapes(engine, actions)


def _shapes(engine, actions):
  if engine.backdrop is None:
    raise ValueError('the Engine has no Backdrop yet')
  H, W = engine.rows, engine.cols
  if engine.hidden_performance is not None or engine.hidden_penalty is not None:
    _fail('hidden performance is not offered on the shape tier')
  actions = detect_actions(engine) if actions is None else list(actions)
  if len(actions) != N_ACTIONS:
    raise ValueError('exactly {} actions are needed'.format(N_ACTIONS))
  # (a live Sprite / Drape / Engine / Plot reached through a module global, a closure or a default
  # argument: the recording stand-ins below never see it read - refused by its name, statically)
  behind = tabulate.reached_behind_the_engine(engine)
  if behind:
    _fail(behind)

  probe = tabulate.clone_engine(engine)
  probe._batch, probe._device, probe._fused = None, None, None
  probe._the_plot.__class__ = _watched_plot(tabulate.probe_plot_class(type(probe._the_plot)))
  for ent in list(probe.things.values()) + [probe.backdrop]:
    if ent is probe.backdrop and type(ent).update is _things.Backdrop.update:
      continue
    try:
      ent.__class__ = _watched_class(type(ent))
    except TypeError as e:
      _fail('{!r}: its class cannot be watched ({})'.format(getattr(ent, 'character', BACKDROP), e))
  start_masks = {ch: _thing_mask(ent, H, W) for ch, ent in probe.things.items()}
  backdrop_art = probe.backdrop.curtain.detach().cpu().numpy().astype(np.uint8).copy()
  obs, _, _ = probe.its_showtime()
  if probe.game_over:
    _fail('the episode is over after its_showtime()')
  reads0 = tabulate.FRAME_READS[0]
  order = list(probe.things.keys())                      # z-order, back to front
  schedule, group_of = [], {}
  for gi, (_, members) in enumerate(probe._update_groups):
    for ent in members:
      schedule.append(ent.character)
      group_of[ent.character] = gi
  is_sprite = {ch: isinstance(ent, _things.Sprite) for ch, ent in probe.things.items()}
  if all(is_sprite.values()):
    _fail('a game with no drape at all: the reference renderer zeroes its own backdrop on the '
          'second render (campx/rendering.py:111,128)')
  visible = {ch: (bool(ent.visible) if is_sprite[ch] else True)
             for ch, ent in probe.things.items()}
  masks = {ch: _thing_mask(ent, H, W) for ch, ent in probe.things.items()}
  for ch in order:
    if not np.array_equal(masks[ch], start_masks[ch]):
      _fail('{!r} moves during its_showtime()'.format(ch))
    if masks[ch].max() > 1:
      _fail('the curtain of {!r} holds values other than 0 and 1'.format(ch))
  hidden0 = tabulate.hidden_image(probe, False)
  model = _Model(H, W, order, schedule, masks, is_sprite, visible, backdrop_art)
  if not np.array_equal(model.board(), obs.board.detach().cpu().numpy().astype(np.uint8)):
    _fail('the first observation is not "backdrop, then every thing in z-order"')
  model.backdrop0 = model.backdrop.copy()                # (the first frame's trail cells)

  def refuse_reads(where):
    if tabulate.FRAME_READS[0] != reads0:
      _fail('the game reads the_plot.frame')
    if _READS:
      who, what = _READS[0]
      _fail('{}{!r} looks at / touches {}: a thing of a shape game depends on nothing but '
            'itself and the action'.format(where, who, what))

  def play(eng, a, where=''):
    del _CALLS[:]
    del _READS[:]
    _RECORDING[0] = True
    try:
      obs, reward, discount = eng.play(copy.deepcopy(actions[a]))
    finally:
      _RECORDING[0] = False
    refuse_reads(where)
    return obs, tabulate.reward_f32(reward), float(np.float32(discount)), list(_CALLS)

  # ---- 1. one frame of every action from the start: offsets, rewards, who ends the episode
  for a in range(N_ACTIONS):
    eng = tabulate.clone_engine(probe)
    play(eng, a)
    calls = list(_CALLS)
    for ch, ent in eng.things.items():
      after = _thing_mask(ent, H, W)
      found = _offsets_between(masks[ch], after, H, W)
      if not found:
        _fail('action {} does not translate {!r} rigidly (its cells before and after the '
              'frame are not a cyclic shift of each other)'.format(a, ch))
      model.drow[ch][a], model.dcol[ch][a] = found[0]
    for who, what, value in calls:
      if who is None or who not in model.reward:
        _fail('the Backdrop (or code outside any thing\'s update) talks to the Plot')
      if what == 'discount':
        _fail('{!r} changes the default discount'.format(who))
      if what == 'z-order':
        _fail('{!r} asks for a z-order change'.format(who))
      if what == 'end':
        if value != 0.0:
          _fail('{!r} ends the episode with discount {}'.format(who, value))
        model.ends[who][a] = True
      else:
        if model.reward[who][a] is not None:
          _fail('{!r} adds more than one reward in a frame'.format(who))
        # (as float32, summed in float32: that this equals the generic tier's own sum for
        # every action is part of what the walks below check)
        model.reward[who][a] = tabulate.reward_f32(value)

  # ---- 2. every thing alone, at every position it can reach, under every action
  _prove_independent(probe, model, actions, reads0)

  # ---- 3. the model against the engine, frame by frame (a cross-check of THIS module)
  def follow(eng, seq, what):
    model.reset()
    for t, a in enumerate(seq):
      where = '{} frame {} (action {})'.format(what, t, a)
      obs, reward, discount, _ = play(eng, a, where + ': ')
      want_reward, want_discount = model.step(a)
      if list(eng.things.keys()) != order:
        _fail(where + ': the z-order changed')
      for ch, ent in eng.things.items():
        if is_sprite[ch] and bool(ent.visible) != visible[ch]:
          _fail(where + ': sprite {!r} changed its visibility'.format(ch))
        if not np.array_equal(_thing_mask(ent, H, W), model.mask(ch)):
          _fail(where + ': {!r} is not where its per-action offsets put it (it depends on '
                'something besides the action)'.format(ch))
      if (np.array([reward]).view(np.uint32)[0] != np.array([want_reward]).view(np.uint32)[0]
          or discount != want_discount or bool(eng.game_over) != model.over):
        _fail(where + ': reward / discount / game-over {} {} {} where the per-action constants '
              'give {} {} {}'.format(reward, discount, eng.game_over, want_reward, want_discount,
                                     model.over))
      if not np.array_equal(obs.board.detach().cpu().numpy().astype(np.uint8), model.board()):
        _fail(where + ': the rendered board is not backdrop (with the trails of the sprites '
              'behind the first drape) + things in z-order')
      if tabulate.hidden_image(eng, False) != hidden0:
        _fail(where + ': state outside the curtains changed (an entity attribute, a Plot entry)')
      if eng.game_over:
        return

  for a in range(N_ACTIONS):
    for b in range(N_ACTIONS):
      follow(tabulate.clone_engine(probe), (a, b), 'opening {}{},'.format(a, b))
  live = [a for a in range(N_ACTIONS) if not any(model.ends[ch][a] for ch in order)]
  # every action repeated until whatever it moves has been once round the board: a thing that
  # stops at an edge instead of wrapping (or bounces) shows here, wherever it starts
  for a in live:
    follow(tabulate.clone_engine(probe), [a] * (max(H, W) + 2), 'action {} repeated,'.format(a))
  rng = np.random.RandomState(20260401)
  for w in range(WALKS):
    # (mostly actions that do not end the episode, so that walks get somewhere)
    seq = [int(rng.choice(live)) if live and rng.rand() < 0.97 else int(rng.randint(N_ACTIONS))
           for _ in range(WALK_FRAMES)]
    follow(tabulate.clone_engine(probe), seq, 'walk {},'.format(w))

  # ---- the description `gamespec.lower_shapes()` takes
  entities = []
  for ch in schedule:
    moves = any(model.drow[ch]) or any(model.dcol[ch])
    pays = any(r is not None for r in model.reward[ch])
    ends = [a for a in range(N_ACTIONS) if model.ends[ch][a]]
    if not (moves or pays or ends) and not is_sprite[ch]:
      entities.append(gamespec.EntityDesc(ch, 'fixed', group_of[ch], masks[ch], {}))
      continue
    signed = lambda d, n: d - n if d > n // 2 else d
    params = dict(op='shape',
                  drow=[signed(d, H) for d in model.drow[ch]],
                  dcol=[signed(d, W) for d in model.dcol[ch]],
                  rewards=[None if r is None else float(r) for r in model.reward[ch]],
                  quit_action=ends[0] if len(ends) == 1 else None,
                  quit_actions=ends, sprite=is_sprite[ch], visible=visible[ch])
    entities.append(gamespec.EntityDesc(ch, 'shape', group_of[ch], masks[ch], params))
  chars = sorted(set(order) | set(probe.backdrop.palette))
  desc = gamespec.GameDescription(H, W, chars, backdrop_art, entities, order)
  desc.action_set = actions
  return desc
