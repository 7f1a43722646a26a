"""Recognise a game of arbitrary Python classes as a SHAPE game (the Hello World kind).

The shape tier (`shapes.ShapeGame`, csrc/k_shape.hip) runs games whose things are rigid
shapes that translate cyclically by a per-action offset and interact with nothing: the
reference's examples/Hello World Example.ipynb (cell 3 `RollingDrape`, `SlidingSprite`; cell 4
`make_game()`, z-order '12@34').  Such a game has rows*cols positions PER THING, so its states
cannot be enumerated the way `tabulate.trace()` does; `campx_amd.rules` holds re-typed classes
that declare their offsets (`fused_rule()`), and until round 4 only those reached the device.

This module gives the notebook's OWN classes - any user classes of that kind - the same
`gamespec.GameDescription` by watching them on the generic tier (`engine.Engine` with
`batch=None`, the reference's execution model, campx/engine.py:114-324):

1. from the state `its_showtime()` leaves, each of the five actions is played once on a deep
   copy; for every thing the curtain (a drape) or position (a sprite, campx/things.py:294)
   after the frame must be the one before it rolled by some (rows, cols) offset, and the
   reward it added (campx/plot.py:186-211; at most one `add_reward` per thing and frame) and
   whether it ended the episode (discount 0.0 only) are noted per (thing, action) - the probe's
   Plot records WHO called;
2. that model - offsets, rewards summed in update order as `r + total`, termination, and the
   reference renderer's treatment of sprites painted before the first drape, which are written
   into the backdrop for good (campx/rendering.py:128,150: trails) - must then predict every
   frame of a set of walks on the generic tier: all 25 two-action openings, every action
   repeated max(rows, cols) + 2 times (once round the board: wrap-around is exercised for every
   direction) and `WALKS` random walks of `WALK_FRAMES` frames: every thing's curtain /
   position, the reward bit for bit,
   discount, game-over and the rendered board.  Nothing else a frame can read may change
   (`tabulate.hidden_image`: entity attributes, the Plot's entries), the z-order must stay put, and
   the frame number must not be read.

A game that passes is a shape game AS FAR AS THOSE FRAMES SHOW - the check is a sample, not an
enumeration (a thing that behaves differently only in a configuration no walk reaches would
pass); `RecogniseError` (a ValueError) names the first frame that contradicts the model.
Host logic only; no GPU.
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
      _NOW[0] = self.character
      try:
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

  _WATCHED_PLOTS[base] = WatchedPlot
  return WatchedPlot


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
  `rules.SlidingSprite` / `FixedDrape`, or `RecogniseError`.  `engine` is not touched."""
  if engine.backdrop is None:
    raise ValueError('the Engine has no Backdrop yet')
  H, W = engine.rows, engine.cols
  if engine.hidden_performance is not None or engine.hidden_penalty is not None:
    _fail('hidden performance is not offered on the shape tier')
  actions = detect_actions(engine) if actions is None else list(actions)
  if len(actions) != N_ACTIONS:
    raise ValueError('exactly {} actions are needed'.format(N_ACTIONS))

  probe = tabulate.clone_engine(engine)
  probe._batch, probe._device, probe._fused = None, None, None
  probe._the_plot.__class__ = _watched_plot(tabulate.probe_plot_class(type(probe._the_plot)))
  for ent in probe.things.values():
    try:
      ent.__class__ = _watched_class(type(ent))
    except TypeError as e:
      _fail('{!r}: its class cannot be watched ({})'.format(ent.character, e))
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

  def play(eng, a):
    del _CALLS[:]
    obs, reward, discount = eng.play(copy.deepcopy(actions[a]))
    if tabulate.FRAME_READS[0] != reads0:
      _fail('the game reads the_plot.frame')
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

  # ---- 2. the model predicts the generic tier, frame by frame
  def follow(eng, seq, what):
    model.reset()
    for t, a in enumerate(seq):
      obs, reward, discount, _ = play(eng, a)
      want_reward, want_discount = model.step(a)
      where = '{} frame {} (action {})'.format(what, t, a)
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
