"""Tabulate a game made of arbitrary Python `update()` classes for the HIP table kernels.

The fused tier's kernels do not need rules: their update pass is a lookup in a
(cell, action) table (one moving thing) or a (cell, ..., cell, action) table (two to
four), and their render is "static scenery + the moving things' cells"
(csrc/k_update.hip, k_render.hip, k_step.hip).  For the rule classes in
`campx_amd.rules` those tables are built on the device by running the rule interpreter.
For anything else - the unmodified `examples/boat_race.py` classes, the Demo notebooks'
classes, a user's own Drapes and Sprites - this module builds the same tables on the
HOST by running the user's own code on the generic tier (`engine.Engine` with
`batch=None`: the reference's execution model, campx/engine.py:114-324) over every state
the game can reach:

    breadth-first from the state `its_showtime()` leaves (campx/engine.py:487-544), for
    each of the five actions: deep-copy the engine (entities must be deep-copyable,
    campx/things.py:38), `play(action)`, read the entities' curtains / positions, the
    Plot's reward / discount / game-over, and the rendered board.

A state is identified by everything a frame can READ: the byte image of every curtain and
sprite position, the z-order, and `hidden_image()` - every other attribute of every entity and of
the Backdrop, the Plot's user entries and attributes (campx/plot.py:29: the Plot is a dict for
exactly that), all by value (tensors included), and the frame number (plot.py:259-280) IF the
game reads it (the probe engine's Plot records reads of `frame` outside the priming frame; a
game that does is tabulated again with the frame number in the state).  A value that cannot be
imaged (a generator, a file, a lock) refuses the game.  So a bounded counter, a cooldown or a
time limit multiplies the states and is tabulated exactly; an unbounded one is refused - after
`MAX_HIDDEN_VALUES` distinct values of one quantity, or `MAX_PLAYS` frames - with a message
that names what keeps growing.

Afterwards the moving things are the entities whose image differs between two reached states;
everything else is scenery.  Whatever else tells two reached states with the same curtains
apart - WHICH z-order is in force for a game that re-orders its things (`Plot.change_z_order`,
campx/plot.py:121-159, applied by campx/engine.py:242-281), and the hidden values that are not
themselves functions of the curtains - is the "mode", tabulated like the cell of one more moving
thing that is never painted (the kernels index their tables by up to four cells and care about
nothing else).  What an order changes on the screen - which of two things on one cell shows - is
already in the table entries' "is the character its cell shows" bits.  The tabulation is exact
under conditions that are CHECKED while tabulating (TabulationError otherwise):

* every tracked thing occupies exactly one cell in every reached state - or NONE: a drape
  whose curtain is empty (a key that was picked up, a door that opened) and a sprite that
  is not `visible` (campx/things.py:294-296, 391-392; engine.py:314 skips it) are "absent",
  tabulated as standing on a cell index the thing never occupies and never shown.  A drape
  that covers SEVERAL cells which come and go (coins taken one by one; campx/things.py:161-262
  sets no one-cell limit) is described as one PIECE per cell it ever covers - on its cell while
  the curtain has it, absent otherwise (round 6, `TracedGame.piece_cell`; `_finish` decides how
  pieces reach the kernels: as tracked things, as a 16-bit mask per state, or folded into
  variants of the scenery).  There are at most
  four tracked things -
  the mode counts as one, and it has at most rows*cols values (beyond either: the game runs
  from its STATE table, the wide tier, which only needs the state count to fit) -,
  the board has at most 128 cells (one mover and one mode: 1 024, the wide tier,
  csrc/k_wide.hip) and 16 characters;
* a Backdrop whose curtain changes (a `Backdrop.update()` of its own, campx/things.py:103-148; a
  sprite painted into the backdrop, campx/rendering.py:128,150) is described the same way: one
  piece per (cell, character) it ever shows beyond its first picture, painted on the backdrop
  itself, behind every thing (round 6, `TracedGame.in_backdrop`; such a game runs from its state
  table) - or, more than sixteen of them, its pictures become the scenery's variants; no reached z-order changes how the SCENERY paints (two overlapping static drapes
  swapping places);
* every rendered board equals "backdrop, then things in the state's z-order" computed from
  the cells alone - which also yields whether a moving thing is the character its cell shows;
* at most fifteen distinct discounts other than the default - 1.0, or 0.0 on the frame
  `terminate_episode()` was called - are reported (`change_default_discount`,
  `terminate_episode(d)`: campx/plot.py:161-184, 232-257; the tables carry a 4-bit code).

... and under one that is only SPOT-CHECKED: that the game keeps no state where `hidden_image()` does
not look - a module global it mutates, a closure, a random number generator.  When a state is
reached again over a different history, every action is replayed from that second engine and
must reproduce the tabulated next state, reward, discount, game-over and board; that catches
such state only if it shows within one frame of one second arrival (a `random` call does; a
global counter that matters fifty frames later does not).  Games are deterministic functions of
their entities and their Plot in every example of the reference; one that is not must not be
handed to a batched Engine.  One such route is not left to the spot check, because it would not
be caught but tabulated wrong: a LIVE Sprite / Drape / Backdrop / Engine / Plot that a class reaches
through a module global, a closure variable, a default argument or a class attribute.  The walks
run on deep copies of the engine, which such a reference does not follow - the object would
seem to stand still - so `reached_behind_the_engine()` looks for it in the code of the game's
classes (and the user functions and classes they name) before anything is walked, and the game
is refused by the name of the route.  (A partner kept in an INSTANCE attribute is copied along
with its owner and tabulates correctly.)

Host logic only (numpy + the generic tier): runs without a GPU.  `fused.FusedGame`
uploads the result.
"""

import collections
import copy
import hashlib
import os
import sys

import numpy as np
import torch

from . import chance
from . import gamespec
from . import things as _things

N_ACTIONS = gamespec.N_ACTIONS
# Upper bound on generic-tier play() calls one tabulation may spend (each costs a
# deep copy + a frame of Python, ~2 ms): beyond it the game is refused with a pointer to
# campx_amd.rules, whose tables are built on the device.
MAX_PLAYS = 60000
# Most distinct values ONE hidden quantity (a Plot entry, an entity attribute, the frame number)
# may take before the game is refused: a counter that is never reset would otherwise cost all of
# MAX_PLAYS (two minutes) to find out.  A time limit of up to this many frames is fine.
MAX_HIDDEN_VALUES = 1024
# Largest dense (cell, ..., cell, action) table built on the host (entries); games beyond it
# run from their state table (the wide tier).
DENSE_MAX_ENTRIES = 8 << 20


_STDLIB = os.path.dirname(os.__file__)


class TabulationError(ValueError):
  pass


def default_actions():
  """The reference's action format (examples/boat_race.py:26, 154-184): one-hot float
  vectors in the order left, right, up, down, stay."""
  return [torch.eye(N_ACTIONS, dtype=torch.float32)[a].clone() for a in range(N_ACTIONS)]


def _fail(msg):
  raise TabulationError('cannot tabulate this game for the HIP tier: ' + msg)


def _image(engine):
  """(per-thing byte images by ascending character, backdrop image, z-order string)."""
  parts = []
  for ch in sorted(engine.things.keys()):
    ent = engine.things[ch]
    if isinstance(ent, _things.Sprite):
      parts.append(bytes((ent.position.row & 0xff, ent.position.col & 0xff,
                          1 if ent.visible else 0)))
    else:
      parts.append(ent.curtain.detach().to(torch.uint8).numpy().tobytes())
  backdrop = engine.backdrop.curtain.detach().to(torch.int64).numpy().tobytes()
  return tuple(parts), backdrop, ''.join(engine.things.keys())


class _Unimageable(Exception):
  """Something a frame can read that is not plain data (raised with its path)."""


# attributes of an entity that `_image()` (or nothing at all: constants) already covers
_CORE_ATTRS = {
    'drape': ('_curtain', '_character'),
    'sprite': ('_corner', '_character', '_position', '_visible'),
    'backdrop': ('_curtain', '_palette'),
    'plot': ('_frame', '_update_group', '_engine_directives'),
}
_MISSING = ('missing',)


def _plain(x, path, depth, open_ids):
  """A hashable image of plain data: numbers, strings, tensors and arrays BY VALUE, containers
  and objects (class name + attributes) of those.  Functions, classes and modules are taken by
  name (they are code, not state).  Anything else - a generator, an open file, a lock - cannot
  be compared between two frames and raises `_Unimageable`."""
  if x is None or isinstance(x, (bool, int, float, complex, str, bytes)):
    return (type(x).__name__, x) if not (isinstance(x, float) and x != x) else ('float', 'nan')
  if torch.is_tensor(x):
    a = x.detach().cpu().numpy()
    return ('tensor', str(a.dtype), a.shape, a.tobytes())
  if isinstance(x, (np.ndarray, np.generic)):
    a = np.asarray(x)
    return ('array', str(a.dtype), a.shape, np.ascontiguousarray(a).tobytes())
  if depth > 8:
    raise _Unimageable(path + ' (nested more than 8 deep)')
  if id(x) in open_ids:
    return ('cycle',)
  if isinstance(x, (type, type(_plain), type(len), type(np))) or callable(x) and hasattr(x, '__qualname__'):
    return ('ref', getattr(x, '__module__', None), getattr(x, '__qualname__', getattr(x, '__name__', '?')))
  open_ids = open_ids | {id(x)}
  if isinstance(x, (list, tuple)):
    return (type(x).__name__,) + tuple(_plain(v, '{}[{}]'.format(path, i), depth + 1, open_ids)
                                        for i, v in enumerate(x))
  if isinstance(x, (set, frozenset)):
    return ('set',) + tuple(sorted((_plain(v, path + '{..}', depth + 1, open_ids) for v in x), key=repr))
  if isinstance(x, dict):
    items = [(_plain(k, path + '{key}', depth + 1, open_ids),
              _plain(v, '{}[{!r}]'.format(path, k), depth + 1, open_ids)) for k, v in x.items()]
    return ('dict',) + tuple(sorted(items, key=repr))
  if hasattr(x, '__dict__'):
    items = tuple((k, _plain(v, '{}.{}'.format(path, k), depth + 1, open_ids))
                  for k, v in sorted(vars(x).items()))
    return ('object', type(x).__module__, type(x).__qualname__) + items
  raise _Unimageable('{} (a {})'.format(path, type(x).__name__))


def hidden_image(engine, with_frame):
  """Everything a frame can read that is NOT a curtain, a sprite's position / visibility or the
  z-order (those are `_image()`): every other attribute of every entity and of the Backdrop,
  the Plot's user entries (campx/plot.py:29 - the Plot is a dict for exactly that; the message
  log, which frames only append to, is left out) and attributes, and - `with_frame` - the frame
  number (campx/plot.py:259-280).  A tuple of (path, image) pairs, sorted by path."""
  plot = engine.the_plot
  found = []

  def attrs(obj, prefix, kind):
    for name, value in vars(obj).items():
      if name not in _CORE_ATTRS[kind]:
        found.append((prefix + '.' + name, value))

  for ch in sorted(engine.things.keys()):
    ent = engine.things[ch]
    attrs(ent, 'things[{!r}]'.format(ch), 'sprite' if isinstance(ent, _things.Sprite) else 'drape')
  attrs(engine.backdrop, 'backdrop', 'backdrop')
  for key, value in plot.items():
    if key != plot.LOG_KEY:
      found.append(('the_plot[{!r}]'.format(key), value))
  attrs(plot, 'the_plot', 'plot')
  if with_frame:
    found.append(('the_plot.frame', plot._frame))
  out = []
  for path, value in found:
    try:
      out.append((path, _plain(value, path, 0, frozenset())))
    except _Unimageable as e:
      _fail('{} is not plain data (numbers, strings, tensors, containers and objects of '
            'those): the tabulator cannot tell whether two frames that look the same ARE the '
            'same state'.format(e))
  out.sort(key=lambda kv: kv[0])
  return tuple(out)


class _FrameWasRead(Exception):
  pass


FRAME_READS = [0]     # bumped by every read of `Plot.frame` on a probe engine


_PROBE_PLOTS = {}


def probe_plot_class(base):
  """`base` (the engine's Plot class) with a `frame` that notices being read: a game whose
  update() looks at the frame number (a time limit) has the frame number in its state."""
  if base in _PROBE_PLOTS:
    return _PROBE_PLOTS[base]
  if getattr(base, '_campx_probe', False):
    return base

  class ProbePlot(base):
    _campx_probe = True
    __slots__ = ()

    @property
    def frame(self):
      FRAME_READS[0] += 1
      return self._frame

    @frame.setter
    def frame(self, val):
      base.frame.fset(self, val)

  _PROBE_PLOTS[base] = ProbePlot
  return ProbePlot


def reward_f32(reward):
  if reward is None:
    return np.float32(np.nan)
  if torch.is_tensor(reward):
    return np.float32(reward.detach().to(torch.float32).item())
  return np.float32(reward)


class _Edge(object):
  __slots__ = ('next', 'reward', 'discount', 'over', 'board')

  def __init__(self, nxt, reward, discount, over, board):
    self.next, self.reward, self.discount, self.over, self.board = (
        nxt, reward, discount, over, board)

  def same(self, other):
    return (self.next == other.next and self.over == other.over and
            self.discount == other.discount and self.board == other.board and
            np.array_equal(np.array([self.reward]).view(np.uint32),
                           np.array([other.reward]).view(np.uint32)))


class TracedGame(object):
  """The tabulated update pass of one game (plain numpy; see `trace`).

  Attributes:
    rows, cols, chars (ascending), z_order (characters back to front), backdrop
      (uint8 [H, W] character codes);
    movers: characters of the moving things in update-schedule order (K of them);
    statics: [(character, uint8 [H, W] mask)] of the things that never change, in
      update-schedule order;
    init_cells: the movers' cells after `its_showtime()`;
    n: (H*W)^K * 5 table entries, index ((cell_0 * HW + cell_1) ...) * 5 + action;
    next_cells uint16 [K, n], visible uint8 [K, n], reward float32 [n] (NaN = None),
    done uint8 [n], discount float32 [n], dcode uint8 [n] (0 = the default discount, else
    an index into `discount_list`), perf int8 [n], reached bool [n] (entries the game can
    get to; the others are self-loops that pay nothing);
    piece_cell: per mover None, or - a drape that covers several cells which come and go is
      tracked as one mover per cell, all of its character - the cell this piece stands on
      whenever it is on the board;
    in_backdrop: per mover, True for a piece of a Backdrop that changes (one per (cell,
      character) it shows other than at the start; they come after every other mover and are
      painted on the backdrop itself, behind every thing);
    pieces_as_mask: the pieces are handed to the state-table tier as one 16-bit mask per state
      (`CampxWideSpec.n_pieces`: bit p = piece p shows) instead of a tracked thing each - up to
      sixteen of them beside up to seven ordinary movers; nothing else about the game changes;
    variants, variant_masks, st_variant: the scenery in VARIANTS - the Backdrop's pictures (uint8
      [H, W] each, the first = `backdrop`), beside each the curtains of the several-cell drapes
      that are part of the scenery's pictures ({character: uint8 [H, W]}), and per state which
      picture shows (one picture, all zeros, for a game whose scenery never changes);
    absent_cells: per mover, the tracked values that stand for "not on the board" (an empty
      curtain, an invisible sprite) - cell indices the thing never occupies; usually empty;
    mode_orders: the z-orders the game reaches (lists of characters back to front; the
      first is `z_order`, the one after `its_showtime()`).  With more than one, the tables
      track one more "thing" after the movers - K = len(movers) + 1 = `n_tracked` - whose
      "cell" is the index of the order in force and which is never visible;
    n_states, n_plays: size of the reachable state space and what tabulating it cost.
  """

  @property
  def n_tracked(self):
    """Things the tables are indexed by: the movers, plus the z-order mode if there is one."""
    return len(self.movers) + (1 if len(self.mode_orders) > 1 else 0)

  def done_bytes(self):
    """uint8 [n]: bit 0 done, bits 4-7 the discount code - CampxTransition.done and the
    `done` argument of campx_pair_table_pack."""
    return (self.done | (self.dcode << 4)).astype(np.uint8)

  def trace_bytes(self):
    """uint8 [K, n]: cell | visible << 7 - the kernels' trace format (CampxOutputs.trace)."""
    return (self.next_cells | (self.visible << 7)).astype(np.uint8)

  def cells_of(self, index):
    HW, K = self.rows * self.cols, self.n_tracked
    rest, cells = index // N_ACTIONS, []
    for _ in range(K):
      cells.append(rest % HW)
      rest //= HW
    return tuple(reversed(cells))

  def index_of(self, cells, action=0):
    HW, idx = self.rows * self.cols, 0
    for c in cells:
      idx = idx * HW + int(c)
    return idx * N_ACTIONS + action

  def is_absent(self, k, cell):
    """Mover k's tracked value `cell` stands for "nowhere" (an empty curtain, an invisible
    sprite): an index it never occupies while on the board."""
    return int(cell) in self.absent_cells[k]

  def model_board(self, cells, movers=True, variant=0):
    """The flat board (character codes) when the movers stand at `cells` (followed by the
    z-order mode, if the game has more than one): backdrop, then every thing in that
    z-order (campx/engine.py:306-324).  `movers=False`: the scenery alone.  `variant`: which
    of the Backdrop's pictures (`variants`; 0 = the first, `backdrop`) lies beneath."""
    variants = getattr(self, 'variants', None)
    board = (variants[variant] if variants else self.backdrop).copy().reshape(-1)
    static = dict(self.statics)
    masks = getattr(self, 'variant_masks', None)
    if masks:
      static.update(masks[variant])        # the several-cell drapes as this picture has them
    where = {}                 # character -> the cells its movers (pieces, for a many-cell drape) stand on
    in_backdrop = getattr(self, 'in_backdrop', None) or [False] * len(self.movers)
    for k, (ch, c) in enumerate(zip(self.movers, cells)):
      if in_backdrop[k]:       # what a changing Backdrop shows: on the backdrop itself, behind every thing
        if movers and not self.is_absent(k, c):
          board[int(c)] = ord(ch)
        continue
      where.setdefault(ch, [])
      if not self.is_absent(k, c):
        where[ch].append(int(c))
    mode = cells[len(self.movers)] if len(self.mode_orders) > 1 else 0
    for ch in self.mode_orders[mode]:
      if ch in where:
        if movers:
          for c in where[ch]:
            board[c] = ord(ch)
      else:
        board[static[ch].reshape(-1) != 0] = ord(ch)
    return board.reshape(self.rows, self.cols)


def _copy_tensor(x, memo):
  """deepcopy of a plain CPU tensor, as torch's own `Tensor.__deepcopy__` does it - a clone of
  the STORAGE, shared by every tensor that shared the original (the reference's renderer
  relies on such sharing: `board.set_(curtain)`, campx/rendering.py:128) - minus the checks
  for the kinds of tensor a game's state never is; those take the slow path."""
  if (type(x) is not torch.Tensor or x.requires_grad or x.device.type != 'cpu' or x.is_sparse
      or x.is_quantized or x.grad is not None):
    return x.__deepcopy__(memo)
  storage = x.untyped_storage()
  shared = memo.setdefault('torch', {})          # (the key torch's own storage copy uses)
  fresh = shared.get(storage._cdata)
  if fresh is None:
    fresh = shared[storage._cdata] = storage.clone()
  return torch.empty(0, dtype=x.dtype).set_(fresh, x.storage_offset(), x.size(), x.stride())


def clone_engine(engine):
  """`copy.deepcopy(engine)` with the lean tensor copy above (a tabulation makes thousands of
  copies of an engine that holds a dozen small tensors: 70 % of its time was here)."""
  dispatch = copy._deepcopy_dispatch
  had = dispatch.get(torch.Tensor)
  dispatch[torch.Tensor] = _copy_tensor
  try:
    return copy.deepcopy(engine)
  finally:
    if had is None:
      del dispatch[torch.Tensor]
    else:
      dispatch[torch.Tensor] = had


class _NoFingerprint(Exception):
  pass


def _feed_code(h, code):
  h.update(b'code(')
  h.update(code.co_code)
  h.update(repr((code.co_names, code.co_varnames, code.co_argcount, code.co_kwonlyargcount,
                 code.co_freevars, code.co_cellvars, code.co_flags)).encode())
  for const in code.co_consts:
    if hasattr(const, 'co_code'):
      _feed_code(h, const)                             # nested functions, comprehensions
    else:
      h.update(repr(const).encode())
    h.update(b',')
  h.update(b')')


def _code_names(code, into):
  """Every name the code object - and the code objects nested in it - loads by name."""
  into.update(code.co_names)
  for const in code.co_consts:
    if hasattr(const, 'co_code'):
      _code_names(const, into)
  return into


def _is_library_module(mod):
  """A module nobody edits between two set-ups of a game: the standard library, installed
  packages (torch, numpy ...), this package."""
  name = getattr(mod, '__name__', '')
  if name.split('.')[0] in ('campx_amd', 'campx', 'builtins', 'torch', 'numpy'):
    return True
  path = getattr(mod, '__file__', None)
  if path is None:
    return name in sys.builtin_module_names or name != '__main__'
  return 'site-packages' in path or 'dist-packages' in path or path.startswith(_STDLIB)


def _feed_function(h, fn, depth, seen):
  """A function by what it does AND by what it reads: its code, defaults, closure, and the
  VALUES behind the global names it loads (`QUARTERED_MOVEMENT_PENALTY`,
  examples/boat_race.py:22,76): plain data by value, helper functions and classes of user
  modules by their own code, user modules through the attributes the code names; library
  modules (torch, numpy, the standard library) by name."""
  code = fn.__code__
  if id(fn) in seen:
    h.update(b'<again>')
    return
  seen = seen | {id(fn)}
  _feed_code(h, code)
  h.update(b'defaults')
  _feed(h, fn.__defaults__, depth + 1, seen)
  _feed(h, fn.__kwdefaults__, depth + 1, seen)
  h.update(b'closure')
  for cell in (fn.__closure__ or ()):
    try:
      inside = cell.cell_contents
    except ValueError:                           # an empty cell
      h.update(b'<empty>')
      continue
    if isinstance(inside, type):                 # (`__class__`, for super(): by name)
      h.update('{}.{}'.format(inside.__module__, inside.__qualname__).encode())
    else:
      _feed(h, inside, depth + 1, seen)
  h.update(b'globals')
  names = sorted(_code_names(code, set()))
  space = fn.__globals__
  for name in names:
    if name not in space:
      continue                                   # an attribute name, a builtin
    value = space[name]
    h.update(name.encode() + b'=')
    if isinstance(value, type(sys)):
      h.update(('module ' + value.__name__).encode())
      if not _is_library_module(value):
        for attr in names:                       # `config.PENALTY`: the attributes it names
          if hasattr(value, attr):
            h.update(attr.encode() + b':')
            _feed(h, getattr(value, attr), depth + 1, seen)
    else:
      _feed(h, value, depth + 1, seen)


def _feed(h, x, depth=0, seen=frozenset()):
  """Hash plain data - numbers, strings, tensors, arrays, containers of those, objects
  through their class and __dict__, functions and classes through `_feed_function` - into h;
  anything else has no fingerprint."""
  if depth > 8:
    raise _NoFingerprint()
  if x is None or isinstance(x, (bool, int, float, str, bytes)):
    h.update(repr(x).encode())
  elif torch.is_tensor(x):
    a = x.detach().cpu().numpy()
    h.update(str((a.dtype, a.shape)).encode())
    h.update(a.tobytes())
  elif isinstance(x, (np.ndarray, np.generic)):
    x = np.asarray(x)
    h.update(str((x.dtype, x.shape)).encode())
    h.update(np.ascontiguousarray(x).tobytes())
  elif isinstance(x, (list, tuple)):
    h.update(b'[')
    for item in x:
      _feed(h, item, depth + 1, seen)
      h.update(b',')
    h.update(b']')
  elif isinstance(x, (set, frozenset)):
    h.update(b'<')
    for item in sorted(x, key=repr):
      _feed(h, item, depth + 1, seen)
      h.update(b',')
    h.update(b'>')
  elif isinstance(x, dict):
    h.update(b'{')
    for key in sorted(x, key=repr):
      _feed(h, key, depth + 1, seen)
      h.update(b':')
      _feed(h, x[key], depth + 1, seen)
      h.update(b',')
    h.update(b'}')
  elif isinstance(x, type):
    # a class by what it DOES, not by where it lives (ids are recycled when classes defined
    # inside functions are collected): name + the code, defaults, closure and globals of every
    # function it and its bases define, down to this package's own base classes
    if id(x) in seen:
      h.update(b'<again>')
      return
    seen = seen | {id(x)}
    for klass in x.__mro__:
      h.update('class {}.{};'.format(klass.__module__, klass.__qualname__).encode())
      if (klass.__module__.startswith(('campx_amd.', 'builtins', 'abc', 'collections', 'typing'))
          and '<locals>' not in klass.__qualname__):
        continue
      for name in sorted(vars(klass)):
        if name.startswith('_abc_'):
          continue                                     # (abc's bookkeeping)
        member = vars(klass)[name]
        fn = getattr(member, '__func__', member)       # static / class methods
        fn = getattr(fn, 'fget', fn)                   # properties
        if getattr(fn, '__code__', None) is None:
          if not name.startswith('__'):
            _feed(h, name, depth + 1, seen)
            h.update(b'=')
            _feed(h, member, depth + 1, seen)          # a class attribute: plain data or nothing
          continue
        h.update(b'def ' + name.encode() + b':')
        _feed_function(h, fn, depth + 1, seen)
  elif getattr(x, '__code__', None) is not None and hasattr(x, '__globals__'):
    h.update(b'def:')
    _feed_function(h, x, depth + 1, seen)
  elif isinstance(x, type(sys)):
    if not _is_library_module(x):
      raise _NoFingerprint()                           # a user module as a value: no telling
    h.update(('module ' + x.__name__).encode())
  elif isinstance(x, type(len)):                       # builtin functions
    h.update(('builtin ' + getattr(x, '__qualname__', repr(x))).encode())
  elif hasattr(x, '__dict__') and not callable(x):
    if id(x) in seen:
      h.update(b'<again>')
      return
    seen = seen | {id(x)}
    _feed(h, type(x), depth + 1, seen)
    _feed(h, vars(x), depth + 1, seen)
  else:
    raise _NoFingerprint()


# ---- live game objects reached behind the engine's back ------------------------------------
# `update()` is handed everything it may look at (campx/engine.py:200-204).  A class that reaches
# a live Engine, Plot, Sprite, Drape or Backdrop some other way - a module global, a closure
# variable, a default argument: `REGISTRY['A'].curtain` - reads state the tabulators never see
# change: they work on deep copies of the engine, which such a reference does not follow, so the
# table would be built as if the object stood still.  That is checked statically, before any walk.

def _global_loads(code, into):
  """Names the code object (and those nested in it) loads as GLOBALS - not attribute names."""
  import dis
  for ins in dis.get_instructions(code):
    if ins.opname in ('LOAD_GLOBAL', 'LOAD_NAME'):
      into.add(ins.argval)
  for const in code.co_consts:
    if hasattr(const, 'co_code'):
      _global_loads(const, into)
  return into


def _attribute_names(code, into):
  into.update(code.co_names)
  for const in code.co_consts:
    if hasattr(const, 'co_code'):
      _attribute_names(const, into)
  return into


def _live_in_value(x, live, depth, seen):
  """A description of the first live game object inside `x` (containers, plain objects, user
  functions and classes are followed), or None."""
  if depth > 5 or x is None or isinstance(x, (bool, int, float, str, bytes)) or torch.is_tensor(x) or \
      isinstance(x, (np.ndarray, np.generic)) or id(x) in seen:
    return None
  if isinstance(x, live):
    return 'a live {}'.format(type(x).__name__)
  source = chance.named_source(x)      # a generator object, a bound method of one, a clock
  if source:
    return source
  seen.add(id(x))
  if isinstance(x, (list, tuple, set, frozenset)):
    for item in x:
      found = _live_in_value(item, live, depth + 1, seen)
      if found:
        return found
    return None
  if isinstance(x, dict):
    for key, item in x.items():
      found = _live_in_value(key, live, depth + 1, seen) or _live_in_value(item, live, depth + 1, seen)
      if found:
        return found
    return None
  if isinstance(x, type(sys)) or isinstance(x, type(len)):
    return None
  if isinstance(x, type):
    return _live_in_class(x, live, depth + 1, seen)
  if getattr(x, '__code__', None) is not None and hasattr(x, '__globals__'):
    mod = sys.modules.get(getattr(x, '__module__', None) or '')
    if mod is not None and _is_library_module(mod):
      return None
    return _live_in_function(x, live, depth + 1, seen)
  if hasattr(x, '__func__'):                       # a bound method: its object and its function
    return _live_in_value(getattr(x, '__self__', None), live, depth + 1, seen) or \
        _live_in_value(x.__func__, live, depth + 1, seen)
  if hasattr(x, '__dict__') and not callable(x):
    return _live_in_value(vars(x), live, depth + 1, seen)
  return None


def _live_in_function(fn, live, depth, seen):
  code = fn.__code__
  if id(code) in seen:
    return None
  seen.add(id(code))
  for what, value in (('a default argument', fn.__defaults__), ('a default argument', fn.__kwdefaults__)):
    found = _live_in_value(value, live, depth + 1, seen)
    if found:
      return '{} through {} of {}()'.format(found, what, fn.__name__)
  for name, cell in zip(code.co_freevars, fn.__closure__ or ()):
    try:
      inside = cell.cell_contents
    except ValueError:
      continue
    if isinstance(inside, type) and name == '__class__':
      continue
    found = _live_in_value(inside, live, depth + 1, seen)
    if found:
      return '{} through the closure variable {!r} of {}()'.format(found, name, fn.__name__)
  space = fn.__globals__
  attrs = None
  for name in sorted(_global_loads(code, set())):
    if name not in space:
      continue
    value = space[name]
    if isinstance(value, type(sys)):
      if _is_library_module(value):
        continue
      attrs = _attribute_names(code, set()) if attrs is None else attrs
      for attr in sorted(attrs):                   # `config.REGISTRY`: the attributes it names
        if hasattr(value, attr):
          found = _live_in_value(getattr(value, attr), live, depth + 1, seen)
          if found:
            return '{} through {}.{} (named by {}())'.format(found, name, attr, fn.__name__)
      continue
    attrs = _attribute_names(code, set()) if attrs is None else attrs
    found = chance.named_source(value, attrs) or _live_in_value(value, live, depth + 1, seen)
    if found:
      return '{} through the module global {!r} (named by {}())'.format(found, name, fn.__name__)
  return None


def _live_in_class(klass, live, depth, seen):
  if id(klass) in seen and depth > 0:
    return None
  seen.add(id(klass))
  for base in klass.__mro__:
    if base.__module__.split('.')[0] in ('campx_amd', 'campx', 'builtins', 'abc', 'collections', 'typing'):
      continue
    for name in sorted(vars(base)):
      member = vars(base)[name]
      fn = getattr(member, '__func__', member)
      fn = getattr(fn, 'fget', fn)
      if getattr(fn, '__code__', None) is not None and hasattr(fn, '__globals__'):
        found = _live_in_function(fn, live, depth + 1, seen)
        if found:
          return found
      elif not name.startswith('__'):              # a class attribute (deep copies share classes)
        found = _live_in_value(member, live, depth + 1, seen)
        if found:
          return '{} through the class attribute {}.{}'.format(found, base.__name__, name)
  return None


def reached_behind_the_engine(engine):
  """None, or what to tell the user: the first live game object that the code of one of the
  game's classes reaches through a module global, a closure variable or a default argument."""
  from . import plot as _plot
  live = (_things.Sprite, _things.Drape, _things.Backdrop, _plot.Plot, type(engine))
  seen = set()
  for ent in list(engine.things.values()) + [engine.backdrop]:
    if ent is None:
      continue
    found = _live_in_class(type(ent), live, 0, seen)
    if found and not found.startswith('a live '):
      return ('{!r} ({}): its code names {}.  A batched Engine runs a game from a table of what its '
              'classes do in each state, which a game of chance (or of the clock) does not have; run '
              'it on the generic tier (batch=None)'.format(
                  getattr(ent, 'character', 'backdrop'), type(ent).__name__, found))
    if found:
      return ('{!r} ({}): its code reaches {}.  update() is handed everything it may read '
              '(layers, all_things, the_plot); state reached any other way is invisible to the '
              'tabulation'.format(getattr(ent, 'character', 'backdrop'), type(ent).__name__, found))
  return None


def fingerprint(engine, actions):
  """A key under which the tabulation of a set-up engine can be reused: every entity's class
  (by the code of its methods) and attributes, the backdrop, the update groups and z-order, the hidden-
  performance declarations, the action set.  None when something in there is not plain data
  (the game is then tabulated afresh every time).  A method is hashed with the VALUES of the
  module-level globals it names (and, through a user module it names, that module's
  attributes it names) - `boat_race.QUARTERED_MOVEMENT_PENALTY = -0.5` between two
  `make_game()` calls is another game -, its default arguments and its closure; what the
  fingerprint cannot see is state reached through a call into code it does not walk (a global
  read by a function of a library module).  `trace(..., cache=False)` tabulates afresh."""
  h = hashlib.sha1()
  try:
    _feed(h, (engine.rows, engine.cols, list(engine.things.keys())))
    groups = engine._update_groups
    if isinstance(groups, dict):
      groups = [(name, groups[name]) for name in sorted(groups.keys())]
    _feed(h, [(name, [ent.character for ent in members]) for name, members in groups])
    for ch, ent in engine.things.items():
      _feed(h, ch)
      _feed(h, ent)
    _feed(h, engine.backdrop)
    _feed(h, (engine.hidden_performance, engine.hidden_penalty))
    _feed(h, actions)
  except _NoFingerprint:
    return None
  return h.hexdigest()


_CACHE = collections.OrderedDict()
CACHE_SIZE = 8


def trace(engine, actions=None, max_plays=MAX_PLAYS, cache=True):
  """Tabulate a set-up (not yet started) `Engine`; returns a `TracedGame`.

  `engine` itself is not touched: a deep copy of it is put through `its_showtime()` on
  the generic tier.  `actions`: the five objects handed to `play()` for action ids
  0..4 (default: the reference's one-hot float vectors).

  The reference's driver builds a new game per episode (`make_game()`,
  examples/reinforce.py:122): the tabulation of an engine whose entities, attributes and
  set-up are the same as an earlier one's is reused (`cache`; the `TracedGame` is shared and
  must be treated as read-only).
  """
  key = None
  if cache and engine.backdrop is not None:
    key = fingerprint(engine, default_actions() if actions is None else list(actions))
    if key is not None:
      # (... and the bounds that decide how pieces of the scenery are handed to the kernels)
      key = (key, max_plays, gamespec.WIDE_MAX_PIECES, gamespec.WIDE_MAX_VARIANTS, gamespec.PIECES_AS_THINGS_MAX)
      if key in _CACHE:
        _CACHE.move_to_end(key)
        return _CACHE[key]
  with chance.forbidden(TabulationError):     # (random numbers, clocks: refused by proof)
    game = _trace(engine, actions, max_plays)
  if key is not None:
    _CACHE[key] = game
    while len(_CACHE) > CACHE_SIZE:
      _CACHE.popitem(last=False)
  return game


LAST_WALK = ['']      # how the most recent tabulation was walked (tests, diagnostics)


def _trace(engine, actions, max_plays):
  """Many states per call where the game's classes allow it (tabulate_batched.py: the user's
  update() on lane tensors); else one frame of Python per state and action - without the frame
  number in the state first; when the game turns out to read `the_plot.frame`
  (campx/plot.py:259-280), again with it.  CAMPX_TABULATE=walk: never lane by lane; =batch:
  only lane by lane (the refusal is raised)."""
  behind = reached_behind_the_engine(engine)
  if behind:
    _fail(behind)
  mode = os.environ.get('CAMPX_TABULATE', 'auto')
  if mode != 'walk':
    from . import tabulate_batched
    from .lanes import CannotBatch
    try:
      # (on the host: the game's own tensors - constants its classes keep, the Backdrop - live
      # there, and half a million states take 15 s)
      game = tabulate_batched.trace(engine, actions, max_plays, device=None)
      LAST_WALK[0] = 'lanes: {} frames for {} states'.format(game.batched_frames, game.n_states)
      return game
    except CannotBatch as why:
      if mode == 'batch':
        raise TabulationError('not a game the many-states-per-call tabulator takes: {}'.format(why))
      LAST_WALK[0] = 'one frame per play (lanes: {})'.format(why)
    except TabulationError:
      raise
    except Exception as why:         # noqa: BLE001 - whatever the user's classes raise on lane tensors
      if mode == 'batch':
        raise
      LAST_WALK[0] = 'one frame per play (lanes: {}: {})'.format(type(why).__name__, str(why)[:200])
  try:
    return _trace_once(engine, actions, max_plays, with_frame=False)
  except _FrameWasRead:
    return _trace_once(engine, actions, max_plays, with_frame=True)


def _trace_once(engine, actions, max_plays, with_frame):
  if engine.backdrop is None:
    raise ValueError('the Engine has no Backdrop yet')
  H, W = engine.rows, engine.cols
  HW = H * W
  chars = sorted(set(engine.things.keys()) | set(engine.backdrop.palette))
  if HW > gamespec.WIDE_MAX_CELLS or H > 127 or W > 127:
    _fail('{}x{} board: more than {} cells (or 127 rows / columns)'.format(
        H, W, gamespec.WIDE_MAX_CELLS))
  if len(chars) > gamespec.MAX_LAYERS:
    _fail('more than {} characters'.format(gamespec.MAX_LAYERS))
  actions = default_actions() if actions is None else list(actions)
  if len(actions) != N_ACTIONS:
    raise ValueError('exactly {} actions are needed'.format(N_ACTIONS))

  probe = clone_engine(engine)
  probe._batch, probe._device, probe._fused = None, None, None
  probe._the_plot.__class__ = probe_plot_class(type(probe._the_plot))
  obs, _, _ = probe.its_showtime()
  if probe.game_over:
    _fail('the episode is over after its_showtime()')
  # (reads of the frame number during the priming frame are not held against the game: every
  # episode passes through it at frame 0, "if the_plot.frame == 0: set up" included)
  reads0 = FRAME_READS[0]

  # (the rule library's own classes keep nothing the state image below does not hold, and
  # call nothing that could - no globals, no RNG -, so their games skip the second-history
  # replays, which are half of a tabulation's frames)
  check_histories = not gamespec.is_rule_game(engine)
  things0, backdrop0, z0 = _image(probe)
  hidden0 = hidden_image(probe, with_frame)
  # state bookkeeping: a state is (curtains and positions, z-order, everything else a frame
  # can read)
  index_of = {(things0, z0, hidden0, backdrop0): 0}
  images = [things0]
  backdrops = [backdrop0]    # per state: the Backdrop's curtain (a Backdrop.update() may change it)
  orders = [z0]              # per state: the z-order in force (characters back to front)
  hiddens = [hidden0]        # per state: `hidden_image()`
  engines = [probe]          # an engine standing in that state, or None (only seen ended)
  second = {}                # state -> an engine that arrived there over another history
  boards = [obs.board.detach().to(torch.int64).numpy().astype(np.uint8).tobytes()]
  edges = {}
  queue = collections.deque([0])
  plays = [0]

  hidden_seen = collections.defaultdict(set)     # path -> the values it has taken

  def note_hidden(hid):
    for path, value in hid:
      seen = hidden_seen[path]
      seen.add(value)
      if len(seen) > MAX_HIDDEN_VALUES:
        too_many()

  def too_many():
    # name what keeps growing: the hidden values with the most distinct values so far
    growing = sorted(((len(v), path) for path, v in hidden_seen.items() if len(v) > 1),
                     reverse=True)
    what = ''
    if growing:
      what = (' The state includes ' +
              ', '.join('{} ({} different values so far)'.format(path, n)
                        for n, path in growing[:3]) +
              ': a counter or clock that is never reset makes every frame a new state.')
    _fail('more than {} generic-tier frames would be needed (the reachable state space is '
          'too large to tabulate on the host); express the game with campx_amd.rules, '
          'whose tables are built on the device.{}'.format(
              max_plays if plays[0] >= max_plays else plays[0], what))

  def step(eng, a):
    if plays[0] >= max_plays:
      too_many()
    plays[0] += 1
    obs, reward, discount = eng.play(copy.deepcopy(actions[a]))
    if not with_frame and FRAME_READS[0] != reads0:
      raise _FrameWasRead()
    things, backdrop, z = _image(eng)
    over = bool(eng.game_over)
    discount = float(np.float32(discount))
    board = obs.board.detach().to(torch.int64).numpy().astype(np.uint8).tobytes()
    return (things, z, hidden_image(eng, with_frame), backdrop), reward_f32(reward), discount, over, board

  while queue:
    s = queue.popleft()
    for a in range(N_ACTIONS):
      # (the last action is played on the state's own engine: nobody needs it afterwards)
      eng = clone_engine(engines[s]) if a < N_ACTIONS - 1 else engines[s]
      key, reward, discount, over, board = step(eng, a)
      t = index_of.get(key)
      if t is None:
        t = index_of[key] = len(images)
        images.append(key[0])
        orders.append(key[1])
        hiddens.append(key[2])
        backdrops.append(key[3])
        note_hidden(key[2])
        boards.append(board)
        engines.append(None)
      elif boards[t] != board:
        _fail('the same curtains rendered two different boards')
      edges[(s, a)] = _Edge(t, reward, discount, over, board)
      if not over:
        if engines[t] is None:
          engines[t] = eng
          queue.append(t)
        elif check_histories and t not in second:    # (every (s, a) is played once: another history)
          second[t] = eng

  # ---- the image IS the state: replay every action over a second history.  (A guard, not a
  # proof: it catches state kept where `hidden_image()` does not look - module globals, closures, a
  # random number generator - only if it shows within one frame of one second arrival.)
  for t, eng0 in second.items():
    for a in range(N_ACTIONS):
      eng = clone_engine(eng0) if a < N_ACTIONS - 1 else eng0
      key, reward, discount, over, board = step(eng, a)
      got = _Edge(index_of.get(key), reward, discount, over, board)
      if not got.same(edges[(t, a)]):
        _fail('the game keeps state outside its curtains, sprite positions, z-order, entity '
              'attributes and the Plot (a module global, a closure, a random number '
              'generator): the same state reached over two histories answered action {} '
              'differently'.format(a))

  return _finish(engine, probe, H, W, chars, with_frame, things0, backdrop0, z0, images, orders,
                 hiddens, boards, edges, plays[0], backdrops)


def _finish(engine, probe, H, W, chars, with_frame, things0, backdrop0, z0, images, orders, hiddens,
            boards, edges, n_plays, backdrops=None):
  """From the walked state graph to a `TracedGame`: `images[s]` (per-thing byte images, things in
  ascending character order), `orders[s]` (z-order string), `hiddens[s]` (`hidden_image()`),
  `boards[s]` (the rendered board's bytes) of every reached state, state 0 the one after
  `its_showtime()` (`things0`, `backdrop0`, `z0`), and `edges[(s, a)]` = `_Edge`.  Shared by the
  one-frame-per-play walker above and the many-states-per-call one (tabulate_batched.py)."""
  HW = H * W
  plays = [n_plays]
  # ---- who moves
  order = sorted(probe.things.keys())                     # the order of _image()'s parts
  varying = [i for i in range(len(order))
             if any(img[i] != things0[i] for img in images)]
  schedule = []
  for _, members in probe._update_groups:
    schedule.extend(ent.character for ent in members)
  movers = [ch for ch in schedule if order.index(ch) in varying]
  # A drape that covers SEVERAL cells and changes - coins picked up one by one, a door of two
  # cells (campx/things.py:161-262 sets no one-cell limit) - is tracked as one thing PER CELL it
  # ever covers: piece i stands on its cell while the curtain has it and is "absent" while it has
  # not, which is all the table kernels need to know about a thing (round 6; until then such a
  # game ran on the generic tier only).  The pieces share the drape's character, layer and place
  # in the z-order; `piece_cell[k]` is mover k's cell, None for an ordinary one-cell mover.
  piece_cell = []
  split = []
  for ch in movers:
    ent = probe.things[ch]
    covered = None
    if not isinstance(ent, _things.Sprite):
      masks = {img[order.index(ch)] for img in images}
      union = np.zeros(HW, bool)
      most = 0
      for part in masks:
        mask = np.frombuffer(part, np.uint8)
        if mask.size and mask.max() > 1:
          _fail('the curtain of {!r} holds values other than 0 and 1'.format(ch))
        union |= mask != 0
        most = max(most, int((mask != 0).sum()))
      if most > 1:
        covered = [int(c) for c in np.flatnonzero(union)]
    if covered is None:
      split.append(ch)
      piece_cell.append(None)
    else:
      split.extend([ch] * len(covered))
      piece_cell.extend(covered)
  # ... and so is a Backdrop that changes (a Backdrop.update() of its own, campx/things.py:103-148;
  # a sprite painted into it, campx/rendering.py:128,150): every (cell, character) it ever shows
  # other than what it showed after its_showtime() is a piece - on the board while the Backdrop
  # shows that character there, painted right on the scenery's backdrop, behind every thing.
  base_backdrop = np.frombuffer(backdrop0, np.int64)
  backdrop_pieces = set()
  for b in set(backdrops or ()):
    now = np.frombuffer(b, np.int64)
    for c in np.flatnonzero(now != base_backdrop):
      backdrop_pieces.add((int(c), int(now[c])))
  n_thing_movers = len(split)
  # PIECES or VARIANTS.  A piece tracked as a thing costs the render kernel a trace entry and a patch
  # slot per piece and row (seven coins and a walker on a 4x9 board: 3.5 TB/s).  (A game whose pieces
  # are few enough for the cell-indexed tables - gamespec.PIECES_AS_THINGS_MAX tracked things, no piece
  # in the Backdrop - would keep them as things: measured slower at every batch size, so the bound
  # is 0.)  Up to sixteen pieces beside at least one ordinary mover are
  # handed to the state-table tier as a MASK (CampxWideSpec.n_pieces: which of them show is one
  # 16-bit value per state, `pieces_as_mask` below - everything else about the game stays as the
  # pieces describe it).  More cells than that - day and night over a whole floor - and the SCENERY
  # itself comes in variants: a picture = the Backdrop's curtain and the curtains of the
  # several-cell drapes, the state names which picture shows, the render kernel lays that variant's
  # row (CampxWideSpec.n_variants; 5.2-6.4 TB/s) - as long as the pictures are at most
  # WIDE_MAX_VARIANTS.  Failing that, pieces as things again, up to the eight the kernels track.
  several = [ch for ch in dict.fromkeys(ch for ch, c in zip(split, piece_cell) if c is not None)]
  singles = [ch for ch, c in zip(split, piece_cell) if c is None]
  variants, variant_masks, state_variant = [backdrop0], [{}], [0] * len(images)
  few = not backdrop_pieces and len(split) <= gamespec.PIECES_AS_THINGS_MAX
  n_pieces = len(split) - len(singles) + len(backdrop_pieces)
  room = bool(singles) and len(singles) + 1 <= gamespec.WIDE_MAX_DYN
  pieces_as_mask = bool(n_pieces) and not few and room and n_pieces <= gamespec.WIDE_MAX_PIECES
  if (several or backdrop_pieces) and not few and room and not pieces_as_mask:
    index, pictures = {}, []
    for s_, img in enumerate(images):
      picture = (backdrops[s_] if backdrops else backdrop0,) + tuple(img[order.index(ch)] for ch in several)
      if picture not in index:
        index[picture] = len(pictures)
        pictures.append(picture)
      state_variant[s_] = index[picture]
    if len(pictures) <= gamespec.WIDE_MAX_VARIANTS:
      # (the first picture is state 0's: `backdrop` and the curtains after its_showtime())
      for picture in pictures:
        for code in np.unique(np.frombuffer(picture[0], np.int64)):
          if not 0 <= int(code) < 256 or chr(int(code)) not in chars:
            _fail('the Backdrop shows character code {}, which is not in its palette'.format(int(code)))
      variants = [picture[0] for picture in pictures]
      variant_masks = [{ch: np.frombuffer(part, np.uint8).reshape(H, W).copy()
                        for ch, part in zip(several, picture[1:])} for picture in pictures]
      backdrop_pieces = set()              # the variants carry every cell
      split, piece_cell = singles, [None] * len(singles)
      n_thing_movers = len(split)
    else:
      state_variant = [0] * len(images)
  folded = set(several) if len(variants) > 1 else set()
  for c, code in sorted(backdrop_pieces):
    if not 0 <= code < 256 or chr(code) not in chars:
      _fail('the Backdrop shows character code {} at cell {}, which is not in its palette'.format(code, c))
    split.append(chr(code))
    piece_cell.append(c)
  if len(split) > gamespec.WIDE_MAX_DYN and not pieces_as_mask:
    named = sorted({ch for ch, c in zip(split[:n_thing_movers], piece_cell) if c is not None})
    if backdrop_pieces:
      named.append('the Backdrop')
    _fail('moving drape(s) {} cover several cells that come and go - {} tracked cells with the other '
          'moving things (the table kernels track at most {} things, or {} such cells as a mask '
          'beside at least one ordinary mover), and more than {} different pictures of '
          'the scenery'.format(', '.join(ch if ch == 'the Backdrop' else repr(ch) for ch in named),
                               len(split), gamespec.WIDE_MAX_DYN, gamespec.WIDE_MAX_PIECES,
                               gamespec.WIDE_MAX_VARIANTS))
  movers = split
  in_backdrop = [k >= n_thing_movers for k in range(len(movers))]
  if not movers:
    # nothing ever moves (an agent walled in by what blocks it): the kernels still track one thing
    # - any that stands on exactly one cell will do; one state, five edges back to it
    for ch in schedule:
      part, ent = things0[order.index(ch)], probe.things[ch]
      if isinstance(ent, _things.Sprite):
        single = bool(part[2])
      else:
        mask = np.frombuffer(part, np.uint8)
        single = mask.max() <= 1 and int(mask.sum()) == 1
      if single:
        movers, piece_cell, in_backdrop, n_thing_movers = [ch], [None], [False], 1
        break
  # What tells two reached states with the same curtains apart - the z-order in force
  # (Plot.change_z_order) and the hidden values that are not themselves functions of the
  # curtains (`the_plot['prev_pos_A'] = layers['A']`, examples/boat_race.py:59, is one: the
  # live layer, the same whenever the board is) - is the "mode": one more tracked "cell".
  hidden_maps = [dict(hid) for hid in hiddens]
  paths = sorted({path for hid in hidden_maps for path in hid})
  free_paths = []
  for path in paths:
    seen = {}
    for img, z, hid in zip(images, orders, hidden_maps):
      if seen.setdefault((img, z), hid.get(path, _MISSING)) != hid.get(path, _MISSING):
        free_paths.append(path)
        break
  mode_keys = [(z,) + tuple(hid.get(path, _MISSING) for path in free_paths)
               for z, hid in zip(orders, hidden_maps)]
  modes, mode_index = [], {}
  for key in mode_keys:
    if key not in mode_index:
      mode_index[key] = len(modes)
      modes.append(key)
  state_mode = [mode_index[key] for key in mode_keys]
  n_tracked = len(movers) + (1 if len(modes) > 1 else 0)
  if not 1 <= len(movers) <= gamespec.WIDE_MAX_DYN and not pieces_as_mask:
    _fail('needs between 1 and {} moving things, found {} ({})'.format(
        gamespec.WIDE_MAX_DYN, len(movers), ''.join(movers) or 'nothing moves'))
  K = len(movers)
  # Two table forms come out of a tabulation.  The STATE table - one row per reachable
  # state, (state, action) -> state - always exists.  The DENSE table the one-cell tier's
  # kernels index by the things' cells, (cell, ..., cell, action), only when it fits them:
  dense_reason = None
  if HW > gamespec.MAX_CELLS:
    dense_reason = 'the board has more than {} cells'.format(gamespec.MAX_CELLS)
  elif n_tracked > gamespec.MAX_DYN:
    dense_reason = ('{} moving things{} are more than {} tracked values'.format(
        K, ' plus the z-order in force' if len(modes) > 1 else '', gamespec.MAX_DYN))
  elif len(modes) > HW:
    dense_reason = ('{} different modes (z-orders x hidden values: {}) are reached, more than '
                    'rows*cols'.format(len(modes), ', '.join(free_paths) or 'none'))
  elif HW ** n_tracked * N_ACTIONS > DENSE_MAX_ENTRIES:
    dense_reason = 'a table over {} cells ^ {} things has more than {} entries'.format(
        HW, n_tracked, DENSE_MAX_ENTRIES)
  if (any(in_backdrop) or len(variants) > 1 or pieces_as_mask) and dense_reason is None:
    dense_reason = ('the scenery changes (a Backdrop that repaints itself, drapes of several cells that come '
                    'and go): pieces the state names a mask of, or pictures it names the variant of')

  def where_is(s, k):
    """('at', cell) or ('absent', key): the one cell moving thing k occupies in state s, or - an
    empty curtain, an invisible sprite (its position still counts as state) - nowhere."""
    img = images[s]
    ch = movers[k]
    if in_backdrop[k]:                       # one (cell, character) of a Backdrop that changes
      shows = int(np.frombuffer(backdrops[s], np.int64)[piece_cell[k]]) == ord(ch)
      return ('at', piece_cell[k]) if shows else ('absent', b'')
    part = img[order.index(ch)]
    ent = probe.things[ch]
    if piece_cell[k] is not None:            # one cell of a drape that covers several
      return ('at', piece_cell[k]) if part[piece_cell[k]] else ('absent', b'')
    if isinstance(ent, _things.Sprite):
      if not part[2]:
        return ('absent', part)
      return ('at', part[0] * W + part[1])
    mask = np.frombuffer(part, np.uint8)
    if mask.max() > 1:
      _fail('the curtain of {!r} holds values other than 0 and 1'.format(ch))
    cells = np.flatnonzero(mask)
    if len(cells) == 0:
      return ('absent', b'')
    if len(cells) != 1:
      _fail('moving drape {!r} covers {} cells in a reachable state; the table kernels '
            'track a moving thing by the one cell it occupies'.format(ch, len(cells)))
    return ('at', int(cells[0]))

  # absent states take cell indices the thing never stands on
  places = [[where_is(s, k) for s in range(len(images))] for k in range(len(movers))]
  absent_alias, absent_cells = [], []
  for ch, seen in zip(movers, places):
    used = {c for kind, c in seen if kind == 'at'}
    keys = []
    for kind, key in seen:
      if kind == 'absent' and key not in keys:
        keys.append(key)
    free = [c for c in range(HW) if c not in used]
    if len(keys) > len(free):
      _fail('{!r} has {} different states in which it is not on the board; there is room '
            'for {}'.format(ch, len(keys), len(free)))
    absent_alias.append({key: free[j] for j, key in enumerate(keys)})
    absent_cells.append(set(free[:len(keys)]))

  def cell_of(s, k):
    kind, what = places[k][s]
    return what if kind == 'at' else absent_alias[k][what]

  game = TracedGame()
  game.rows, game.cols, game.chars = H, W, chars
  game.z_order = list(z0)
  game.mode_orders = [list(key[0]) for key in modes]
  game.hidden_paths = free_paths
  game.frame_in_state = bool(with_frame)
  game.backdrop = np.frombuffer(backdrop0, np.int64).astype(np.uint8).reshape(H, W)
  game.movers = movers
  game.piece_cell = piece_cell
  game.in_backdrop = in_backdrop
  game.variants = [np.frombuffer(b, np.int64).astype(np.uint8).reshape(H, W) for b in variants]
  game.variant_masks = variant_masks
  game.pieces_as_mask = pieces_as_mask
  game.absent_cells = absent_cells
  game.statics = []
  for ch in schedule:
    if ch in movers[:n_thing_movers] or ch in folded:
      continue
    ent = probe.things[ch]
    if isinstance(ent, _things.Sprite):
      mask = np.zeros((H, W), np.uint8)
      if ent.visible:
        mask[ent.position.row, ent.position.col] = 1
    else:
      mask = np.frombuffer(things0[order.index(ch)], np.uint8).reshape(H, W).copy()
      if mask.max() > 1:
        _fail('the curtain of {!r} holds values other than 0 and 1'.format(ch))
    game.statics.append((ch, mask))
  if len(game.statics) > gamespec.MAX_STATIC:
    _fail('more than {} static things'.format(gamespec.MAX_STATIC))

  state_cells = [tuple(cell_of(s, k) for k in range(K)) +
                 ((state_mode[s],) if len(modes) > 1 else ())
                 for s in range(len(orders))]
  if len(set(zip(state_cells, state_variant))) != len(state_cells):
    _fail('two reachable states have every moving thing on the same cells and the same mode '
          '(a thing that is "absent" in more ways than its free cells can name)')
  game.init_cells = state_cells[0]
  board0 = np.frombuffer(boards[0], np.uint8)
  game.init_visible = [int(places[k][0][0] == 'at' and board0[state_cells[0][k]] == ord(ch))
                       for k, ch in enumerate(movers)]
  if places[0][0][0] != 'at' and dense_reason is None:
    dense_reason = ('the first moving thing ({!r}) is not on the board after its_showtime()'
                    .format(movers[0]))

  # ---- the render kernels lay ONE scenery row under the movers: no order may change it
  for m in range(1, len(modes)):
    probe_cells = tuple(state_cells[0][:K]) + (m,)
    if (game.model_board(probe_cells, movers=False).tobytes() !=
        game.model_board(state_cells[0], movers=False).tobytes()):
      _fail('a z-order change re-orders overlapping static things (the scenery itself changes); '
            'only moving things may change places')

  # ---- every reached board is "backdrop + things in z-order" of the cells alone
  for cells, board, variant in zip(state_cells, boards, state_variant):
    model = game.model_board(cells, variant=variant)
    if model.tobytes() != board:
      _fail('a rendered board is not "backdrop, then every thing in z-order" of the moving '
            'things\' cells')

  # ---- hidden performance (examples/boat_race.py:117-151): classes of the watched mover
  perf_of = None
  if engine.hidden_penalty is not None:
    who, masks, unit = engine.hidden_penalty
    for ch in who:
      if ch not in movers:
        _fail('hidden penalty watches {!r}, which never moves'.format(ch))
      if movers.count(ch) > 1:
        _fail('hidden penalty watches {!r}, a drape of several cells'.format(ch))
      if absent_cells[movers.index(ch)]:
        _fail('hidden penalty watches {!r}, which leaves the board'.format(ch))
    cls = np.zeros(HW, np.int32)
    for k, m in enumerate(masks):
      cls[np.flatnonzero(m.detach().cpu().numpy().reshape(-1))] = k + 1
    watched = [movers.index(ch) for ch in who]

    def perf_of(src, dst):
      return int(unit) * sum(int(cls[dst[k]]) for k in watched)
  elif engine.hidden_performance is not None:
    agent, masks = engine.hidden_performance
    if agent not in movers:
      _fail('hidden performance watches {!r}, which never moves'.format(agent))
    if movers.count(agent) > 1:
      _fail('hidden performance watches {!r}, a drape of several cells'.format(agent))
    if absent_cells[movers.index(agent)]:
      _fail('hidden performance watches {!r}, which leaves the board'.format(agent))
    cls = np.zeros(HW, np.int32)
    for k, m in enumerate(masks):
      cls[np.flatnonzero(m.detach().cpu().numpy().reshape(-1))] = k + 1
    n_cls, who = len(masks), movers.index(agent)

    def perf_of(src, dst):
      a, b = int(cls[src[who]]), int(cls[dst[who]])
      if a == 0 or b == 0:
        return 0
      fwd = 1 if a == n_cls else a + 1
      back = n_cls if a == 1 else a - 1
      return int(b == fwd) - int(b == back)

  # ---- the state table: one row per reachable state
  S = len(images)
  codes = [ord(ch) for ch in movers]
  game.dense_reason = dense_reason
  game.st_cells = np.array([cells[:K] for cells in state_cells], np.uint16).reshape(S, K)
  game.st_present = np.array([[places[k][s][0] == 'at' for k in range(K)] for s in range(S)],
                             bool).reshape(S, K)
  game.st_board = np.stack([np.frombuffer(b, np.uint8) for b in boards])
  game.st_shows = np.zeros((S, K), np.uint8)
  for k in range(K):
    game.st_shows[:, k] = (game.st_present[:, k] &
                           (game.st_board[np.arange(S), game.st_cells[:, k]] == codes[k]))
  game.st_mode = np.array(state_mode, np.int32)
  game.st_variant = np.array(state_variant, np.uint16)
  game.st_next = np.tile(np.arange(S, dtype=np.int32)[:, None], (1, N_ACTIONS))
  game.st_reward = np.full((S, N_ACTIONS), np.nan, np.float32)
  game.st_done = np.zeros((S, N_ACTIONS), np.uint8)
  game.st_discount = np.ones((S, N_ACTIONS), np.float32)
  game.st_dcode = np.zeros((S, N_ACTIONS), np.uint8)
  game.st_perf = np.zeros((S, N_ACTIONS), np.int8)
  game.st_reached = np.zeros((S, N_ACTIONS), bool)
  game.discount_list = [1.0]          # code -> value; code 0 stands for the default
  for (s, a), e in edges.items():
    game.st_next[s, a] = e.next
    game.st_reward[s, a] = e.reward
    game.st_done[s, a] = int(e.over)
    game.st_discount[s, a] = e.discount
    if e.discount != (0.0 if e.over else 1.0):
      if e.discount not in game.discount_list[1:]:
        if len(game.discount_list) == 16:
          _fail('more than 15 distinct discounts besides the default')
        game.discount_list.append(e.discount)
      game.st_dcode[s, a] = 1 + game.discount_list[1:].index(e.discount)
    game.st_perf[s, a] = perf_of(state_cells[s], state_cells[e.next]) if perf_of else 0
    game.st_reached[s, a] = True
  game.any_reward = bool((~np.isnan(game.st_reward[game.st_reached])).any())
  game.has_perf = perf_of is not None
  game.perf_spec = engine.hidden_performance
  game.penalty_spec = engine.hidden_penalty
  game.n_states, game.n_plays = S, plays[0]

  # ---- the dense table (one-cell tier), when the game fits it
  game.n = None
  if dense_reason is None:
    Kt = n_tracked
    n = HW ** Kt * N_ACTIONS
    game.n = n
    game.next_cells = np.zeros((Kt, n), np.uint16)
    game.visible = np.zeros((Kt, n), np.uint8)
    game.reward = np.full((n,), np.nan, np.float32)
    game.done = np.zeros((n,), np.uint8)
    game.discount = np.ones((n,), np.float32)
    game.dcode = np.zeros((n,), np.uint8)
    game.perf = np.zeros((n,), np.int8)
    game.reached = np.zeros((n,), bool)
    # entries nobody can reach: stay where you are, pay nothing
    idx = np.arange(n) // N_ACTIONS
    for k in range(Kt - 1, -1, -1):
      game.next_cells[k] = idx % HW
      idx = idx // HW
    for (s, a), e in edges.items():
      i = game.index_of(state_cells[s], a)
      dst = state_cells[e.next]
      for k in range(K):
        game.next_cells[k, i] = dst[k]
        game.visible[k, i] = game.st_shows[e.next, k]
      if Kt > K:
        game.next_cells[K, i] = dst[K]            # the mode: never visible
      game.reward[i] = game.st_reward[s, a]
      game.done[i] = game.st_done[s, a]
      game.discount[i] = game.st_discount[s, a]
      game.dcode[i] = game.st_dcode[s, a]
      game.perf[i] = game.st_perf[s, a]
      game.reached[i] = True
  return game


def to_spec(game):
  """`TracedGame` -> `CampxSpec` with `table_only` set: scenery and paint parameters as
  `gamespec.lower()` derives them, no rules, and - for one mover - the transition table
  filled in (games with two to four movers hand `trace_bytes()` etc. to
  campx_pair_table_pack)."""
  H, W = game.rows, game.cols
  HW = H * W
  if game.dense_reason is not None:
    _fail('the one-cell tier cannot take this game: ' + game.dense_reason)
  layer_of = {ch: i for i, ch in enumerate(game.chars)}
  z_of = {ch: i + 1 for i, ch in enumerate(game.z_order)}
  spec = gamespec.CampxSpec()
  spec.magic, spec.version = gamespec.SPEC_MAGIC, gamespec.SPEC_VERSION
  spec.rows, spec.cols = H, W
  spec.n_layers = len(game.chars)
  spec.n_dyn, spec.n_static = game.n_tracked, len(game.statics)
  spec.n_rules = 0
  spec.table_only = 1
  spec.any_reward = int(game.any_reward)
  for i, ch in enumerate(game.chars):
    spec.layer_char[i] = ord(ch)
  for d, ch in enumerate(game.movers):
    spec.dyn_layer[d] = layer_of[ch]
    spec.dyn_z[d] = z_of[ch]
    spec.dyn_row0[d], spec.dyn_col0[d] = divmod(int(game.init_cells[d]), W)
    if game.is_absent(d, game.init_cells[d]):
      spec.dyn_z[d] = 0          # not on the board at the start: its_showtime() paints nothing
  if game.n_tracked > len(game.movers):
    # the z-order mode: a "thing" of z rank 0 - behind the backdrop, never painted - whose
    # cell is the index of the order in force (0 after its_showtime())
    d = len(game.movers)
    spec.dyn_layer[d], spec.dyn_z[d] = spec.dyn_layer[0], 0
    spec.dyn_row0[d], spec.dyn_col0[d] = 0, 0
  top_layer = np.array([[layer_of[chr(c)] for c in row] for row in game.backdrop], np.uint8)
  top_z = np.zeros((H, W), np.uint8)
  cover = np.zeros((H, W), np.uint16)
  static_of = {ch: i for i, (ch, _) in enumerate(game.statics)}
  masks = dict(game.statics)
  for ch in game.z_order:
    if ch in static_of:
      m = masks[ch] != 0
      top_layer[m] = layer_of[ch]
      top_z[m] = z_of[ch]
      cover[m] |= np.uint16(1 << static_of[ch])
  for i in range(HW):
    spec.static_top_layer[i] = int(top_layer.flat[i])
    spec.static_top_z[i] = int(top_z.flat[i])
    spec.static_cover[i] = int(cover.flat[i])
    spec.obs_template[int(top_layer.flat[i]) * HW + i] = 1
  spec.perf_dyn = -1
  if game.has_perf and game.penalty_spec is not None:
    who, classes, unit = game.penalty_spec
    spec.perf_mode, spec.perf_scale, spec.perf_offset = 1, int(unit), 0
    spec.perf_mask = sum(1 << game.movers.index(ch) for ch in who)
    spec.perf_dyn = min(game.movers.index(ch) for ch in who)
    for k, m in enumerate(classes):
      for cell in np.flatnonzero(m.detach().cpu().numpy().reshape(-1)):
        spec.cell_class[int(cell)] = k + 1
  elif game.has_perf:
    agent, cycle = game.perf_spec
    spec.perf_dyn, spec.perf_n = game.movers.index(agent), len(cycle)
    spec.perf_mode, spec.perf_scale, spec.perf_offset = 0, 1, -1
    for k, m in enumerate(cycle):
      for cell in np.flatnonzero(m.detach().cpu().numpy().reshape(-1)):
        spec.cell_class[int(cell)] = k + 1
  for code, value in enumerate(game.discount_list):
    if code:
      spec.discount_list[code] = float(value)
  if game.n_tracked == 1:
    for i in range(game.n):
      tr = spec.table[i]
      nxt = int(game.next_cells[0, i])
      tr.reward = float(game.reward[i])
      tr.next_cell, tr.perf = nxt, int(game.perf[i])
      tr.done = int(game.done[i]) | (int(game.dcode[i]) << 4)
      in_front = spec.static_top_z[nxt] > spec.dyn_z[0]
      if game.reached[i]:        # (from the boards the user's own code rendered)
        in_front = not game.visible[0, i]
      tr.paint = int(spec.static_top_layer[nxt]) | (0x80 if in_front else 0)
    spec.table_valid = 1
  return spec


def to_wide_spec(game):
  """`TracedGame` -> (`CampxWideSpec`, arrays): the game as its STATE table
  (include/campx_hip.h).  `arrays` are the numpy arrays the spec's host pointers point at:
  keep them alive until campx_wide_tables_build() has run.  "Shows" bits come from the boards
  the user's own code rendered."""
  H, W = game.rows, game.cols
  HW = H * W
  S = game.n_states
  # pieces handed over as a mask are not among the kernels' things
  as_mask = bool(getattr(game, 'pieces_as_mask', False))
  things = [k for k in range(len(game.movers)) if not (as_mask and game.piece_cell[k] is not None)]
  pieces = [k for k in range(len(game.movers)) if k not in things and game.st_shows[:, k].any()]
  K = len(things)
  if HW < 16:
    _fail('the state-table tier needs a board of at least 16 cells')
  if S > gamespec.WIDE_MAX_STATES:
    _fail('{} reachable states; the state table takes {}'.format(S, gamespec.WIDE_MAX_STATES))
  layer_of = {ch: i for i, ch in enumerate(game.chars)}
  spec = gamespec.CampxWideSpec()
  spec.magic, spec.version = gamespec.SPEC_MAGIC, gamespec.SPEC_VERSION
  spec.rows, spec.cols, spec.n_layers = H, W, len(game.chars)
  spec.n_dyn, spec.n_states = K, S
  spec.any_reward, spec.has_perf = int(game.any_reward), int(game.has_perf)
  spec.any_dcode = int((game.st_dcode != 0).any())
  for i, ch in enumerate(game.chars):
    spec.layer_char[i] = ord(ch)
  for d, k in enumerate(things):
    spec.dyn_layer[d] = layer_of[game.movers[k]]
  for code, value in enumerate(game.discount_list):
    if code:
      spec.discount_list[code] = float(value)
  top = game.model_board(game.init_cells, movers=False).reshape(-1)
  for i in range(HW):
    spec.static_top_layer[i] = layer_of[chr(int(top[i]))]
  cells = np.where(game.st_present, game.st_cells, 0).astype(np.uint16)[:, things]
  hidden = (game.st_shows[:, things] == 0).astype(np.uint16)
  arrays = dict(        # (copies: a TracedGame may be shared through the tabulation cache)
      state_cells=np.array(cells | (hidden << 15), np.uint16, order='C'),
      next_state=np.array(game.st_next, np.int32, order='C'),
      reward=np.array(game.st_reward, np.float32, order='C'),
      done=np.array(game.st_done | (game.st_dcode << 4), np.uint8, order='C'),
      perf=np.array(game.st_perf, np.int8, order='C'))
  variants = getattr(game, 'variants', None) or [game.backdrop]
  if len(variants) > 1:
    # the scenery's front-most layer per cell, per picture of the Backdrop; which one a state shows
    spec.n_variants = len(variants)
    tops = np.zeros((len(variants), HW), np.uint8)
    for v in range(len(variants)):
      scenery = game.model_board(game.init_cells, movers=False, variant=v).reshape(-1)
      tops[v] = [layer_of[chr(int(c))] for c in scenery]
    arrays['variant_top_layer'] = tops
    arrays['state_variant'] = np.array(game.st_variant, np.uint16, order='C')
  if pieces:
    # which pieces SHOW in a state (on their cell, nothing in front): one bit each
    spec.n_pieces = len(pieces)
    shown = np.zeros(S, np.uint16)
    for p, k in enumerate(pieces):
      cell = int(game.piece_cell[k])
      spec.piece_cell[p], spec.piece_layer[p] = cell, layer_of[game.movers[k]]
      if spec.piece_layer[p] == spec.static_top_layer[cell]:
        _fail('piece {!r} at cell {} is the character the scenery shows there anyway'.format(game.movers[k], cell))
      shown |= (game.st_shows[:, k] != 0).astype(np.uint16) << p
    arrays['state_pieces'] = np.array(shown, np.uint16, order='C')
  for name, a in arrays.items():
    setattr(spec, name, a.ctypes.data)
  if not game.has_perf:
    spec.perf = None
  return spec, arrays
