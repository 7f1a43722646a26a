"""Tabulate a game of arbitrary Python `update()` classes MANY STATES PER CALL.

`tabulate.trace()` walks a game's state graph with one generic-tier frame of Python per (state,
action): 60 000 frames is what it will spend (`MAX_PLAYS`), so a user's own two-box sokoban -
10^5 states and more - was refused, and only the re-typed `campx_amd.rules` classes reached the
device-side enumerator ("what's missing" 2).  The reference's own classes are written
in arithmetic-only tensor operations (examples/boat_race.py:40-57, README.md:3) precisely so that
they run on other tensor types.  This module runs them on one: `lanes.Lanes`, N tensors at once.

The walk is the same breadth-first one, a LEVEL at a time: the curtains of every state of the
frontier are stacked into `[N, H, W]` lanes, the frame is rendered for all of them (the engine's
own `_render()` on a renderer that paints lanes), and the game's unmodified `update()` methods run
ONCE per action for the whole frontier - `Engine._update_and_render()` / `_apply_and_clear_plot()`
themselves (campx/engine.py:168-293), on a deep copy of the user's engine whose drape curtains
and rendered layers are `Lanes`.  Next states are keyed exactly (the cell of every one-cell
drape; anything else is not this tier's game), deduplicated with sort / searchsorted, and numbered
in the order the one-frame-per-play walker discovers them, so both walkers hand `tabulate._finish`
the same graph and produce the same `TracedGame`.

What this tier takes - checked, `CannotBatch` otherwise, and `tabulate.trace()` then falls back
to the one-frame-per-play walk, which takes anything and says what is wrong with the rest:
  * things are Drapes with 0/1 curtains, or Sprites (a Sprite's position is a Python tuple, one
    value for all lanes: a frame whose states have their sprites in different places is split by
    where they stand, like a frame that branches; an invisible Sprite must stand where it stood
    at the start); the Backdrop's `update()` is the base class's no-op;
  * nothing a frame can read changes besides the curtains and what renders from them: entity
    attributes, Plot entries (other than aliases of the renderer's live layers,
    `the_plot['prev_pos_A'] = layers['A']`, boat_race.py:59), the z-order, the frame number;
  * Python-level reads of a value that differs between states - `if pushed and not blocked:`,
    `int(x)`, `.item()` - SPLIT the frame (round 5, late): it is run again for each group of
    states that read the same, recursively, at most `MAX_SPLITS` groups per (level, action); so
    rewards, termination and discounts may differ from state to state.  `.numpy()` / `.tolist()`
    of a tensor that differs between states split the same way, by the tensor's whole contents
    (the value handed out is a read-only copy).  What has no lane-by-lane form at all -
    `nonzero()` ON a lane tensor, a tensor index into a tensor - still hands the game to the
    other walker.
A sample of the tabulated edges (every action from the first state, and `CHECK_EDGES` random
ones) is then replayed on the ordinary generic tier - the user's code on plain tensors - and must
agree bit for bit; a disagreement falls back as well.

Host logic (torch on the CPU, or on the engine's device when there is one); no HIP kernel.
"""

import copy

import numpy as np
import torch

from . import gamespec
from . import lanes
from . import rendering
from . import tabulate
from . import things as _things
from .lanes import CannotBatch, Diverged, Lanes

N_ACTIONS = gamespec.N_ACTIONS
CHECK_EDGES = 48
MAX_STATES = 1 << 24          # the wide tier's state index


class _LanesRenderer(object):
  """`rendering.BaseObservationRenderer`'s interface (campx/rendering.py:104-219) over lanes: the
  board and every `layers[ch]` are persistent `Lanes` objects re-bound per render, as the
  reference's are persistent tensors re-bound with `set_` (rendering.py:209) - so what a game
  stored (`the_plot['prev_pos_A'] = layers['A']`) stays live."""

  def __init__(self, rows, cols, chars, device):
    self.rows, self.cols, self.chars = rows, cols, sorted(chars)
    self.device = device
    self.n = 1
    self._board = lanes.wrap(torch.zeros((1, rows, cols), dtype=torch.int64, device=device))
    self._layers = {ch: lanes.wrap(torch.zeros((1, rows, cols), dtype=torch.uint8, device=device))
                    for ch in self.chars}
    self._canvas = None

  def clear(self):
    self._canvas = None

  def paint_all_of(self, curtain):
    if isinstance(curtain, Lanes):
      raise CannotBatch('the Backdrop\'s curtain became lane-varying')
    self._canvas = curtain.to(self.device).unsqueeze(0).expand(self.n, self.rows, self.cols).clone()

  def paint_sprite(self, character, position):
    self._canvas[:, position.row, position.col] = ord(character)

  def paint_drape(self, character, curtain):
    if isinstance(curtain, Lanes):
      mask = lanes.plain(curtain)
      if mask.shape[0] != self.n:
        raise CannotBatch('a curtain of {} lanes in a frame of {}'.format(mask.shape[0], self.n))
    else:
      mask = curtain.to(self.device).unsqueeze(0)
    # (board - m * board + m * ord(ch), rendering.py:176-178, for the 0/1 masks this tier checks)
    self._canvas = torch.where(mask != 0, torch.full_like(self._canvas, ord(character)), self._canvas)

  def render(self):
    with lanes._guard():
      torch.Tensor.set_(self._board, self._canvas)
      for ch in self.chars:
        torch.Tensor.set_(self._layers[ch], (self._canvas == ord(ch)).to(torch.uint8))
    return rendering.Observation(board=self._board, layers=self._layers, layered_board=None)


def _walk_values(obj, seen, path, out, depth=0):
  """(path, container, key) of every tensor reachable through lists / tuples / dicts."""
  if depth > 4 or id(obj) in seen:
    return
  if isinstance(obj, dict):
    seen.add(id(obj))
    for k, v in obj.items():
      if torch.is_tensor(v):
        out.append(('{}[{!r}]'.format(path, k), obj, k))
      else:
        _walk_values(v, seen, '{}[{!r}]'.format(path, k), out, depth + 1)
  elif isinstance(obj, list):
    seen.add(id(obj))
    for i, v in enumerate(obj):
      if torch.is_tensor(v):
        out.append(('{}[{}]'.format(path, i), obj, i))
      else:
        _walk_values(v, seen, '{}[{}]'.format(path, i), out, depth + 1)


def _tensor_slots(eng):
  """Every place outside the curtains where the engine's entities and Plot hold a tensor."""
  out, seen = [], set()
  for ch, ent in eng.things.items():
    for name, value in vars(ent).items():
      if name == '_curtain':
        continue
      if torch.is_tensor(value):
        out.append(('things[{!r}].{}'.format(ch, name), vars(ent), name))
      else:
        _walk_values(value, seen, 'things[{!r}].{}'.format(ch, name), out)
  for name, value in vars(eng.backdrop).items():
    if name != '_curtain' and torch.is_tensor(value):
      out.append(('backdrop.' + name, vars(eng.backdrop), name))
  plot = eng._the_plot
  for key in list(dict.keys(plot)):
    value = dict.__getitem__(plot, key)
    if torch.is_tensor(value):
      out.append(('the_plot[{!r}]'.format(key), plot, key))
    else:
      _walk_values(value, seen, 'the_plot[{!r}]'.format(key), out)
  return out


def _get(container, key):
  return dict.__getitem__(container, key) if isinstance(container, dict) else container[key]


def _put(container, key, value):
  if isinstance(container, dict):
    dict.__setitem__(container, key, value)
  else:
    container[key] = value


def _plain_image(eng):
  """`tabulate.hidden_image()` of everything that is not a `Lanes`: what must stay put."""
  stash = []
  for path, container, key in _tensor_slots(eng):
    value = _get(container, key)
    if isinstance(value, Lanes):
      stash.append((container, key, value))
      _put(container, key, ('lanes', path))
  try:
    return tabulate.hidden_image(eng, False)
  finally:
    for container, key, value in stash:
      _put(container, key, value)


class _Frontier(object):
  """The lifted engine: one deep copy of the user's, curtains and rendered layers as lanes."""

  def __init__(self, probe, actions, device):
    self.eng = eng = tabulate.clone_engine(probe)
    self.actions = actions
    self.device = device
    self.H, self.W = eng.rows, eng.cols
    if type(eng.backdrop).update is not _things.Backdrop.update:
      raise CannotBatch('the Backdrop has an update() of its own')
    self.drapes = sorted(eng.things.keys())          # every thing, ascending: _image()'s order
    # A Sprite's state is a Python tuple and a flag, one value for every lane: among the states'
    # arrays it is a one-cell "curtain" (empty while it does not show), and a frame whose states
    # disagree about a sprite is split like one whose states branch differently (frame()).  An
    # invisible sprite keeps no cell, so it has to stand where it stood at the start.
    self.sprites = {ch: (ent.position.row, ent.position.col) for ch, ent in eng.things.items()
                    if isinstance(ent, _things.Sprite)}
    old = eng._renderer
    chars = set(eng.things.keys()) | set(eng.backdrop.palette)
    self.renderer = _LanesRenderer(self.H, self.W, chars, device)
    # whatever the game kept of the old renderer's live tensors now means the new one's
    remap = {id(old._board): self.renderer._board}
    for ch, t in old._layers.items():
      remap[id(t)] = self.renderer._layers[ch]
    eng._renderer = self.renderer
    for path, container, key in _tensor_slots(eng):
      value = _get(container, key)
      if id(value) in remap:
        _put(container, key, remap[id(value)])
    for ch in self.drapes:
      if ch in self.sprites:
        continue
      ent = eng.things[ch]
      ent._curtain = lanes.wrap(ent._curtain.to(device).to(torch.uint8).unsqueeze(0).clone())
    self.slots = [(path, c, k) for path, c, k in _tensor_slots(eng)]
    self.owned = {id(self.renderer._board)} | {id(t) for t in self.renderer._layers.values()}
    self.image0 = _plain_image(eng)
    self.reads0 = tabulate.FRAME_READS[0]
    self.z0 = ''.join(eng.things.keys())

  def sprite_mask(self, ch, n):
    """The one-cell curtain of sprite `ch` as it stands in the engine now, for n states."""
    ent = self.eng.things[ch]
    mask = torch.zeros((n, self.H, self.W), dtype=torch.uint8, device=self.device)
    if ent.visible:
      mask[:, ent.position.row, ent.position.col] = 1
    elif (ent.position.row, ent.position.col) != self.sprites[ch]:
      raise CannotBatch('{!r} is a Sprite that moves while it does not show'.format(ch))
    return mask

  def place_sprites(self, curtains):
    """Every sprite where its mask says - one place for all the states of this frame, else the
    frame is split (`Diverged`: states grouped by where the sprites stand)."""
    if not self.sprites:
      return
    n = int(next(iter(curtains.values())).shape[0])
    if n > 1:
      flat = torch.cat([curtains[ch].reshape(n, -1) for ch in sorted(self.sprites)], dim=1)
      if not bool((flat == flat[:1]).all()):
        _, inverse = torch.unique(flat, dim=0, return_inverse=True)
        raise Diverged('the states of this frame have their sprites in different places', inverse)
    for ch in self.sprites:
      ent = self.eng.things[ch]
      cell = torch.nonzero(curtains[ch][0].reshape(-1)).reshape(-1)
      if cell.numel() > 1:
        raise CannotBatch('sprite {!r} on more than one cell'.format(ch))
      if cell.numel():
        ent._position = ent.Position(int(cell[0]) // self.W, int(cell[0]) % self.W)
        ent._visible = True
      else:
        ent._position = ent.Position(*self.sprites[ch])
        ent._visible = False

  def frame(self, curtains, a):
    """One frame of action `a` for N states (`curtains[ch]`: uint8 `[N, H, W]`).  Returns
    (next curtains, reward f32 [N] with NaN for None, discount, over, boards uint8 [N, H*W])."""
    eng = self.eng
    n = int(next(iter(curtains.values())).shape[0])
    self.renderer.n = n
    self.place_sprites(curtains)
    with lanes._guard():
      for ch in self.drapes:
        if ch in self.sprites:
          continue
        ent = eng.things[ch]
        if not isinstance(ent._curtain, Lanes):
          raise CannotBatch('{!r} replaced its curtain by an ordinary tensor'.format(ch))
        torch.Tensor.set_(ent._curtain, curtains[ch].clone())
    eng._render()                       # the states' own rendering: what this frame's update() reads
    eng._game_over = False
    eng._the_plot._clear_engine_directives()
    frame_number = eng._the_plot._frame
    try:
      eng._update_and_render(copy.deepcopy(self.actions[a]))
      reward, discount, rerender = eng._apply_and_clear_plot()
    finally:                             # (one frame, over and over - also when it is given up half way)
      eng._the_plot._frame = frame_number
      eng._the_plot.update_group = None
    if tabulate.FRAME_READS[0] != self.reads0:
      raise CannotBatch('the game reads the_plot.frame')
    if rerender or ''.join(eng.things.keys()) != self.z0:
      raise CannotBatch('the game changes the z-order')
    over = bool(eng._game_over)
    if isinstance(discount, Lanes):
      raise CannotBatch('a discount that differs between states')
    nxt = {}
    for ch in self.drapes:
      if ch in self.sprites:
        nxt[ch] = self.sprite_mask(ch, n)
        continue
      cur = eng.things[ch]._curtain
      if not isinstance(cur, Lanes):
        raise CannotBatch('{!r} replaced its curtain by an ordinary tensor'.format(ch))
      p = lanes.plain(cur)
      if tuple(p.shape) != (n, self.H, self.W):
        raise CannotBatch('the curtain of {!r} changed its shape'.format(ch))
      nxt[ch] = p.to(torch.uint8).clone()
    if reward is None:
      r = torch.full((n,), float('nan'), dtype=torch.float32, device=self.device)
    elif isinstance(reward, Lanes):
      p = lanes.plain(reward)
      if p.dim() != 1 and p[0].numel() != 1:
        raise CannotBatch('a reward that is not a number')
      r = p.reshape(n).to(torch.float32)
    else:
      r = torch.full((n,), float(tabulate.reward_f32(reward)), dtype=torch.float32, device=self.device)
    # nothing else may have moved
    for path, container, key in _tensor_slots(eng):
      value = _get(container, key)
      if isinstance(value, Lanes) and id(value) not in self.owned:
        raise CannotBatch('{} holds a lane-varying tensor of its own: state outside the curtains'.format(path))
    if _plain_image(eng) != self.image0:
      raise CannotBatch('something besides the curtains changed (an entity attribute, a Plot entry)')
    board = lanes.plain(self.renderer._board).to(torch.uint8).reshape(n, self.H * self.W)
    return nxt, r, float(np.float32(discount)), over, board


MAX_SPLITS = 2048        # groups of states one (level, action) frame may fall into (a class that
                         # reads its one-cell curtain as numpy: one group per cell it stands on)


def _frame_any(front, curtains, a, budget=None):
  """`front.frame()` for N states whose Python-level reads need not agree: when a branch (`if
  x:`, `int(x)`, `.item()`) reads different values in different states, the frame is run again
  for each group of states that read the same - recursively, a later branch may split a group
  again.  Returns (next curtains, reward f32 [N], discount f32 [N], over bool [N], boards)."""
  n = int(next(iter(curtains.values())).shape[0])
  budget = [MAX_SPLITS] if budget is None else budget
  try:
    nxt, r, discount, over, board = front.frame(curtains, a)
    dev = r.device
    return (nxt, r, torch.full((n,), discount, dtype=torch.float32, device=dev),
            torch.full((n,), bool(over), dtype=torch.bool, device=dev), board)
  except Diverged as split:
    # What the abandoned frame did BEFORE the branch that split it stays in the lifted engine:
    # curtains are re-bound by the next frame(), but an attribute written, a Plot entry set or a
    # plain tensor edited in place would be the pre-state of the sub-groups' re-runs - and a
    # write the rest of the frame undoes (a flag set, read, cleared) would pass the image check
    # of the completed frames.  So: the abandoned prefix must have changed nothing.
    if _plain_image(front.eng) != front.image0:
      raise CannotBatch('{} - after the frame had already changed something besides the curtains '
                        '(an entity attribute, a Plot entry): it cannot be run again for each '
                        'group of states from there'.format(split))
    values = split.values.reshape(-1)
    if values.numel() != n or (values.is_floating_point() and bool(torch.isnan(values).any())):
      raise CannotBatch(str(split))          # (not this frame's lanes, or a NaN: no grouping by value)
    groups = torch.unique(values)
    budget[0] -= int(groups.numel())
    if groups.numel() < 2 or budget[0] < 0:
      raise CannotBatch('{} - in more than {} different ways in one frame'.format(split, MAX_SPLITS))
    parts = []
    for g in groups:
      idx = torch.nonzero(values == g).reshape(-1)
      parts.append((idx, _frame_any(front, {ch: c[idx] for ch, c in curtains.items()}, a, budget)))
    dev = parts[0][1][1].device
    nxt = {ch: torch.empty_like(c) for ch, c in curtains.items()}
    r = torch.empty((n,), dtype=torch.float32, device=dev)
    discount = torch.empty((n,), dtype=torch.float32, device=dev)
    over = torch.empty((n,), dtype=torch.bool, device=dev)
    board = torch.empty((n, front.H * front.W), dtype=torch.uint8, device=dev)
    for idx, (pn, pr, pd, po, pb) in parts:
      for ch in nxt:
        nxt[ch][idx] = pn[ch]
      r[idx], discount[idx], over[idx], board[idx] = pr, pd, po, pb
    return nxt, r, discount, over, board


def trace(engine, actions=None, max_plays=None, device=None):
  """The `TracedGame` `tabulate.trace()` would return, from many-states-per-call frames; raises
  `CannotBatch` for games this tier does not take (module docstring).  A game that draws random
  numbers or reads the clock is refused (`tabulate.TabulationError`; campx_amd/chance.py)."""
  from . import chance
  with chance.forbidden(tabulate.TabulationError):
    return _trace_on_lanes(engine, actions, max_plays, device)


def _trace_on_lanes(engine, actions, max_plays, device):
  if engine.backdrop is None:
    raise ValueError('the Engine has no Backdrop yet')
  H, W = engine.rows, engine.cols
  HW = H * W
  chars = sorted(set(engine.things.keys()) | set(engine.backdrop.palette))
  if HW > gamespec.WIDE_MAX_CELLS or H > 127 or W > 127 or len(chars) > gamespec.MAX_LAYERS:
    raise CannotBatch('board or character set beyond every tier')       # (the walker says it properly)
  actions = tabulate.default_actions() if actions is None else list(actions)
  if device is None:
    device = torch.device('cpu')
  device = torch.device(device)

  # (many small tensor operations: on a 256-core host torch's default thread count makes each of
  # them slower, not faster - 66 s against 15 s for the 592 588-state warehouse)
  threads = torch.get_num_threads()
  torch.set_num_threads(min(threads, 16))
  try:
    return _trace(engine, actions, device, H, W, HW, chars)
  finally:
    torch.set_num_threads(threads)


def _trace(engine, actions, device, H, W, HW, chars):
  probe = tabulate.clone_engine(engine)
  probe._batch, probe._device, probe._fused = None, None, None
  probe._the_plot.__class__ = tabulate.probe_plot_class(type(probe._the_plot))
  obs, _, _ = probe.its_showtime()
  if probe.game_over:
    raise CannotBatch('the episode is over after its_showtime()')
  things0, backdrop0, z0 = tabulate._image(probe)
  hidden0 = tabulate.hidden_image(probe, False)
  front = _Frontier(probe, actions, device)
  drapes = front.drapes                                   # ascending characters = _image()'s order
  start = {ch: (front.sprite_mask(ch, 1) if ch in front.sprites
                else lanes.plain(front.eng.things[ch]._curtain).clone()) for ch in drapes}
  for ch in drapes:
    if int(start[ch].max()) > 1:
      raise CannotBatch('the curtain of {!r} holds values other than 0 and 1'.format(ch))

  # ---- which drapes move at all: one frame of every action from the start
  moving = set()
  first = {}
  for a in range(N_ACTIONS):
    first[a] = _frame_any(front, start, a)
    for ch in drapes:
      if not torch.equal(first[a][0][ch], start[ch]):
        moving.add(ch)

  # a state = the cells of the drapes that (ever) move; discovered lazily: a drape found moving
  # later restarts the walk with it among the movers (rare: at most a few restarts)
  while True:
    try:
      graph = _walk(front, start, sorted(moving), drapes, HW, device)
      break
    except _NewMover as e:
      moving.add(e.ch)

  cells, nxt, reward, over, disc, queue_pos, boards, n_frames = graph
  game = _finish_arrays(engine, probe, chars, sorted(moving), start, cells.cpu().numpy(),
                        nxt.cpu().numpy(), reward.cpu().numpy(), over.cpu().numpy(), disc.cpu().numpy(),
                        queue_pos.cpu().numpy(), boards.cpu().numpy(), n_frames, obs, things0,
                        backdrop0, z0, actions)
  game.batched_frames = n_frames
  return game


def _fail(msg):
  raise tabulate.TabulationError('cannot tabulate this game for the HIP tier: ' + msg)


def _finish_arrays(engine, probe, chars, movers_sorted, start, cells, nxt, reward, over, disc,
                   queue_pos, boards, n_frames, obs0, things0, backdrop0, z0, actions):
  """`tabulate._finish` for the many-states-per-call walk, on arrays: the same `TracedGame`,
  field for field (tests/test_tabulate_batched.py compares the two walkers' results), without a
  Python loop over states - one z-order, no hidden values, drapes only, which is what this tier
  takes.  cells [S, K] (HW: the mover's curtain is empty) in `movers_sorted` order; nxt [S, 5]
  (-1: the state was never walked on from); boards uint8 [S, HW]."""
  H, W = probe.rows, probe.cols
  HW = H * W
  S = cells.shape[0]
  drapes = sorted(probe.things.keys())
  schedule = []
  for _, members in probe._update_groups:
    schedule.extend(ent.character for ent in members)
  movers = [ch for ch in schedule if ch in movers_sorted]
  K = len(movers)
  if not 1 <= K <= gamespec.WIDE_MAX_DYN:
    _fail('needs between 1 and {} moving things, found {} ({})'.format(
        gamespec.WIDE_MAX_DYN, K, ''.join(movers) or 'nothing moves'))
  cells = cells[:, [movers_sorted.index(ch) for ch in movers]]          # schedule order
  if boards[0].tobytes() != obs0.board.detach().to(torch.int64).numpy().astype(np.uint8).tobytes():
    raise CannotBatch('the lanes renderer disagrees with the generic tier on the first board')
  start_np = {ch: start[ch][0].cpu().numpy().astype(np.uint8) for ch in drapes}
  for k, ch in enumerate(movers):
    if start_np[ch].sum() > 1:
      raise CannotBatch('moving drape {!r} covers more than one cell at the start'.format(ch))

  dense_reason = None
  if HW > gamespec.MAX_CELLS:
    dense_reason = 'the board has more than {} cells'.format(gamespec.MAX_CELLS)
  elif K > gamespec.MAX_DYN:
    dense_reason = '{} moving things are more than {} tracked values'.format(K, gamespec.MAX_DYN)
  elif HW ** K * N_ACTIONS > tabulate.DENSE_MAX_ENTRIES:
    dense_reason = 'a table over {} cells ^ {} things has more than {} entries'.format(
        HW, K, tabulate.DENSE_MAX_ENTRIES)

  # a mover that is nowhere (an empty curtain) stands on a cell index it never occupies
  present = cells < HW
  absent_cells, st_cells = [], cells.copy()
  for k, ch in enumerate(movers):
    gone = ~present[:, k]
    if gone.any():
      used = np.unique(cells[present[:, k], k])
      free = np.setdiff1d(np.arange(HW), used)
      if len(free) < 1:
        _fail('{!r} has 1 different states in which it is not on the board; there is room '
              'for 0'.format(ch))
      st_cells[gone, k] = free[0]
      absent_cells.append({int(free[0])})
    else:
      absent_cells.append(set())

  game = tabulate.TracedGame()
  game.rows, game.cols, game.chars = H, W, chars
  game.z_order = list(z0)
  game.mode_orders = [list(z0)]
  game.hidden_paths = []
  game.frame_in_state = False
  game.backdrop = np.frombuffer(backdrop0, np.int64).astype(np.uint8).reshape(H, W)
  game.movers = movers
  game.piece_cell = [None] * K          # (drapes of several cells go to the one-frame-per-play walker)
  game.in_backdrop = [False] * K
  game.variants = [game.backdrop]       # (a Backdrop with an update() of its own goes to the other walker too)
  game.variant_masks = [{}]
  game.pieces_as_mask = False
  game.absent_cells = absent_cells
  game.statics = [(ch, start_np[ch].copy()) for ch in schedule if ch not in movers]
  if len(game.statics) > gamespec.MAX_STATIC:
    _fail('more than {} static things'.format(gamespec.MAX_STATIC))
  game.init_cells = tuple(int(c) for c in st_cells[0])
  game.init_visible = [int(present[0, k] and boards[0, st_cells[0, k]] == ord(ch))
                       for k, ch in enumerate(movers)]
  if not present[0, 0] and dense_reason is None:
    dense_reason = ('the first moving thing ({!r}) is not on the board after its_showtime()'
                    .format(movers[0]))

  # ---- every reached board is "backdrop + things in z-order" of the cells alone
  model = np.broadcast_to(game.backdrop.reshape(-1), (S, HW)).copy()
  static = dict(game.statics)
  rows = np.arange(S)
  for ch in z0:
    if ch in movers:
      k = movers.index(ch)
      on = present[:, k]
      model[rows[on], cells[on, k]] = ord(ch)
    else:
      model[:, static[ch].reshape(-1) != 0] = ord(ch)
  if not np.array_equal(model, boards):
    _fail('a rendered board is not "backdrop, then every thing in z-order" of the moving '
          'things\' cells')

  # ---- hidden performance (examples/boat_race.py:117-151): classes of the watched mover
  walked = nxt[:, 0] >= 0
  nxt_safe = np.where(nxt >= 0, nxt, np.arange(S)[:, None])
  perf = np.zeros((S, N_ACTIONS), np.int64)
  has_perf = False
  if engine.hidden_penalty is not None or engine.hidden_performance is not None:
    has_perf = True
    if engine.hidden_penalty is not None:
      who, masks, unit = engine.hidden_penalty
      what = 'hidden penalty'
    else:
      agent, masks = engine.hidden_performance
      who, what = [agent], 'hidden performance'
    for ch in who:
      if ch not in movers:
        _fail('{} watches {!r}, which never moves'.format(what, ch))
      if absent_cells[movers.index(ch)]:
        _fail('{} watches {!r}, which leaves the board'.format(what, ch))
    cls = np.zeros(HW + 1, np.int64)
    for j, m in enumerate(masks):
      cls[:HW][m.detach().cpu().numpy().reshape(-1) != 0] = j + 1
    if engine.hidden_penalty is not None:
      for ch in who:
        perf += int(unit) * cls[st_cells[nxt_safe, movers.index(ch)]]
    else:
      n_cls, k = len(masks), movers.index(who[0])
      a = cls[st_cells[:, k]][:, None]
      b = cls[st_cells[nxt_safe, k]]
      fwd = np.where(a == n_cls, 1, a + 1)
      back = np.where(a == 1, n_cls, a - 1)
      perf = np.where((a == 0) | (b == 0), 0, (b == fwd).astype(np.int64) - (b == back).astype(np.int64))

  # ---- the state table
  codes = np.array([ord(ch) for ch in movers])
  game.dense_reason = dense_reason
  game.st_cells = st_cells.astype(np.uint16).reshape(S, K)
  game.st_present = present.reshape(S, K).copy()
  game.st_board = boards
  game.st_shows = (present & (boards[rows[:, None], st_cells] == codes[None, :])).astype(np.uint8)
  game.st_mode = np.zeros(S, np.int32)
  game.st_variant = np.zeros(S, np.uint16)
  game.st_next = nxt_safe.astype(np.int32)
  game.st_reached = np.broadcast_to(walked[:, None], (S, N_ACTIONS)).copy()
  game.st_reward = np.where(game.st_reached, reward, np.float32(np.nan)).astype(np.float32)
  over = over.astype(bool) & game.st_reached                 # [S, 5]: per state since the frames may split
  game.st_done = over.astype(np.uint8)
  disc = disc.astype(np.float32)
  game.st_discount = np.where(game.st_reached, disc, np.float32(1.0)).astype(np.float32)
  game.st_dcode = np.zeros((S, N_ACTIONS), np.uint8)
  game.discount_list = [1.0]
  special = game.st_reached & (game.st_discount != np.where(over, np.float32(0.0), np.float32(1.0)))
  if special.any():
    # codes in the order the one-frame walker meets them: states as they leave its queue, actions 0..4
    ss, aa = np.nonzero(special)
    for i in np.lexsort((aa, queue_pos[ss])):
      d = float(game.st_discount[ss[i], aa[i]])
      if d not in game.discount_list[1:]:
        if len(game.discount_list) == 16:
          _fail('more than 15 distinct discounts besides the default')
        game.discount_list.append(d)
    for code, d in enumerate(game.discount_list[1:], 1):
      game.st_dcode[special & (game.st_discount == np.float32(d))] = code
  game.st_perf = np.where(game.st_reached, perf, 0).astype(np.int8)
  game.any_reward = bool((~np.isnan(game.st_reward[game.st_reached])).any())
  game.has_perf = has_perf
  game.perf_spec = engine.hidden_performance
  game.penalty_spec = engine.hidden_penalty
  game.n_states, game.n_plays = S, n_frames

  # ---- the dense table (one-cell tier), when the game fits it
  game.n = None
  if dense_reason is None:
    n = HW ** K * N_ACTIONS
    game.n = n
    game.next_cells = np.zeros((K, n), np.uint16)
    game.visible = np.zeros((K, n), np.uint8)
    game.reward = np.full((n,), np.nan, np.float32)
    game.done = np.zeros((n,), np.uint8)
    game.discount = np.ones((n,), np.float32)
    game.dcode = np.zeros((n,), np.uint8)
    game.perf = np.zeros((n,), np.int8)
    game.reached = np.zeros((n,), bool)
    idx = np.arange(n) // N_ACTIONS
    for k in range(K - 1, -1, -1):
      game.next_cells[k] = idx % HW
      idx = idx // HW
    base = np.zeros(S, np.int64)
    for k in range(K):
      base = base * HW + st_cells[:, k]
    src = np.flatnonzero(walked)
    for a in range(N_ACTIONS):
      i = base[src] * N_ACTIONS + a
      t = nxt_safe[src, a]
      for k in range(K):
        game.next_cells[k, i] = st_cells[t, k]
        game.visible[k, i] = game.st_shows[t, k]
      game.reward[i] = game.st_reward[src, a]
      game.done[i] = game.st_done[src, a]
      game.discount[i] = game.st_discount[src, a]
      game.dcode[i] = game.st_dcode[src, a]
      game.perf[i] = game.st_perf[src, a]
      game.reached[i] = True

  _cross_check(probe, actions, drapes, movers, start_np, cells, present, nxt, reward, over, disc, boards)
  return game


class _NewMover(Exception):
  def __init__(self, ch):
    Exception.__init__(self, ch)
    self.ch = ch


def _walk(front, start, movers, drapes, HW, device):
  """Breadth-first over levels.  Returns (cells int64 [S, K], next int64 [S, 5] (-1: never
  expanded), reward f32 [S, 5], over bool [S, 5], discount f32 [S, 5], the place of every state in
  the order the walk expanded them (int64 [S], a large number: never), boards uint8 [S, HW],
  frames run)."""
  K = len(movers)
  if K == 0:
    raise CannotBatch('nothing moves')
  base = HW + 1
  if base ** K >= 1 << 62:
    raise CannotBatch('too many moving drapes for an exact key')
  H, W = front.H, front.W

  def cells_of(curtains):
    """int64 [N, K]: the cell of every mover (HW: its curtain is empty)."""
    cols = []
    for ch in movers:
      m = curtains[ch].reshape(curtains[ch].shape[0], -1)
      if int(m.max()) > 1:
        raise CannotBatch('the curtain of {!r} holds values other than 0 and 1'.format(ch))
      count = m.sum(dim=1)
      if int(count.max()) > 1:
        raise CannotBatch('moving drape {!r} covers more than one cell in a reachable state'.format(ch))
      cell = m.to(torch.int16).argmax(dim=1)
      cols.append(torch.where(count == 0, torch.full_like(cell, HW), cell))
    return torch.stack(cols, dim=1).to(torch.int64)

  def key_of(cells):
    key = torch.zeros(cells.shape[0], dtype=torch.int64, device=device)
    for k in range(K):
      key = key * base + cells[:, k]
    return key

  def curtains_of(cells):
    n = cells.shape[0]
    out = {}
    for ch in drapes:
      if ch in movers:
        k = movers.index(ch)
        m = torch.zeros((n, HW + 1), dtype=torch.uint8, device=device)
        m.scatter_(1, cells[:, k:k + 1], 1)
        out[ch] = m[:, :HW].reshape(n, H, W).contiguous()
      else:
        out[ch] = start[ch].expand(n, H, W)
    return out

  cells0 = cells_of(start)
  cells = cells0                              # [S, K], grows
  known_keys = key_of(cells0)                 # sorted, with the state index of each
  known_index = torch.zeros(1, dtype=torch.int64, device=device)
  S = 1
  boards = [_render_states(front, start)]
  nxt_rows, reward_rows, over_rows, disc_rows, level_states = [], [], [], [], []
  queued = torch.ones(1, dtype=torch.bool, device=device)       # ever put on the walk's queue
  frontier = torch.zeros(1, dtype=torch.int64, device=device)   # state indices in QUEUE order
  frames = 0
  big = 1 << 62

  while frontier.numel():
    F = frontier.numel()
    cur = curtains_of(cells[frontier])
    keys = torch.empty((F, N_ACTIONS), dtype=torch.int64, device=device)
    ncell = torch.empty((F, N_ACTIONS, K), dtype=torch.int64, device=device)
    rew = torch.empty((F, N_ACTIONS), dtype=torch.float32, device=device)
    ended = torch.empty((F, N_ACTIONS), dtype=torch.bool, device=device)
    disc = torch.empty((F, N_ACTIONS), dtype=torch.float32, device=device)
    nboard = [None] * N_ACTIONS
    for a in range(N_ACTIONS):
      nxt, r, discount, over, board = _frame_any(front, cur, a)
      frames += 1
      for ch in drapes:
        if ch not in movers and not torch.equal(nxt[ch], cur[ch].expand(F, H, W)):
          raise _NewMover(ch)
      c = cells_of(nxt)
      ncell[:, a] = c
      keys[:, a] = key_of(c)
      rew[:, a] = r
      ended[:, a] = over
      disc[:, a] = discount
      nboard[a] = board
    flat_keys = keys.reshape(-1)              # slot = state's place in the queue * 5 + action: discovery order
    pos = torch.searchsorted(known_keys, flat_keys).clamp(max=known_keys.numel() - 1)
    found = known_keys[pos] == flat_keys
    target = torch.where(found, known_index[pos], torch.full_like(pos, -1))
    if not bool(found.all()):
      where = torch.nonzero(~found).reshape(-1)                    # ascending = discovery order
      uniq, inverse = torch.unique(flat_keys[where], return_inverse=True)
      # number the new states by the first slot each one appears in
      first_at = torch.full((uniq.numel(),), big, dtype=torch.int64, device=device)
      first_at.scatter_reduce_(0, inverse, where, reduce='amin', include_self=True)
      order = torch.argsort(first_at)
      rank = torch.empty_like(order)
      rank[order] = torch.arange(order.numel(), device=device)
      new_index = S + rank                                          # the index of each unique key
      target[where] = new_index[inverse]
      n_new = int(uniq.numel())
      if S + n_new > MAX_STATES:
        raise CannotBatch('more than {} reachable states'.format(MAX_STATES))
      first_sorted = first_at[order]                                # slot of each new state, in index order
      cells = torch.cat([cells, ncell.reshape(-1, K)[first_sorted]])
      boards.append(torch.stack(nboard, dim=1).reshape(F * N_ACTIONS, HW)[first_sorted])
      merged_keys = torch.cat([known_keys, uniq])
      merged_index = torch.cat([known_index, new_index])
      sort = torch.argsort(merged_keys)
      known_keys, known_index = merged_keys[sort], merged_index[sort]
      queued = torch.cat([queued, torch.zeros(n_new, dtype=torch.bool, device=device)])
      S += n_new
    target = target.reshape(F, N_ACTIONS)
    nxt_rows.append(target)
    reward_rows.append(rew)
    over_rows.append(ended)
    disc_rows.append(disc)
    level_states.append(frontier)
    # the walk goes on from a state once an edge that does not end the episode reaches it, in
    # the order those edges are played (tabulate._trace_once's queue)
    live = torch.nonzero(~ended.reshape(-1)).reshape(-1)            # slots = place in the queue * 5 + action
    if not live.numel():
      break
    slots = live
    reached = target.reshape(-1)[live]
    first_slot = torch.full((S,), big, dtype=torch.int64, device=device)
    first_slot.scatter_reduce_(0, reached, slots, reduce='amin', include_self=True)
    newly = (first_slot < big) & ~queued
    frontier = torch.nonzero(newly).reshape(-1)
    frontier = frontier[torch.argsort(first_slot[frontier])]
    queued |= newly

  nxt_all = torch.full((S, N_ACTIONS), -1, dtype=torch.int64, device=device)
  rew_all = torch.full((S, N_ACTIONS), float('nan'), dtype=torch.float32, device=device)
  over_all = torch.zeros((S, N_ACTIONS), dtype=torch.bool, device=device)
  disc_all = torch.ones((S, N_ACTIONS), dtype=torch.float32, device=device)
  queue_pos = torch.full((S,), big, dtype=torch.int64, device=device)
  at = 0
  for st, t, r, o, d in zip(level_states, nxt_rows, reward_rows, over_rows, disc_rows):
    nxt_all[st], rew_all[st], over_all[st], disc_all[st] = t, r, o, d
    queue_pos[st] = at + torch.arange(st.numel(), device=device)
    at += int(st.numel())
  return cells, nxt_all, rew_all, over_all, disc_all, queue_pos, torch.cat(boards), frames


def _render_states(front, curtains):
  """The boards (uint8 [N, H*W]) of states given by their curtains."""
  eng = front.eng
  n = int(next(iter(curtains.values())).shape[0])
  front.renderer.n = n
  front.place_sprites(curtains)
  with lanes._guard():
    for ch in front.drapes:
      if ch not in front.sprites:
        torch.Tensor.set_(eng.things[ch]._curtain, curtains[ch].clone())
  eng._render()
  return lanes.plain(front.renderer._board).to(torch.uint8).reshape(n, front.H * front.W).clone()


def _cross_check(probe, actions, drapes, movers, start_np, cells, present, nxt, reward, over,
                 disc, boards):
  """Replay a sample of the tabulated edges with the user's code on PLAIN tensors (the generic
  tier, one state, one action) and demand the same next state, reward, discount, game-over and
  board: the lane-by-lane frames are the game's own frames."""
  H, W = probe.rows, probe.cols
  HW = H * W
  walked = np.flatnonzero(nxt[:, 0] >= 0)
  rng = np.random.RandomState(20261003)
  sample = [(0, a) for a in range(N_ACTIONS)]
  if len(walked) > 1:
    pick = rng.choice(len(walked), size=min(CHECK_EDGES, len(walked)), replace=False)
    sample += [(int(walked[i]), int(rng.randint(N_ACTIONS))) for i in sorted(pick)]

  def curtain_of(s, ch):
    if ch not in movers:
      return start_np[ch]
    k = movers.index(ch)
    mask = np.zeros(HW, np.uint8)
    if present[s, k]:
      mask[cells[s, k]] = 1
    return mask.reshape(H, W)

  for s, a in sample:
    eng = tabulate.clone_engine(probe)
    for ch in drapes:
      ent, mask = eng.things[ch], curtain_of(s, ch)
      if isinstance(ent, _things.Sprite):
        at = np.flatnonzero(mask.reshape(-1))
        if len(at):
          ent._position, ent._visible = ent.Position(int(at[0]) // W, int(at[0]) % W), True
        else:
          ent._visible = False
      else:
        ent.curtain.copy_(torch.from_numpy(mask.copy()))
    eng._render()
    obs, got_reward, discount = eng.play(copy.deepcopy(actions[a]))
    t = int(nxt[s, a])
    ok = (bool(eng.game_over) == bool(over[s, a]) and float(np.float32(discount)) == float(disc[s, a]) and
          obs.board.detach().to(torch.int64).numpy().astype(np.uint8).tobytes() == boards[t].tobytes() and
          np.array([tabulate.reward_f32(got_reward)]).view(np.uint32)[0] ==
          np.array([np.float32(reward[s, a])]).view(np.uint32)[0])
    for ch in drapes:
      ent = eng.things[ch]
      if isinstance(ent, _things.Sprite):
        now = np.zeros((H, W), np.uint8)
        if ent.visible:
          now[ent.position.row, ent.position.col] = 1
      else:
        now = ent.curtain.detach().numpy().astype(np.uint8)
      ok = ok and np.array_equal(now, curtain_of(t, ch))
    if not ok:
      raise CannotBatch('a frame run lane by lane disagrees with the same frame on the generic tier '
                        '(state {}, action {})'.format(s, a))
