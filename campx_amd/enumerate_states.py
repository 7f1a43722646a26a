"""State tables of RULE games on big boards, enumerated on the device.

A game of `campx_amd.rules` classes on a board above 128 cells (campx/engine.py:31 sets no size
limit; PyColab-sized boards are 16x16 and up) cannot take the one-cell tier, whose tables are
indexed by 7-bit cells, and runs from its STATE table instead (`wide`, csrc/k_wide.hip).  Until
round 4 that table was filled on the host by running the classes' Python on the generic tier, one
frame per (state, action) (`tabulate.trace`): fine for a maze's few hundred states, hopeless for a
sokoban whose agent and two boxes reach millions.  Here the same rules, lowered to `CampxRule`
records as for the one-cell tier (`gamespec.lower(desc, wide=True)`), are applied by
`campx_wide_enumerate_launch()` (include/campx_hip.h; the update pass of campx/engine.py:168-208
as the rule interpreter kernel has it) to whole frontiers at once:

    frontier = {the state its_showtime() leaves};  seen = frontier
    while frontier:   next = step(frontier x 5 actions);  frontier = unique(next) - seen;  seen |= frontier

with states packed as 10 bits per moving thing in an int64 and the set operations done by torch
on the device (sort / unique / searchsorted); one more pass over all of `seen` then yields the
table - next state, reward, done, "is the character its cell shows" bits, hidden performance.
The result is a `tabulate.TracedGame` in its state-table form, which `tabulate.to_wide_spec` and
`wide.WideGame` take as they take a host-tabulated one.  States that end the episode are expanded
like any other (what lies behind them is unreachable in play and costs only rows).

Needs a HIP device (it is only ever called for a batched Engine); no CPU path.
"""

import ctypes

import numpy as np
import torch

from . import _hip
from . import gamespec
from . import tabulate

N_ACTIONS = gamespec.N_ACTIONS
CELL_BITS = 10


class EnumerationError(tabulate.TabulationError):
  pass


def _pack(cells):
  """int64 [N, K] cells -> int64 [N] keys."""
  key = torch.zeros(cells.shape[0], dtype=torch.int64, device=cells.device)
  for k in range(cells.shape[1]):
    key |= cells[:, k] << (CELL_BITS * k)
  return key


def _unpack(keys, K):
  return torch.stack([(keys >> (CELL_BITS * k)) & ((1 << CELL_BITS) - 1) for k in range(K)], dim=1)


class _Stepper(object):
  """campx_wide_enumerate_launch() for one game: device tables + the host rule block."""

  def __init__(self, lowered, device):
    self.device = device
    self.K = int(lowered.n_dyn)
    self._tables = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in
                    (lowered.static_top_layer, lowered.static_top_z, lowered.static_cover,
                     lowered.cell_class)]
    r = gamespec.CampxWideRules()
    r.magic, r.version = gamespec.SPEC_MAGIC, gamespec.SPEC_VERSION
    r.rows, r.cols, r.n_layers = lowered.rows, lowered.cols, lowered.n_layers
    r.n_dyn, r.n_rules, r.any_reward = lowered.n_dyn, lowered.n_rules, lowered.any_reward
    r.perf_dyn, r.perf_n = lowered.perf_dyn, getattr(lowered, 'perf_n', 0)
    r.perf_mode, r.perf_mask = lowered.perf_mode, lowered.perf_mask
    r.perf_scale, r.perf_offset = lowered.perf_scale, lowered.perf_offset
    for d in range(self.K):
      r.dyn_layer[d], r.dyn_z[d] = int(lowered.dyn_layer[d]), int(lowered.dyn_z[d])
    ctypes.memmove(ctypes.addressof(r.rules), ctypes.addressof(lowered.rules), ctypes.sizeof(r.rules))
    r.top_layer, r.top_z, r.cover, r.cell_class = (t.data_ptr() for t in self._tables)
    self.rules = r
    self.has_perf = lowered.perf_dyn >= 0

  def step(self, keys, want_all=False):
    """int64 [N] state keys -> next keys int64 [N, 5] (and reward, done, shows, perf)."""
    N, K, dev = int(keys.shape[0]), self.K, self.device
    cells = _unpack(keys, K).to(torch.int16).contiguous()
    nxt = torch.empty((N, N_ACTIONS, K), dtype=torch.int16, device=dev)
    reward = torch.empty((N, N_ACTIONS), dtype=torch.float32, device=dev)
    done = torch.empty((N, N_ACTIONS), dtype=torch.uint8, device=dev)
    shows = torch.empty((N, N_ACTIONS), dtype=torch.uint8, device=dev)
    perf = torch.empty((N, N_ACTIONS), dtype=torch.int8, device=dev) if self.has_perf else None
    with torch.cuda.device(dev):
      _hip.check(_hip.lib.campx_wide_enumerate_launch(
          ctypes.byref(self.rules), ctypes.c_void_p(cells.data_ptr()), N,
          ctypes.c_void_p(nxt.data_ptr()), ctypes.c_void_p(reward.data_ptr()),
          ctypes.c_void_p(done.data_ptr()), ctypes.c_void_p(shows.data_ptr()),
          ctypes.c_void_p(perf.data_ptr() if perf is not None else 0),
          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
          'campx_wide_enumerate_launch')
    nk = _pack(nxt.to(torch.int64).reshape(N * N_ACTIONS, K) & 0xffff).reshape(N, N_ACTIONS)
    if want_all:
      return nk, nxt, reward, done, shows, perf
    return nk


def enumerate_rule_game(engine, device, max_states=None):
  """A set-up (not started) rule-class `Engine` -> `tabulate.TracedGame` (state-table form),
  every reachable state found on `device`."""
  if not torch.cuda.is_available():
    raise RuntimeError('state enumeration needs a HIP device')
  device = torch.device('cuda' if device is None else device)
  if device.index is None:
    device = torch.device('cuda', torch.cuda.current_device())
  max_states = gamespec.WIDE_MAX_STATES if max_states is None else int(max_states)
  desc = gamespec.describe(engine)
  low = gamespec.lower(desc, wide=True)
  H, W, K = desc.rows, desc.cols, int(low.n_dyn)
  stepper = _Stepper(low, device)
  init = torch.tensor([[int(low.dyn_row0[d]) * W + int(low.dyn_col0[d]) for d in range(K)]],
                      dtype=torch.int64, device=device)
  first = _pack(init)
  seen = first.clone()                 # sorted, unique
  frontier = first
  levels = 0
  while frontier.numel():
    nk = torch.unique(stepper.step(frontier).reshape(-1))            # sorted
    pos = torch.searchsorted(seen, nk).clamp_(max=seen.numel() - 1)
    fresh = nk[seen[pos] != nk]
    if fresh.numel():
      seen = torch.sort(torch.cat([seen, fresh])).values
    if seen.numel() > max_states:
      raise EnumerationError(
          'cannot tabulate this game for the HIP tier: more than {} reachable states'
          .format(max_states))
    frontier = fresh
    levels += 1
  # ---- the table: state 0 is the its_showtime() state, the others in key order
  at0 = int(torch.searchsorted(seen, first)[0])
  order = torch.cat([first, seen[:at0], seen[at0 + 1:]])
  S = int(order.numel())

  def index_of(keys):
    pos = torch.searchsorted(seen, keys)
    return torch.where(pos == at0, torch.zeros_like(pos), torch.where(pos < at0, pos + 1, pos))

  nk, _, reward, done, shows, perf = stepper.step(order, want_all=True)
  nxt_index = index_of(nk.reshape(-1)).reshape(S, N_ACTIONS).to(torch.int32)
  cells = _unpack(order, K)                                      # int64 [S, K]
  # whether a thing shows in state s: every state but the first is somebody's successor -
  # scatter the successors' bits (all edges into a state agree: it is a function of the cells)
  shows_of = torch.zeros((S,), dtype=torch.uint8, device=device)
  shows_of[nxt_index.reshape(-1).to(torch.int64)] = shows.reshape(-1)
  # ... and the first one from its own "stay where you are"-free reading: the scenery and
  # z-order alone (nothing has moved yet)
  shows0 = _shows_of_cells(desc, low, [int(c) for c in init[0]])
  if not bool((nxt_index == 0).any()):
    shows_of[0] = shows0
  elif int(shows_of[0]) != shows0:
    raise EnumerationError('internal: the first state shows differently when re-entered')

  game = tabulate.TracedGame()
  game.rows, game.cols, game.chars = H, W, list(desc.chars)
  dynamic = [e for e in desc.entities if e.moves]
  game.movers = [e.char for e in dynamic]
  game.piece_cell = [None] * len(dynamic)
  game.in_backdrop = [False] * len(dynamic)
  game.statics = [(e.char, e.mask) for e in desc.entities if not e.moves]
  game.z_order = list(desc.z_order)
  game.mode_orders = [list(desc.z_order)]
  game.hidden_paths, game.frame_in_state = [], False
  game.absent_cells = [set() for _ in dynamic]
  game.backdrop = np.asarray(desc.backdrop, np.uint8)
  game.init_cells = tuple(int(c) for c in init[0])
  game.init_visible = [(shows0 >> k) & 1 for k in range(K)]
  game.dense_reason = 'a rule game on a board of more than {} cells'.format(gamespec.MAX_CELLS)
  game.n = None
  game.st_cells = cells.to(torch.int32).cpu().numpy().astype(np.uint16)
  game.st_present = np.ones((S, K), bool)
  bits = shows_of.cpu().numpy()
  game.st_shows = np.stack([(bits >> k) & 1 for k in range(K)], axis=1).astype(np.uint8)
  game.st_mode = np.zeros(S, np.int32)
  game.st_variant = np.zeros(S, np.uint16)
  game.variants = [game.backdrop]
  game.variant_masks = [{}]
  game.pieces_as_mask = False
  game.st_next = nxt_index.cpu().numpy()
  game.st_reward = reward.cpu().numpy()
  game.st_done = done.cpu().numpy()
  game.st_discount = np.where(game.st_done != 0, np.float32(0), np.float32(1)).astype(np.float32)
  game.st_dcode = np.zeros((S, N_ACTIONS), np.uint8)
  game.st_perf = perf.cpu().numpy() if perf is not None else np.zeros((S, N_ACTIONS), np.int8)
  game.st_reached = np.ones((S, N_ACTIONS), bool)
  game.st_board = None               # (millions of boards: the checker of these games is the C oracle)
  game.discount_list = [1.0]
  game.any_reward = bool(low.any_reward)
  game.has_perf = perf is not None
  game.perf_spec, game.penalty_spec = engine.hidden_performance, engine.hidden_penalty
  game.n_states, game.n_plays, game.n_levels = S, 0, levels
  return game


def _shows_of_cells(desc, low, cells):
  """Bit d: moving thing d is the character its cell shows when the things stand at `cells`
  (campx/engine.py:306-324: the front-most of backdrop, static drapes and moving things)."""
  bits = 0
  for d, cell in enumerate(cells):
    z_top, who = int(low.static_top_z[cell]), None
    for k, other in enumerate(cells):
      if other == cell and int(low.dyn_z[k]) > z_top:
        z_top, who = int(low.dyn_z[k]), k
    bits |= int(who == d) << d
  return bits
