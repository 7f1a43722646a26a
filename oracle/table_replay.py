"""Walk a tabulated game on the CPU (numpy): the checker for host-tabulated games.

TEST INFRASTRUCTURE (only tests/ and bench.py's cpu_baseline may import this).

`campx_amd.tabulate.trace()` turns a game of arbitrary Python classes into a table
(next cells, visibility, reward, done, perf per (cells, action)); the HIP kernels walk
that table for B environments.  This module walks the same table with plain numpy
indexing, one frame at a time, with the batched engine's episode rule: an environment
whose episode ended is rebuilt from the start state before its next action (what the
reference's driver does with `make_game()` per episode, examples/reinforce.py:122).
Observations are rendered the way the reference renders them - backdrop, then every
thing in (the environment's current) z-order onto a flat board, layers by equality with the board
(campx/engine.py:306-324, campx/rendering.py:204-215) - from the cells alone.

It shares no code with the HIP path; that the table itself is right is pinned elsewhere
(tests/test_tabulate.py: against the generic tier's own frames and against the
reference-generated `tests/golden/boat_race_table.npz`).
"""

import numpy as np

N_ACTIONS = 5


class TableWalker(object):
  """State of B environments of a `tabulate.TracedGame`."""

  def __init__(self, game, batch):
    self.game = game
    self.B = int(batch)
    self.K = game.n_tracked          # the movers, plus the z-order mode of a re-ordering game
    self.HW = game.rows * game.cols
    self.cells = np.tile(np.array(game.init_cells, np.int64)[:, None], (1, self.B))
    self.over = np.zeros(self.B, bool)
    self.ret = np.zeros(self.B, np.float32)

  def _index(self, cells, actions):
    idx = np.zeros(self.B, np.int64)
    for k in range(self.K):
      idx = idx * self.HW + cells[k]
    return idx * N_ACTIONS + actions

  def rollout(self, actions, reset_first=False):
    """actions int8 [T, B] -> dict(cells [K, T, B], visible [K, T, B], reward, discount,
    done, perf [T, B])."""
    g = self.game
    actions = np.asarray(actions)
    T = actions.shape[0]
    out = dict(cells=np.zeros((self.K, T, self.B), np.uint16),
               visible=np.zeros((self.K, T, self.B), np.uint8),
               reward=np.zeros((T, self.B), np.float32),
               discount=np.zeros((T, self.B), np.float32),
               done=np.zeros((T, self.B), np.uint8),
               perf=np.zeros((T, self.B), np.int8))
    if reset_first:
      self.over[:] = True
    init = np.array(g.init_cells, np.int64)[:, None]
    for t in range(T):
      a = actions[t].astype(np.int64)
      a = np.where((a < 0) | (a > 4), 4, a)          # an id outside 0..4 acts as "stay"
      cells = np.where(self.over[None, :], init, self.cells)
      self.ret = np.where(self.over, np.float32(0), self.ret).astype(np.float32)
      idx = self._index(cells, a)
      assert g.reached[idx].all(), 'the walk left the tabulated (reachable) entries'
      self.cells = g.next_cells[:, idx].astype(np.int64)
      reward = g.reward[idx]
      self.ret = (self.ret + np.where(np.isnan(reward), np.float32(0), reward)).astype(np.float32)
      self.over = g.done[idx] != 0
      out['cells'][:, t] = self.cells
      out['visible'][:, t] = g.visible[:, idx]
      out['reward'][t] = reward
      out['done'][t] = self.over
      out['discount'][t] = g.discount[idx]
      out['perf'][t] = g.perf[idx]
    return out

  def render(self, cells):
    """cells [K, N] -> (board int8 [N, H, W], layered int8 [N, L, H, W]).  A game that
    re-orders its things (campx/engine.py:242-281) is painted in each environment's own
    order, the one its last tracked value names."""
    g = self.game
    N = cells.shape[1]
    board = np.tile(g.backdrop.reshape(1, -1).astype(np.int16), (N, 1))
    static = dict(g.statics)
    who = {}          # character -> its movers (several: the pieces of a drape of several cells)
    for k, ch in enumerate(g.movers):
      who.setdefault(ch, []).append(k)
    modes = cells[len(g.movers)] if len(g.mode_orders) > 1 else np.zeros(N, np.int64)
    for m, order in enumerate(g.mode_orders):
      rows = np.flatnonzero(modes == m)
      for ch in order:                                # back to front
        if ch in who:
          for k in who[ch]:
            here = rows[~np.isin(cells[k][rows], sorted(g.absent_cells[k]))]   # (on the board)
            board[here, cells[k][here]] = ord(ch)
        else:
          board[np.ix_(rows, np.flatnonzero(static[ch].reshape(-1)))] = ord(ch)
    layered = np.stack([(board == ord(ch)) for ch in g.chars], axis=1).astype(np.int8)
    return (board.astype(np.int8).reshape(N, g.rows, g.cols),
            layered.reshape(N, len(g.chars), g.rows, g.cols))


class StateWalker(object):
  """B environments of a `tabulate.TracedGame` walked through its STATE table
  ((state, action) -> state; state 0 is the `its_showtime()` state), the checker of the wide
  tier.  What a state looks like is not recomputed: `game.st_board[s]` is the board the user's
  own classes rendered when the tabulator stood in state s on the generic tier (and checked
  against "backdrop, then things in z-order"), and the layers are that board compared with
  each character (campx/rendering.py:204-215)."""

  def __init__(self, game, batch):
    self.game = game
    self.B = int(batch)
    self.state = np.zeros(self.B, np.int64)
    self.over = np.zeros(self.B, bool)
    self.ret = np.zeros(self.B, np.float32)

  def rollout(self, actions, reset_first=False):
    """actions int8 [T, B] -> dict(state [T, B], cells / shows [K, T, B], reward, discount,
    done, perf [T, B])."""
    g = self.game
    actions = np.asarray(actions)
    T, K = actions.shape[0], len(g.movers)
    out = dict(state=np.zeros((T, self.B), np.int64),
               cells=np.zeros((K, T, self.B), np.uint16),
               shows=np.zeros((K, T, self.B), np.uint8),
               reward=np.zeros((T, self.B), np.float32),
               discount=np.zeros((T, self.B), np.float32),
               done=np.zeros((T, self.B), np.uint8),
               perf=np.zeros((T, self.B), np.int8))
    if reset_first:
      self.over[:] = True
    for t in range(T):
      a = actions[t].astype(np.int64)
      a = np.where((a < 0) | (a > 4), 4, a)          # an id outside 0..4 acts as "stay"
      s = np.where(self.over, 0, self.state)
      self.ret = np.where(self.over, np.float32(0), self.ret).astype(np.float32)
      assert g.st_reached[s, a].all(), 'the walk left the tabulated (reachable) entries'
      self.state = g.st_next[s, a].astype(np.int64)
      reward = g.st_reward[s, a]
      self.ret = (self.ret + np.where(np.isnan(reward), np.float32(0), reward)).astype(np.float32)
      self.over = g.st_done[s, a] != 0
      out['state'][t] = self.state
      out['cells'][:, t] = np.where(g.st_present[self.state], g.st_cells[self.state], 0).T
      out['shows'][:, t] = g.st_shows[self.state].T
      out['reward'][t] = reward
      out['done'][t] = self.over
      out['discount'][t] = g.st_discount[s, a]
      out['perf'][t] = g.st_perf[s, a]
    return out

  def render(self, states):
    """state indices [N] -> (board int8 [N, H, W], layered int8 [N, L, H, W])."""
    g = self.game
    board = g.st_board[np.asarray(states, np.int64)]
    layered = np.stack([(board == ord(ch)) for ch in g.chars], axis=1).astype(np.int8)
    n = board.shape[0]
    return (board.astype(np.int8).reshape(n, g.rows, g.cols),
            layered.reshape(n, len(g.chars), g.rows, g.cols))
