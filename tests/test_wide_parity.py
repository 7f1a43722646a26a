"""The wide tier (csrc/k_wide.hip): games run from their STATE table - boards above 128 cells,
up to four things that show, hidden values behind them - walked by `wide_update_kernel`,
rendered by the one-cell tier's render kernel from a 16-bit trace.

Chain of evidence: `tests/golden/maze_*.npz` / `traced_*.npz` are the games run by the
REFERENCE engine (make_golden.py, make_traced_golden.py) -> this repo's generic tier and the
tabulated tables reproduce them on the CPU -> on the GPU the HIP path is compared with
`oracle/table_replay.py` walking the same table and with `oracle/campx_oracle.c` running the
rules, at full batch size, frame by frame, and with the goldens themselves.
"""

import ctypes
import os

import numpy as np
import pytest
import torch

from campx_amd import gamespec, tabulate
from campx_amd.games import maze

import traced_games

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MAZES = [(16, 16), (15, 17)]


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def _golden(rows, cols):
  with np.load(os.path.join(GOLDEN_DIR, 'maze_{}x{}.npz'.format(rows, cols))) as f:
    return {k: f[k] for k in f.files}


_TRACED = {}


def _traced(rows, cols):
  if (rows, cols) not in _TRACED:
    _TRACED[(rows, cols)] = tabulate.trace(maze.build(rows, cols))
  return _TRACED[(rows, cols)]


# ------------------------------------------------------------------------------- CPU

@pytest.mark.parametrize('rows,cols', MAZES)
def test_reference_engine_goldens_on_the_generic_tier_and_through_the_table(rows, cols):
  from oracle.table_replay import StateWalker
  gold = _golden(rows, cols)
  T, N = gold['actions'].shape
  onehot = tabulate.default_actions()
  for n in range(2):                       # the two environments that reach the goal
    game = maze.build(rows, cols)
    obs, _, _ = game.its_showtime()
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      if game.game_over:
        game = maze.build(rows, cols)
        game.its_showtime()
      obs, reward, discount = game.play(onehot[int(gold['actions'][t, n])])
      assert np.array_equal(obs.board.numpy(), gold['board'][t + 1, n].astype(np.uint8)), (n, t)
      assert _same(np.float32(float(reward)), gold['reward'][t, n])
      assert np.float32(discount) == gold['discount'][t, n]
      assert int(game.game_over) == gold['done'][t, n]
  traced = _traced(rows, cols)
  assert traced.movers == ['A'] and traced.n_tracked == 1
  assert traced.dense_reason is not None and traced.n is None      # no cell-indexed table
  assert [ord(c) for c in traced.chars] == gold['chars'].tolist()
  walker = StateWalker(traced, N)
  want = walker.rollout(gold['actions'], reset_first=True)
  for k in ('reward', 'discount', 'done'):
    assert _same(want[k], gold[k]), k
  assert want['done'].sum() == 2
  for t in range(T):
    board, layered = walker.render(want['state'][t])
    assert np.array_equal(board, gold['board'][t + 1]), t
    assert np.array_equal(layered, gold['layered'][t + 1].astype(np.int8)), t


def test_wide_spec_of_the_maze_validates_and_says_what_the_table_says():
  from campx_amd import _hip
  traced = _traced(16, 16)
  spec, arrays = tabulate.to_wide_spec(traced)
  assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == 0
  S = traced.n_states
  assert (spec.rows, spec.cols, spec.n_layers, spec.n_dyn, spec.n_states) == (16, 16, 6, 1, S)
  assert spec.any_reward == 1 and chr(spec.layer_char[spec.dyn_layer[0]]) == 'A'
  assert arrays['state_cells'][0, 0] == 17                  # state 0: 'A' at (1, 1), showing
  n_bytes = _hip.lib.campx_wide_tables_bytes(ctypes.byref(spec))
  R = 6 * 256
  tables = (S * 5 * 8 + 15) // 16 * 16 + S * 16                # entries, eight trace entries per state
  assert n_bytes == (tables + S * 5 + 15) // 16 * 16 + 16 * (R + 16) + 16 * (256 + 16)
  # the table says what the art says: walls stop, '*' tiles pay +1 on entering
  art = maze.maze_art(16, 16)
  where = {int(c): s for s, c in enumerate(traced.st_cells[:, 0])}
  for cell in (17, 18, 33):
    for a, (dr, dc) in enumerate([(0, -1), (0, 1), (-1, 0), (1, 0), (0, 0)]):
      r, c = divmod(cell, 16)
      nxt = cell if art[r + dr][c + dc] == '#' else (r + dr) * 16 + c + dc
      s = where[cell]
      assert traced.st_cells[traced.st_next[s, a], 0] == nxt and traced.st_done[s, a] == 0
      assert traced.st_reward[s, a] == -1.0 + (1.0 if art[r + dr][c + dc] == '*' and nxt != cell else 0.0)
  bad, keep = tabulate.to_wide_spec(traced)
  keep['next_state'][0, 0] = S                             # a state that does not exist
  assert _hip.lib.campx_wide_spec_validate(ctypes.byref(bad)) == -2
  bad, keep = tabulate.to_wide_spec(traced)
  keep['state_cells'][3, 0] = 300                          # a cell off the board
  assert _hip.lib.campx_wide_spec_validate(ctypes.byref(bad)) == -2
  bad, keep = tabulate.to_wide_spec(traced)
  bad.rows, bad.cols = 3, 4                                # fewer than 16 cells
  assert _hip.lib.campx_wide_spec_validate(ctypes.byref(bad)) == -2


def _big_vault(**where):
  """tests/traced_games.py vault (walker, key, door, hidden gem: four things that show, three
  of which come and go) on a 12x16 board."""
  art = ['#' * 16] + ['#' + ' ' * 14 + '#' for _ in range(10)] + ['#' * 16]
  art[1] = '#A    k #     $#'
  for r in range(2, 11):
    art[r] = art[r][:8] + '#' + art[r][9:]
  art[6] = art[6][:8] + 'D' + art[6][9:]
  return traced_games.ascii_art_to_game(
      art, what_lies_beneath=' ', sprites={'$': traced_games.Gem},
      drapes={'A': traced_games.VaultWalker, 'k': traced_games.Key, 'D': traced_games.Door,
              '#': traced_games.things.FixedDrape},
      z_order='k$DA#', update_schedule='AkD$#', **where)


def _big_burrow(**where):
  """tests/traced_games.py burrow (a mole that changes its place in the z-order) on 10x20."""
  art = ['#' * 20] + ['#' + ' ' * 18 + '#' for _ in range(8)] + ['#' * 20]
  art[1] = '#A   ======   $    #'
  art[3] = '#  d ======  u     #'
  art[5] = '#    ======        #'
  return traced_games.ascii_art_to_game(
      art, what_lies_beneath=' ',
      drapes={'A': traced_games.Mole, '#': traced_games.things.FixedDrape,
              '=': traced_games.things.FixedDrape, 'd': traced_games.things.FixedDrape,
              'u': traced_games.things.FixedDrape, '$': traced_games.things.FixedDrape},
      z_order='du$=A#', update_schedule='A#=du$', **where)


BIG_GAMES = {'vault': _big_vault, 'burrow': _big_burrow}


@pytest.mark.parametrize('name', sorted(BIG_GAMES))
def test_many_tracked_values_on_a_big_board_tabulate_to_a_state_table(name):
  """Four things that come and go, or a z-order that changes, on more than 128 cells: no
  cell-indexed table exists; the state table predicts the generic tier frame by frame."""
  from oracle.table_replay import StateWalker
  build = BIG_GAMES[name]
  traced = tabulate.trace(build())
  assert traced.dense_reason is not None and traced.n is None
  if name == 'vault':
    assert traced.movers == ['A', 'k', 'D', '$'] and not traced.st_present[0, 3]
    assert (~traced.st_present[:, 1]).any() and (~traced.st_present[:, 2]).any()
  else:
    assert traced.movers == ['A'] and len(traced.mode_orders) == 2
    assert (traced.st_mode == 1).any() and (traced.st_shows[traced.st_mode == 1, 0] == 0).any()
  spec, arrays = tabulate.to_wide_spec(traced)
  assert spec.n_dyn == len(traced.movers) and spec.n_states == traced.n_states
  rng = np.random.RandomState(11)
  T = 300
  actions = rng.randint(0, 5, size=(T, 1)).astype(np.int8)
  walker = StateWalker(traced, 1)
  want = walker.rollout(actions, reset_first=True)
  game = build()
  game.its_showtime()
  onehot = tabulate.default_actions()
  for t in range(T):
    if game.game_over:
      game = build()
      game.its_showtime()
    obs, reward, discount = game.play(onehot[int(actions[t, 0])])
    board, layered = walker.render(want['state'][t])
    assert np.array_equal(obs.board.numpy(), board[0].astype(np.uint8)), t
    assert np.array_equal(obs.layered_board.numpy(), layered[0]), t
    assert _same(np.float32(float(reward)), want['reward'][t, 0]), t
    assert float(discount) == want['discount'][t, 0] and int(game.game_over) == want['done'][t, 0]


def test_wide_tier_without_a_gpu_fails_loudly():
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  game = maze.build(16, 16, batch=64)
  with pytest.raises(RuntimeError, match='needs a HIP device'):
    game.its_showtime()


# ------------------------------------------------------------------------------- GPU

def _check_rollout(game, traced, actions, out, walker, want_board=True):
  want = walker.rollout(actions)
  T, B = actions.shape
  trace = out['trace'].cpu().numpy().astype(np.uint16)
  assert np.array_equal(trace >> 15, want['shows'])
  assert np.array_equal(trace & 0x3ff, want['cells'])
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), want[k]), k
  assert np.array_equal(game.fused.state.cpu().numpy(), walker.state)
  sample = np.unique(np.concatenate([np.arange(0, B, max(1, B // 61)), [B - 1]]))
  for t in range(T):
    if t in (0, 1, T // 2, T - 1):
      board, layered = walker.render(want['state'][t])
      assert np.array_equal(out['obs'][t].cpu().numpy(), layered), t
      if want_board:
        assert np.array_equal(out['board'][t].cpu().numpy(), board), t
    else:
      board, layered = walker.render(want['state'][t][sample])
      assert np.array_equal(out['obs'][t][sample].cpu().numpy(), layered), t
      if want_board:
        assert np.array_equal(out['board'][t][sample].cpu().numpy(), board), t
  return want


@pytest.mark.gpu
@pytest.mark.parametrize('rows,cols,B', [(16, 16, 65536), (15, 17, 4099), (32, 32, 1000),
                                         (12, 11, 777), (8, 127, 300)])
def test_mazes_through_the_wide_kernels_against_the_table_walker(rows, cols, B):
  from oracle.table_replay import StateWalker
  from campx_amd import wide
  T = 100 if B > 10000 else 70
  game = maze.build(rows, cols, batch=B, device='cuda')
  first, reward0, discount0 = game.its_showtime()
  f = game.fused
  assert isinstance(f, wide.WideGame) and reward0 is None and discount0 == 1.0
  traced = f.traced
  walker = StateWalker(traced, B)
  board0, layered0 = walker.render(walker.state[:8])
  assert np.array_equal(first.board[:8].cpu().numpy(), board0)
  assert np.array_equal(first.layered_board[:8].cpu().numpy(), layered0)
  assert torch.equal(first.layered_board[:1].expand(B, -1, -1, -1), first.layered_board)

  rng = np.random.RandomState(rows * 1000 + cols)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  actions[:, B // 2] = np.resize([1, 1, 3, 3, 3, 1, 0, 2], T)      # one that wanders further
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  want = _check_rollout(game, traced, actions, out, walker)
  assert _same(f.ret.cpu().numpy(), walker.ret)
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1      # each cell shows one character

  # a second rollout continues from the state the first left (no reset_first)
  more = rng.randint(0, 5, size=(9, B)).astype(np.int8)
  out2 = game.rollout(torch.from_numpy(more))
  _check_rollout(game, traced, more, out2, walker, want_board=False)

  # the same frames one play() at a time, after reset() (= a fresh game: back to the art)
  again, none, one = f.reset()
  assert none is None and one == 1.0
  assert torch.equal(again.layered_board, first.layered_board)
  assert int(f.done.sum()) == 0 and float(f.ret.abs().sum()) == 0.0
  for t in range(8):
    obs, reward, discount = game.play(torch.from_numpy(actions[t]))
    assert torch.equal(obs.layered_board, out['obs'][t]), t
    assert torch.equal(obs.board, out['board'][t]), t
    assert _same(reward.cpu().numpy(), want['reward'][t])
    assert _same(discount.cpu().numpy(), want['discount'][t])
  # keep_obs=False: only the last frame, in play()'s buffers
  last = game.rollout(torch.from_numpy(actions), keep_obs=False, want_board=True, reset_first=True)
  assert last['obs'].shape == (B, len(traced.chars), rows, cols)
  assert torch.equal(last['obs'], out['obs'][-1]) and torch.equal(last['board'], out['board'][-1])
  assert _same(last['reward'].cpu().numpy(), want['reward'])


@pytest.mark.gpu
@pytest.mark.parametrize('name,B', [('vault', 65536), ('vault', 1001), ('burrow', 32768)])
def test_many_tracked_values_on_a_big_board_through_the_wide_kernels(name, B):
  """Four things that show (three come and go) / a changing z-order on more than 128 cells:
  K trace planes, the render kernel with up to eight patches per row."""
  from oracle.table_replay import StateWalker
  from campx_amd import wide
  build = BIG_GAMES[name]
  game = build(batch=B, device='cuda')
  first, _, _ = game.its_showtime()
  f = game.fused
  assert isinstance(f, wide.WideGame) and f.n_dyn == len(f.traced.movers)
  walker = StateWalker(f.traced, B)
  board0, layered0 = walker.render(walker.state[:4])
  assert np.array_equal(first.board[:4].cpu().numpy(), board0)
  assert np.array_equal(first.layered_board[:4].cpu().numpy(), layered0)
  T = 120
  rng = np.random.RandomState(B)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  if name == 'vault':     # environment 0: to the key, through the door, up to the gem
    actions[:, 0] = np.resize([1] * 5 + [3] * 5 + [1] * 3 + [2] * 5 + [1] * 5, T)
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  want = _check_rollout(game, f.traced, actions, out, walker)
  assert (want['shows'] == 0).any() and want['done'].sum() >= 1
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1
  # the user's classes themselves, on the generic tier, for two environments
  onehot = tabulate.default_actions()
  for env in (0, B - 1):
    g = build()
    g.its_showtime()
    for t in range(T):
      if g.game_over:
        g = build()
        g.its_showtime()
      obs, reward, _ = g.play(onehot[int(actions[t, env])])
      assert np.array_equal(out['board'][t, env].cpu().numpy(), obs.board.numpy().astype(np.int8)), (env, t)
      assert _same(out['reward'][t, env].cpu().numpy(), np.float32(float(reward)))
  # play() frame by frame after reset()
  f.reset()
  for t in range(10):
    obs, reward, discount = game.play(torch.from_numpy(actions[t]))
    assert torch.equal(obs.layered_board, out['obs'][t]), t
    assert _same(reward.cpu().numpy(), want['reward'][t])


@pytest.mark.gpu
def test_a_small_game_with_too_many_tracked_values_for_the_cell_tables_takes_the_state_table():
  """tests/traced_games.py vault plus a z-order change would be five tracked values; here:
  burrow's mole AND three vanishing coins on a 6x9 board (54 cells: the one-cell tier's size) -
  four movers + the z-order mode = five tracked values, so the engine picks the wide tier."""
  from oracle.table_replay import StateWalker
  from campx_amd import wide
  art = ['#########',
         '#A ===  #',
         '# d===u #',
         '# 1 2 3 #',
         '#      $#',
         '#########']

  class Coin(traced_games.things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      if (self.curtain * all_things['A'].curtain).sum():
        self.curtain.zero_()
        the_plot.add_reward(2.0)

  def build(**where):
    return traced_games.ascii_art_to_game(
        art, what_lies_beneath=' ',
        drapes={'A': traced_games.Mole, '#': traced_games.things.FixedDrape,
                '=': traced_games.things.FixedDrape, 'd': traced_games.things.FixedDrape,
                'u': traced_games.things.FixedDrape, '$': traced_games.things.FixedDrape,
                '1': Coin, '2': Coin, '3': Coin},
        z_order='du$=123A#', update_schedule='A123#=du$', **where)

  B, T = 4096, 150
  game = build(batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  assert isinstance(f, wide.WideGame) and f.traced.movers == ['A', '1', '2', '3']
  assert 'tracked values' in f.traced.dense_reason
  walker = StateWalker(f.traced, B)
  rng = np.random.RandomState(4)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  want = _check_rollout(game, f.traced, actions, out, walker)
  assert (want['reward'] > 1).sum() > B // 4           # coins collected


def _coin_field(**where):
  """Six things that show: a walker and five coins that vanish when collected (each +2; the
  fifth ends the episode).  6x9 board - small, but six tracked values are two more than the
  cell-indexed tables take."""
  art = ['#########',
         '#A 1 2  #',
         '#  ##   #',
         '# 3  4 5#',
         '#########']

  class Coin(traced_games.things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      if (self.curtain * all_things['A'].curtain).sum():
        self.curtain.zero_()
        the_plot.add_reward(2.0)
        if not any(all_things[c].curtain.sum() for c in '12345' if c != self.character):
          the_plot.terminate_episode()

  return traced_games.ascii_art_to_game(
      art, what_lies_beneath=' ',
      drapes={'A': traced_games.Walker, '#': traced_games.things.FixedDrape,
              '1': Coin, '2': Coin, '3': Coin, '4': Coin, '5': Coin},
      z_order='12345A#', update_schedule='A12345#', **where)


def test_six_things_that_show_tabulate_to_a_state_table():
  traced = tabulate.trace(_coin_field())
  assert traced.movers == ['A', '1', '2', '3', '4', '5']
  assert 'more than 4 tracked values' in traced.dense_reason
  spec, arrays = tabulate.to_wide_spec(traced)
  assert spec.n_dyn == 6 and arrays['state_cells'].shape == (traced.n_states, 6)
  from campx_amd import _hip
  assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == 0
  assert traced.st_done.sum() > 0                     # the last coin ends the episode


@pytest.mark.gpu
def test_six_things_that_show_through_the_wide_kernels():
  """Five to eight things take the render kernel's run-time-K instantiation and uint4 trace
  entries per state: against the state walker and, for two environments, the classes
  themselves on the generic tier."""
  from oracle.table_replay import StateWalker
  B, T = 20001, 160
  game = _coin_field(batch=B, device='cuda')
  first, _, _ = game.its_showtime()
  f = game.fused
  assert f.n_dyn == 6
  walker = StateWalker(f.traced, B)
  board0, layered0 = walker.render(walker.state[:3])
  assert np.array_equal(first.board[:3].cpu().numpy(), board0)
  rng = np.random.RandomState(21)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  # environment 0 collects everything: right along the top, down the right side, back left
  actions[:, 0] = np.resize([1] * 6 + [3] * 2 + [0] * 6 + [2] * 2, T)
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  want = _check_rollout(game, f.traced, actions, out, walker)
  assert out['trace'].shape == (6, T, B)
  assert (want['reward'] == 2.0).sum() > B and want['done'][:, 0].sum() >= 1
  onehot = tabulate.default_actions()
  for env in (0, B - 1):
    g = _coin_field()
    g.its_showtime()
    for t in range(T):
      if g.game_over:
        g = _coin_field()
        g.its_showtime()
      obs, reward, _ = g.play(onehot[int(actions[t, env])])
      assert np.array_equal(out['board'][t, env].cpu().numpy(), obs.board.numpy().astype(np.int8)), (env, t)
      got = np.float32(np.nan) if reward is None else np.float32(float(reward))
      assert _same(out['reward'][t, env].cpu().numpy(), got)
  f.reset()
  for t in range(12):
    obs, reward, discount = game.play(torch.from_numpy(actions[t]))
    assert torch.equal(obs.layered_board, out['obs'][t]), t
    assert torch.equal(obs.board, out['board'][t]), t


@pytest.mark.gpu
def test_state_tables_too_large_for_lds_are_read_through_the_caches():
  """Games with thousands of states keep their table in global memory (wide_update_kernel<false>);
  the library setting wide_lds_max=0 sends a small game down that path: same bytes."""
  game = _big_vault(batch=5000, device='cuda')
  game.its_showtime()
  rng = np.random.RandomState(9)
  acts = rng.randint(0, 5, size=(90, 5000)).astype(np.int8)
  acts[:, 0] = np.resize([1] * 5 + [3] * 5 + [1] * 3 + [2] * 5 + [1] * 5, 90)   # key, door, gem
  actions = torch.from_numpy(acts)
  ref = game.rollout(actions, reset_first=True, want_board=True)
  from campx_amd import _hip
  with _hip.config(wide_lds_max=0):
    out = game.rollout(actions, reset_first=True, want_board=True)
  for k in ('obs', 'board', 'trace', 'done'):
    assert torch.equal(out[k], ref[k]), k
  for k in ('reward', 'discount'):
    assert _same(out[k].cpu().numpy(), ref[k].cpu().numpy()), k
  assert ref['done'][:, 0].sum() > 0 and (ref['reward'] > 0).sum() > 100


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['maze 15x17 (rows of 1 530 bytes)', 'coin field (rows of 300 bytes, pieces)',
                                   'maze 16x16 (whole chunks)', 'seven coins on 4x9 (rows of 180 bytes)',
                                   'coins and a floor that turns (rows of 420 bytes, 13 variants)',
                                   'tide (rows of 288 bytes: whole chunks, two variants)'])
@pytest.mark.parametrize('B', [1, 7, 1000, 4099])
def test_play_in_one_kernel_equals_the_update_and_render_pair(which, B):
  """Engine.play() of a state-table game is ONE kernel: wide_step_kernel when its rows are whole
  16-byte chunks (the 16x16 maze), and since round 6 wide_step_lds_kernel for the others - rows of
  300, 180 and 420 bytes, a scenery of pieces or in variants - while a wave's span fits its LDS window
  (the 15x17 maze's rows of 1 530 bytes do not: the pair).  The setting wide_step=0 sends the same calls through the
  update + render pair: same bytes, frame by frame, int8 and 16-bit observations, boards, scalars,
  carried state."""
  import sys
  from campx_amd import _hip
  from campx_amd.games import maze
  from conftest import REPO
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  if which.startswith('maze 15x17'):
    build = lambda: maze.build(15, 17, batch=B, device='cuda')
  elif which.startswith('maze 16x16'):
    build = lambda: maze.build(16, 16, batch=B, device='cuda')
  elif which.startswith('coin field'):
    import coins_batched
    build = lambda: coins_batched.make_game(floor=False, batch=B, device='cuda')
  elif which.startswith('coins and a floor'):
    import coins_batched
    build = lambda: coins_batched.make_game(batch=B, device='cuda')
  elif which.startswith('tide'):
    import random_pickups
    build = lambda: random_pickups.builder(random_pickups.definitions()[14])(batch=B, device='cuda')
  else:
    import random_pickups
    build = lambda: random_pickups.builder(random_pickups.definitions()[3])(batch=B, device='cuda')
  one, two = build(), build()
  one.its_showtime()
  two.its_showtime()
  gen = torch.Generator(device='cuda').manual_seed(11)
  for t in range(40):
    if t == 20:                   # ... and as 16-bit observations from here on
      one.fused.set_play_obs_dtype(torch.bfloat16)
      two.fused.set_play_obs_dtype(torch.bfloat16)
    ids = torch.randint(0, 5, (B,), generator=gen, device='cuda', dtype=torch.int8)
    oa, ra, da = one.play(ids)
    with _hip.config(wide_step=0):
      ob, rb, db = two.play(ids)
    assert torch.equal(oa.layered_board, ob.layered_board), (which, B, t)
    assert torch.equal(oa.board, ob.board), (which, B, t)
    assert _same(ra.cpu().numpy(), rb.cpu().numpy()) and _same(da.cpu().numpy(), db.cpu().numpy()), (which, B, t)
    for name in ('state', 'done', 'ret', '_step_trace'):
      assert torch.equal(getattr(one.fused, name), getattr(two.fused, name)), (which, B, t, name)


@pytest.mark.gpu
@pytest.mark.parametrize('rows,cols,B', [(16, 16, 8192), (15, 17, 2001), (32, 32, 1024)])
def test_mazes_against_the_c_oracle(rows, cols, B):
  """oracle/campx_oracle.c - the literal restatement of the reference engine (full curtains,
  cyclic shifts, z-order paint; pinned on maze_*.npz by test_oracle_golden.py) - runs the
  maze's RULES; the HIP path walks the table tabulated from the Python classes.  Every byte
  of every frame."""
  from oracle import cpu as oracle_cpu
  T = 100
  game = maze.build(rows, cols, batch=B, device='cuda')
  first, _, _ = game.its_showtime()
  og = oracle_cpu.OracleGame.from_description(gamespec.describe(maze.build(rows, cols)))
  obs0, board0 = og.first_frame()
  assert np.array_equal(first.layered_board[B - 1].cpu().numpy(), obs0)
  assert np.array_equal(first.board[B - 1].cpu().numpy(), board0)
  rng = np.random.RandomState(rows + cols)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  gold = _golden(16, 16) if (rows, cols) == (16, 16) else None
  if gold is not None:      # environments 0 and 1 walk to the goal as in the golden
    actions[:, :2] = gold['actions'][:T, :2]
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  want = og.rollout(actions, reset_first=True)
  assert np.array_equal(out['obs'].cpu().numpy(), want['obs'])
  assert np.array_equal(out['board'].cpu().numpy(), want['board'])
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), want[k]), k
  if gold is not None:
    assert want['done'][:, :2].sum() == 2


@pytest.mark.gpu
@pytest.mark.parametrize('rows,cols', MAZES)
def test_reference_engine_goldens_on_the_gpu(rows, cols):
  gold = _golden(rows, cols)
  T, N = gold['actions'].shape
  game = maze.build(rows, cols, batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
  assert np.array_equal(first.board.cpu().numpy(), gold['board'][0])
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert np.array_equal(out['board'].cpu().numpy(), gold['board'][1:])
  assert np.array_equal(out['obs'].cpu().numpy(), gold['layered'][1:].astype(np.int8))
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), gold[k]), k
  game = maze.build(rows, cols, batch=N, device='cuda')
  game.its_showtime()
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert np.array_equal(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert np.array_equal(obs.layered_board.cpu().numpy(), gold['layered'][t + 1].astype(np.int8))
    assert _same(reward.cpu().numpy(), gold['reward'][t])
    assert _same(discount.cpu().numpy(), gold['discount'][t])
    assert np.array_equal(game.fused.done.cpu().numpy(), gold['done'][t])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('rows,cols,B', [(16, 16, 4096), (15, 17, 333)])
def test_sixteen_bit_observations(rows, cols, B, dtype):
  game = maze.build(rows, cols, batch=B, device='cuda')
  game.its_showtime()
  rng = np.random.RandomState(5)
  actions = torch.from_numpy(rng.randint(0, 5, size=(40, B)).astype(np.int8))
  ref = game.rollout(actions, reset_first=True)
  out = game.rollout(actions, reset_first=True, obs_dtype=dtype)
  assert out['obs'].dtype == dtype
  assert torch.equal(out['obs'].to(torch.int8), ref['obs'])
  assert _same(out['reward'].cpu().numpy(), ref['reward'].cpu().numpy())
  game.fused.reset()
  game.fused.set_play_obs_dtype(dtype)
  for t in range(5):
    obs, _, _ = game.play(actions[t])
    assert obs.layered_board.dtype == dtype
    assert torch.equal(obs.layered_board.to(torch.int8), ref['obs'][t]), t
  game.fused.reset()
  assert game.fused._obs.dtype == dtype


@pytest.mark.gpu
def test_a_user_class_with_hidden_tiles_discounts_and_bad_actions_on_a_wide_board():
  """Arbitrary Python (tests/traced_games.py TollWalker: custom discounts, terminate(0.75)) on a
  12x20 board, under a roof that hides the walker on some tiles; ids outside 0..4 act as stay
  and are reported."""
  from oracle.table_replay import StateWalker
  art = ['#' * 20] + ['#' + ' ' * 18 + '#' for _ in range(10)] + ['#' * 20]
  art[1] = '#A  $   ====   %   #'
  art[5] = '#   ====   $     E #'
  art[8] = '#  %     ===    $  #'

  def build(**where):
    return traced_games.ascii_art_to_game(
        art, what_lies_beneath=' ',
        drapes={'A': traced_games.TollWalker, '#': traced_games.things.FixedDrape,
                '$': traced_games.things.FixedDrape, '%': traced_games.things.FixedDrape,
                'E': traced_games.things.FixedDrape, '=': traced_games.things.FixedDrape},
        z_order='$%EA=#', update_schedule='A#$%E=', **where)

  B, T = 2048, 120
  game = build(batch=B, device='cuda')
  game.its_showtime()
  traced = game.fused.traced
  assert traced.discount_list == [1.0, 0.5, 0.25, 0.75]
  assert (traced.st_shows[:, 0] == 0).any()                 # under the roof
  walker = StateWalker(traced, B)
  rng = np.random.RandomState(77)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  actions[:, 0] = np.resize([1] * 16 + [3] * 4, T)          # along the top row, down to E
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  want = _check_rollout(game, traced, actions, out, walker)
  assert set(np.unique(want['discount']).tolist()) >= {0.25, 0.5, 0.75, 1.0}
  assert want['done'][:, 0].sum() >= 1 and (want['shows'][0] == 0).any()
  # bad ids
  game.fused.validate_actions = 'sync'
  bad = actions[:3].copy()
  bad[1, 5], bad[2, 9] = 7, -3
  with pytest.raises(ValueError, match='2 action ids'):
    game.rollout(torch.from_numpy(bad))


@pytest.mark.gpu
def test_raw_c_abi_through_ctypes_on_a_side_stream():
  """campx_wide_* called as a C program would (no torch ops), on a non-default stream."""
  from campx_amd import _hip
  from oracle.table_replay import StateWalker
  traced = _traced(16, 16)
  spec, arrays = tabulate.to_wide_spec(traced)
  lib, dev = _hip.lib, torch.device('cuda', 0)
  B, T, L, HW = 1500, 33, 6, 256
  stream = torch.cuda.Stream(dev)
  sp = ctypes.c_void_p(stream.cuda_stream)
  tables = torch.empty((lib.campx_wide_tables_bytes(ctypes.byref(spec)),), dtype=torch.uint8, device=dev)
  _hip.check(lib.campx_wide_tables_build(ctypes.byref(spec), ctypes.c_void_p(tables.data_ptr()), sp), 'build')
  pos = torch.full((B,), 7, dtype=torch.int32, device=dev)      # the state indices
  done = torch.ones((B,), dtype=torch.uint8, device=dev)
  ret = torch.full((B,), 5.0, device=dev)
  obs = torch.zeros((T, B, L, 16, 16), dtype=torch.int8, device=dev)
  reward = torch.zeros((T, B), device=dev)
  trace = torch.zeros((T, B), dtype=torch.int16, device=dev)
  rng = np.random.RandomState(3)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  acts = torch.from_numpy(actions).to(dev)
  torch.cuda.synchronize()
  state = _hip.CampxState(pos.data_ptr(), done.data_ptr(), ret.data_ptr(), None)
  out = _hip.CampxOutputs()
  out.obs, out.obs_t_stride = obs.data_ptr(), B * L * HW
  out.trace = trace.data_ptr()
  _hip.check(lib.campx_wide_reset_launch(ctypes.byref(spec), ctypes.c_void_p(tables.data_ptr()), state,
                                         out, B, sp), 'reset')
  out.reward = reward.data_ptr()
  _hip.check(lib.campx_wide_rollout_launch(ctypes.byref(spec), ctypes.c_void_p(tables.data_ptr()), state,
                                           ctypes.c_void_p(acts.data_ptr()), out, B, T, 0, sp), 'rollout')
  stream.synchronize()
  walker = StateWalker(traced, B)
  want = walker.rollout(actions)
  assert _same(reward.cpu().numpy(), want['reward'])
  assert _same(ret.cpu().numpy(), walker.ret) and int(done.sum()) == int(walker.over.sum())
  assert np.array_equal(pos.cpu().numpy(), walker.state)
  for t in (0, T - 1):
    _, layered = walker.render(want['state'][t])
    assert np.array_equal(obs[t].cpu().numpy(), layered)
  # frames that are neither back to back nor "last only" are refused, as is a missing trace
  out.obs_t_stride = B * L * HW + 16
  assert lib.campx_wide_rollout_launch(ctypes.byref(spec), ctypes.c_void_p(tables.data_ptr()), state,
                                       ctypes.c_void_p(acts.data_ptr()), out, B, T, 0, sp) == -1
  out.obs_t_stride, out.trace = B * L * HW, None
  assert lib.campx_wide_rollout_launch(ctypes.byref(spec), ctypes.c_void_p(tables.data_ptr()), state,
                                       ctypes.c_void_p(acts.data_ptr()), out, B, T, 0, sp) == -1


@pytest.mark.gpu
def test_wide_op_passes_opcheck_and_is_capturable_in_a_hip_graph():
  """campx::wide_rollout: schema / fake-tensor / functionalization checks (torch.library.opcheck),
  and a policy-in-the-loop stretch of play() calls captured in a HIP graph and replayed."""
  game = maze.build(16, 16, batch=128, device='cuda')
  game.its_showtime()
  f = game.fused
  acts = torch.randint(0, 5, (128,), dtype=torch.int8, device='cuda')
  args = (f._spec_host, f._tables, f.state, f.done, f.ret, acts, f._obs, f._board, f._reward,
          f._discount, f._step_done, None, f._step_trace, f._bad, None, False)
  torch.library.opcheck(torch.ops.campx.wide_rollout.default, args)
  acts = torch.randint(0, 5, (5, 128), dtype=torch.int8, device='cuda')
  b = f.rollout_buffers(5, want_board=True)
  args = (f._spec_host, f._tables, f.state, f.done, f.ret, acts, b['obs'], b['board'], b['reward'],
          b['discount'], b['done'], None, b['trace'], f._bad, None, True)
  torch.library.opcheck(torch.ops.campx.wide_rollout.default, args)

  game_g, game_e = (maze.build(16, 16, batch=1024, device='cuda') for _ in range(2))
  for g in (game_g, game_e):
    g.its_showtime()
    g.fused.validate_actions = False
  w = torch.randn(6 * 256, 5, device='cuda')

  def frames(g, n):
    for _ in range(n):
      ids = (g.fused._obs.view(g.fused.batch, -1).float() @ w).argmax(dim=1).to(torch.int8)
      g.play(ids)

  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    frames(game_g, 2)
  torch.cuda.current_stream().wait_stream(side)
  frames(game_e, 2)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    frames(game_g, 5)
  for _ in range(3):
    graph.replay()
  frames(game_e, 15)
  torch.cuda.synchronize()
  assert torch.equal(game_g.fused._obs, game_e.fused._obs)
  assert torch.equal(game_g.fused.state, game_e.fused.state)
  assert torch.equal(game_g.fused.ret, game_e.fused.ret)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(traced_games.GAMES))
def test_the_two_table_forms_of_a_game_give_the_same_frames(name, monkeypatch):
  """Every test-local game through BOTH tiers: its cell-indexed tables on the one-cell tier's
  kernels (FusedGame) and its state table on the wide tier's (WideGame; forced by allowing no
  dense table at all) - the same observations, boards, scalars, frame by frame."""
  from campx_amd import fused, wide
  build = traced_games.GAMES[name]
  B, T = 3001, 90
  rng = np.random.RandomState(len(name))
  actions = torch.from_numpy(rng.randint(0, 5, size=(T, B)).astype(np.int8))
  tabulate._CACHE.clear()
  game_a = build(batch=B, device='cuda')
  first_a, _, _ = game_a.its_showtime()
  assert type(game_a.fused) is fused.FusedGame
  monkeypatch.setattr(tabulate, 'DENSE_MAX_ENTRIES', 0)
  tabulate._CACHE.clear()
  game_b = build(batch=B, device='cuda')
  first_b, _, _ = game_b.its_showtime()
  tabulate._CACHE.clear()
  assert type(game_b.fused) is wide.WideGame
  assert torch.equal(first_a.layered_board, first_b.layered_board)
  assert torch.equal(first_a.board, first_b.board)
  a = game_a.rollout(actions, want_board=True)
  b = game_b.rollout(actions, want_board=True)
  for k in ('obs', 'board', 'done'):
    assert torch.equal(a[k], b[k]), k
  for k in ('reward', 'discount'):
    assert _same(a[k].cpu().numpy(), b[k].cpu().numpy()), k
  assert _same(game_a.fused.ret.cpu().numpy(), game_b.fused.ret.cpu().numpy())
  for t in range(6):
    oa, ra, da = game_a.play(actions[t])
    ob, rb, db = game_b.play(actions[t])
    assert torch.equal(oa.layered_board, ob.layered_board) and torch.equal(oa.board, ob.board), t
    assert _same(ra.cpu().numpy(), rb.cpu().numpy()) and _same(da.cpu().numpy(), db.cpu().numpy())


def _maze_with_quadrants(**where):
  """The 16x16 maze with a hidden performance: progress round the four quadrants, clockwise
  (`Engine.set_hidden_performance`, the boat race's measure - examples/boat_race.py:117-151)."""
  game = maze.build(16, 16, **where)
  q = torch.zeros((4, 16, 16), dtype=torch.uint8)
  q[0, :8, :8] = 1
  q[1, :8, 8:] = 1
  q[2, 8:, 8:] = 1
  q[3, 8:, :8] = 1
  game.set_hidden_performance('A', list(q))
  return game, q.numpy()


def test_hidden_performance_on_a_wide_board_is_in_the_state_table():
  game, q = _maze_with_quadrants()
  traced = tabulate.trace(game)
  assert traced.has_perf and set(np.unique(traced.st_perf).tolist()) == {-1, 0, 1}
  cls = (q * np.arange(1, 5)[:, None, None]).sum(0).reshape(-1)        # quadrant 1..4 per cell
  for s in range(traced.n_states):
    for a in range(5):
      if not traced.st_reached[s, a]:
        continue
      x, y = cls[traced.st_cells[s, 0]], cls[traced.st_cells[traced.st_next[s, a], 0]]
      want = 1 if y == x % 4 + 1 else (-1 if x == y % 4 + 1 else 0)
      assert traced.st_perf[s, a] == want, (s, a)


@pytest.mark.gpu
def test_hidden_performance_on_a_wide_board_through_the_kernels():
  from oracle.table_replay import StateWalker
  B, T = 9000, 110
  game, _ = _maze_with_quadrants(batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  assert f.has_perf
  walker = StateWalker(f.traced, B)
  rng = np.random.RandomState(8)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  actions[:, 0] = np.resize([1] * 9 + [3] * 9 + [0] * 9 + [2] * 9, T)     # round the quadrants
  out = game.rollout(torch.from_numpy(actions))
  want = walker.rollout(actions)
  assert out['perf'] is not None
  assert np.array_equal(out['perf'].cpu().numpy(), want['perf'])
  assert (want['perf'] == 1).sum() > 0 and (want['perf'] == -1).sum() > 0
  assert np.array_equal(out['obs'][-1].cpu().numpy(), walker.render(want['state'][-1])[1])
  f.reset()
  for t in range(10):
    game.play(torch.from_numpy(actions[t]))
    assert np.array_equal(f.perf.cpu().numpy(), want['perf'][t]), t


@pytest.mark.gpu
def test_a_launch_of_more_than_65520_frames_is_rendered_in_pieces(monkeypatch):
  """A render launch has one grid row per frame (at most 65 535): a 70 000-frame rollout of a
  small game on the wide tier goes out in two render launches over one update pass."""
  from oracle.table_replay import StateWalker
  from campx_amd import wide
  monkeypatch.setattr(tabulate, 'DENSE_MAX_ENTRIES', 0)
  tabulate._CACHE.clear()
  B, T = 16, 70000
  game = traced_games.toll_road(batch=B, device='cuda')
  game.its_showtime()
  tabulate._CACHE.clear()
  f = game.fused
  assert isinstance(f, wide.WideGame)
  rng = np.random.RandomState(2)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions))
  walker = StateWalker(f.traced, B)
  want = walker.rollout(actions)
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), want[k]), k
  for t in (0, 65519, 65520, 65521, T - 1):
    assert np.array_equal(out['obs'][t].cpu().numpy(), walker.render(want['state'][t])[1]), t
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1
