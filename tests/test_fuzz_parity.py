"""Randomly generated games, batch sizes and episode lengths: HIP kernels vs the oracle.

Seeded, so failures reproduce.  Covers what the fixed games do not: arbitrary wall /
coin / hover-tile / box placements, boards of every aspect ratio up to 128 cells, one- to
four-mover games (with and without their state tables), random one-mover games on boards of 130 to 560 cells (the wide tier), batches that are not multiples of 4 / 16 / 64 / 256 and episode
lengths that are not multiples of the kernels' 16-frame groups and 64-frame chunks."""

import os

import numpy as np
import pytest
import torch

from campx_amd import gamespec, rules
from campx_amd.ascii_art import ascii_art_to_game, Partial
from oracle import cpu

pytestmark = pytest.mark.gpu


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def random_game(rng):
  """(builder, description of what was drawn) for a random legal game."""
  H, W = int(rng.randint(2, 12)), int(rng.randint(2, 12))
  while H * W > 128 or H * W < 4:
    H, W = int(rng.randint(2, 12)), int(rng.randint(2, 12))
  art = np.full((H, W), ' ', dtype='<U1')

  free = list(zip(*np.where(art == ' ')))
  if len(free) < 4:
    return random_game(rng)
  rng.shuffle(free)
  art[free.pop()] = 'A'
  n_boxes = int(rng.choice([0, 0, 1, 2, 3])) if len(free) > 6 else 0
  boxes = 'XYZ'[:n_boxes]
  for ch in boxes:
    art[free.pop()] = ch
  coins = rng.rand() < 0.5 and not boxes
  hover = rng.rand() < 0.5 and not boxes
  goal = bool(boxes) or rng.rand() < 0.3
  for _ in range(int(rng.randint(1, 4)) if coins else 0):
    if free: art[free.pop()] = '*'
  for _ in range(int(rng.randint(1, 3)) if hover else 0):
    if free: art[free.pop()] = '>'
  if goal and free:
    art[free.pop()] = 'G'
  else:
    goal = False
  rows = [''.join(r) for r in art]
  step_reward = None if hover else float(rng.choice([-1, 0, 0.5]))
  dctns = [float(x) for x in rng.choice([0, 1, 2, 3], size=5)]
  base = float(rng.choice([-0.25, 0, 0.5]))

  def build(batch=None, device=None):
    drapes = {'#': rules.FixedDrape,
              'A': Partial(rules.AgentDrape, blocking_chars='#' + boxes, step_reward=step_reward,
                           reward_chars='*' if coins and '*' in ''.join(rows) else '')}
    present = set(''.join(rows))
    if '*' in present:
      drapes['*'] = rules.FixedDrape
    if '>' in present:
      drapes['>'] = Partial(rules.DirectionalHoverRewardDrape,
                            dctns=torch.tensor(dctns), base_reward=base)
    if 'G' in present:
      drapes['G'] = Partial(rules.GoalDrape, agent_char='A', step_reward=-1, goal_reward=7)
    for ch in boxes:
      drapes[ch] = Partial(rules.BoxDrape, agent_char='A',
                           blocking_chars='#' + boxes.replace(ch, ''))
    back = ''.join(c for c in '*>G' if c in drapes)
    z_order = back + boxes + 'A' + ('#' if '#' in drapes else '')
    first = [list(boxes)] if boxes else []
    rest = ['A'] + [c for c in '>G*#' if c in drapes]
    return ascii_art_to_game(rows, what_lies_beneath=' ', drapes=drapes,
                             update_schedule=first + [rest], z_order=z_order,
                             batch=batch, device=device)
  return build, rows


@pytest.mark.parametrize('mode', ['split-table', 'fused-interpreter'])
@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_FUZZ_SEEDS', '12'))))
def test_random_games(seed, mode):
  from campx_amd import fused
  rng = np.random.RandomState(1000 + seed)
  build, rows = random_game(rng)
  saved = fused.SPLIT_ROLLOUT, fused.COMPILE_TABLE
  fused.SPLIT_ROLLOUT = mode.startswith('split')
  fused.COMPILE_TABLE = mode.endswith('table')
  try:
    batch = int(rng.choice([1, 5, 16, 48, 100, 257, 272, 1024]))
    game = build(batch=batch, device='cuda')
    game.its_showtime()
    og = cpu.OracleGame.from_description(gamespec.describe(build()))
    for launch, T in enumerate([int(rng.randint(1, 40)), int(rng.randint(1, 150))]):
      actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
      out = game.rollout(torch.from_numpy(actions), want_board=True)
      ref = og.rollout(actions, reset_first=(launch == 0))
      for k in ('obs', 'board', 'discount', 'done'):
        assert _same(out[k].cpu().numpy(), ref[k]), (rows, batch, T, k)
      if out['reward'] is None:
        assert np.isnan(ref['reward']).all()
      else:
        assert _same(out['reward'].cpu().numpy(), ref['reward']), (rows, batch, T)
    # and frame by frame through play()
    acts = rng.randint(0, 5, size=(5, batch)).astype(np.int8)
    ref = og.rollout(acts)
    for t in range(5):
      obs, reward, discount = game.play(torch.from_numpy(acts[t]))
      assert _same(obs.layered_board.cpu().numpy(), ref['obs'][t]), (rows, batch, t)
      assert _same(discount.cpu().numpy(), ref['discount'][t])
  finally:
    fused.SPLIT_ROLLOUT, fused.COMPILE_TABLE = saved


def _small_random_game(rng):
  """A random game whose reachable state space the host can tabulate in seconds."""
  while True:
    build, rows = random_game(rng)
    cells = len(rows) * len(rows[0])
    boxes = sum(ch in ''.join(rows) for ch in 'XYZ')
    if (boxes == 0 and cells <= 64) or (boxes == 1 and cells <= 16):
      return build, rows, boxes


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_FUZZ_TABLE_SEEDS', '30'))))
def test_device_built_tables_equal_tables_from_running_the_python_rules(seed):
  """Two independent derivations of a game's state table: the rule INTERPRETER KERNEL run over
  every (cell[, cell], action) on the device (campx_spec_compile / campx_pair_table_build),
  and campx_amd.tabulate running the rule classes' ordinary Python update() bodies on the
  generic tier over every reachable state on the host.  Every entry the game can reach must
  agree: next cells, who shows, reward, game-over."""
  from campx_amd import tabulate
  rng = np.random.RandomState(5000 + seed)
  build, rows, boxes = _small_random_game(rng)
  traced = tabulate.trace(build())
  game = build(batch=64, device='cuda')
  game.its_showtime()
  f = game.fused
  assert f.traced is None and f.uses_table          # rule classes: device-built tables
  movers = [f.chars[f.spec.dyn_layer[d]] for d in range(f.n_dyn)]
  assert movers == traced.movers, (rows, movers, traced.movers)
  idx = np.flatnonzero(traced.reached)
  assert len(idx) >= 5
  if f.n_dyn == 1:
    for i in idx:
      tr = f.spec.table[int(i)]
      assert tr.next_cell == traced.next_cells[0, i], (rows, i)
      assert (0 if tr.paint & 0x80 else 1) == traced.visible[0, i], (rows, i)
      assert (tr.done & 1) == traced.done[i] and (tr.done >> 4) == 0
      assert _same(np.float32(tr.reward), traced.reward[i]), (rows, i)
  else:
    raw = f._pair_table.cpu().numpy()
    rewards = raw[:1024].view(np.float32)
    entries = raw[1024:].view(np.uint32)
    e = entries[idx]
    assert np.array_equal(e & 0x7f, traced.next_cells[0, idx]), rows
    assert np.array_equal((e >> 7) & 0x7f, traced.next_cells[1, idx]), rows
    assert np.array_equal((e >> 14) & 1, traced.visible[0, idx]), rows
    assert np.array_equal((e >> 15) & 1, traced.visible[1, idx]), rows
    assert np.array_equal((e >> 16) & 1, traced.done[idx]), rows
    assert _same(rewards[(e >> 19) & 0xff], traced.reward[idx]), rows


def random_big_game(rng):
  """A random one-mover rule game on a board ABOVE 128 cells (the wide tier): random walls
  (the agent wraps round the edges where there is none), coin tiles, directional hover
  tiles, maybe a goal - the rule classes of `random_game`, no boxes."""
  H, W = int(rng.randint(9, 28)), int(rng.randint(9, 28))
  while not 130 <= H * W <= 560:
    H, W = int(rng.randint(9, 28)), int(rng.randint(9, 28))
  art = np.full((H, W), ' ', dtype='<U1')
  art[rng.rand(H, W) < rng.choice([0.0, 0.1, 0.25])] = '#'
  free = list(zip(*np.where(art == ' ')))
  rng.shuffle(free)
  art[free.pop()] = 'A'
  coins = rng.rand() < 0.6
  hover = rng.rand() < 0.5
  for _ in range(int(rng.randint(3, 30)) if coins else 0):
    art[free.pop()] = '*'
  for _ in range(int(rng.randint(2, 12)) if hover else 0):
    art[free.pop()] = '>'
  goal = rng.rand() < 0.5
  if goal:
    art[free.pop()] = 'G'
  rows = [''.join(r) for r in art]
  step_reward = None if hover else float(rng.choice([-1, 0, 0.5]))
  dctns = [float(x) for x in rng.choice([0, 1, 2, 3], size=5)]
  base = float(rng.choice([-0.25, 0, 0.5]))

  def build(batch=None, device=None):
    present = set(''.join(rows))
    drapes = {'A': Partial(rules.AgentDrape, blocking_chars='#' if '#' in present else '',
                           step_reward=step_reward, reward_chars='*' if '*' in present else '')}
    for ch in '#*':
      if ch in present:
        drapes[ch] = rules.FixedDrape
    if '>' in present:
      drapes['>'] = Partial(rules.DirectionalHoverRewardDrape, dctns=torch.tensor(dctns),
                            base_reward=base)
    if 'G' in present:
      drapes['G'] = Partial(rules.GoalDrape, agent_char='A', step_reward=-1, goal_reward=7)
    back = ''.join(c for c in '*>G' if c in drapes)
    return ascii_art_to_game(rows, what_lies_beneath=' ', drapes=drapes,
                             update_schedule=[['A'] + [c for c in '>G*#' if c in drapes]],
                             z_order=back + 'A' + ('#' if '#' in drapes else ''),
                             batch=batch, device=device)
  return build, rows


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_FUZZ_BIG_SEEDS', '6'))))
def test_random_big_boards(seed):
  """The wide tier (state table tabulated on the host from the rule classes' Python) against
  the C oracle (the rules restated in C): every byte of every frame, two launches and play()."""
  from campx_amd import wide
  rng = np.random.RandomState(5000 + seed)
  build, rows = random_big_game(rng)
  batch = int(rng.choice([1, 7, 48, 257, 512]))
  game = build(batch=batch, device='cuda')
  first, _, _ = game.its_showtime()
  assert isinstance(game.fused, wide.WideGame)
  og = cpu.OracleGame.from_description(gamespec.describe(build()))
  obs0, board0 = og.first_frame()
  assert np.array_equal(first.layered_board[batch - 1].cpu().numpy(), obs0), rows
  for launch, T in enumerate([int(rng.randint(1, 40)), int(rng.randint(1, 150))]):
    actions = rng.randint(0, 5, size=(T, batch)).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'board', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), ref[k]), (rows, batch, T, k)
    if out['reward'] is None:
      assert np.isnan(ref['reward']).all()
    else:
      assert _same(out['reward'].cpu().numpy(), ref['reward']), (rows, batch, T)
  acts = rng.randint(0, 5, size=(5, batch)).astype(np.int8)
  ref = og.rollout(acts)
  for t in range(5):
    obs, reward, discount = game.play(torch.from_numpy(acts[t]))
    assert _same(obs.layered_board.cpu().numpy(), ref['obs'][t]), (rows, batch, t)
    assert _same(obs.board.cpu().numpy(), ref['board'][t]), (rows, batch, t)
    assert _same(discount.cpu().numpy(), ref['discount'][t])


def random_big_warehouse(rng):
  """A sokoban level of one or two boxes on a random walled board of 132 to 240 cells: a multi-mover
  rule game above 128 cells - its states (10^4 to 10^6) are enumerated on the device by the rules
  themselves (campx_amd/enumerate_states.py), the state table runs on the wide tier."""
  H, W = int(rng.randint(9, 15)), int(rng.randint(11, 19))
  while not 132 <= H * W <= 240:
    H, W = int(rng.randint(9, 15)), int(rng.randint(11, 19))
  art = np.full((H, W), ' ', dtype='<U1')
  art[0, :] = art[-1, :] = '#'
  art[:, 0] = art[:, -1] = '#'
  inner = art[1:-1, 1:-1]
  inner[rng.rand(H - 2, W - 2) < 0.15] = '#'
  free = list(zip(*np.where(art == ' ')))
  rng.shuffle(free)
  boxes = 'XY'[:int(rng.randint(1, 3))]
  r, c = free.pop()
  art[r, c] = 'A'
  beside = [q for q in free if abs(q[0] - r) + abs(q[1] - c) == 1]
  for i, ch in enumerate(boxes):
    spot = beside.pop() if (i == 0 and beside) else free.pop()
    if spot in free:
      free.remove(spot)
    art[spot] = ch
  art[free.pop()] = 'G'
  rows = [''.join(x) for x in art]

  def build(batch=None, device=None):
    drapes = {'#': rules.FixedDrape, 'A': Partial(rules.AgentDrape, blocking_chars='#' + boxes),
              'G': Partial(rules.GoalDrape, agent_char='A', step_reward=-1, goal_reward=50)}
    for ch in boxes:
      drapes[ch] = Partial(rules.BoxDrape, agent_char='A',
                           blocking_chars='#' + ''.join(b for b in boxes if b != ch))
    return ascii_art_to_game(rows, what_lies_beneath=' ', drapes=drapes,
                             update_schedule=[list(boxes), ['A', 'G', '#']], z_order='G' + boxes + 'A#',
                             batch=batch, device=device)
  return build, rows


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_FUZZ_WAREHOUSE_SEEDS', '5'))))
def test_random_big_warehouses(seed):
  """Device-enumerated state tables of random multi-mover levels against the C oracle: actions that
  keep going (pushes), two launches with the state carried, then play()."""
  from campx_amd import wide
  rng = np.random.RandomState(8000 + seed)
  build, rows = random_big_warehouse(rng)
  batch = int(rng.choice([7, 257, 2048]))
  game = build(batch=batch, device='cuda')
  first, _, _ = game.its_showtime()
  assert isinstance(game.fused, wide.WideGame) and game.fused.n_dyn >= 2, rows
  og = cpu.OracleGame.from_description(gamespec.describe(build()))
  for launch, T in enumerate([int(rng.randint(20, 60)), int(rng.randint(1, 120))]):
    actions = np.repeat(rng.randint(0, 5, size=((T + 3) // 4, batch)), 4, axis=0)[:T].astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'board', 'reward', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), ref[k]), (rows, batch, T, k)
  moved = sum(len(set((ref['board'][:, 0] == ord(ch)).reshape(ref['board'].shape[0], -1).argmax(1).tolist())) > 1
              for ch in 'XY')
  acts = rng.randint(0, 5, size=(5, batch)).astype(np.int8)
  ref = og.rollout(acts)
  for t in range(5):
    obs, reward, discount = game.play(torch.from_numpy(acts[t]))
    assert _same(obs.layered_board.cpu().numpy(), ref['obs'][t]), (rows, batch, t)
    assert _same(reward.cpu().numpy(), ref['reward'][t]), (rows, batch, t)


def random_python_game(rng):
  """A random game of plain Python classes (tests/traced_games.py: a walker whose tiles change
  the frame's discount and end the episode, plus up to two coins that vanish when collected)
  on a random walled board of 40 to 200 cells: host tabulation, then whichever tier takes it
  (cell-indexed tables, or the state table above 128 cells)."""
  import traced_games
  H, W = int(rng.randint(5, 13)), int(rng.randint(6, 18))
  while not 40 <= H * W <= 200:
    H, W = int(rng.randint(5, 13)), int(rng.randint(6, 18))
  art = np.full((H, W), ' ', dtype='<U1')
  art[0, :] = art[-1, :] = '#'
  art[:, 0] = art[:, -1] = '#'
  inner = art[1:-1, 1:-1]
  inner[rng.rand(H - 2, W - 2) < 0.12] = '#'
  free = list(zip(*np.where(art == ' ')))
  rng.shuffle(free)
  art[free.pop()] = 'A'
  for ch, count in (('$', rng.randint(0, 4)), ('%', rng.randint(0, 3)), ('E', rng.randint(0, 2))):
    for _ in range(int(count)):
      art[free.pop()] = ch
  coins = '12'[:int(rng.randint(0, 3))]
  for ch in coins:
    art[free.pop()] = ch
  rows = [''.join(r) for r in art]

  class Coin(traced_games.things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      if (self.curtain * all_things['A'].curtain).sum():
        self.curtain.zero_()
        the_plot.add_reward(2.0)

  def build(**where):
    drapes = {'A': traced_games.TollWalker, '#': traced_games.things.FixedDrape,
              '$': traced_games.things.FixedDrape, '%': traced_games.things.FixedDrape,
              'E': traced_games.things.FixedDrape}
    for ch in coins:
      drapes[ch] = Coin
    return traced_games.ascii_art_to_game(
        rows, what_lies_beneath=' ', drapes=drapes, z_order='$%E' + coins + 'A#',
        update_schedule='A' + coins + '#$%E', **where)
  return build, rows


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_FUZZ_PYTHON_SEEDS', '6'))))
def test_random_python_class_games(seed):
  """Host-tabulated games against the classes themselves (the generic tier, three environments,
  every frame) and the state walker (every environment's scalars, sampled frames)."""
  from campx_amd import tabulate
  from oracle.table_replay import StateWalker
  rng = np.random.RandomState(7000 + seed)
  build, rows = random_python_game(rng)
  B, T = int(rng.choice([3, 64, 257, 1000])), int(rng.randint(20, 90))
  game = build(batch=B, device='cuda')
  game.its_showtime()
  traced = game.fused.traced
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  walker = StateWalker(traced, B)
  want = walker.rollout(actions)
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), want[k]), (rows, k)
  for t in (0, T // 2, T - 1):
    board, layered = walker.render(want['state'][t])
    assert np.array_equal(out['obs'][t].cpu().numpy(), layered), (rows, t)
    assert np.array_equal(out['board'][t].cpu().numpy(), board), (rows, t)
  onehot = tabulate.default_actions()
  for env in range(min(3, B)):
    g = build()
    g.its_showtime()
    for t in range(T):
      if g.game_over:
        g = build()
        g.its_showtime()
      obs, reward, discount = g.play(onehot[int(actions[t, env])])
      assert np.array_equal(out['board'][t, env].cpu().numpy(), obs.board.numpy().astype(np.int8)), (rows, env, t)
      got = np.float32(np.nan) if reward is None else np.float32(float(reward))
      assert _same(out['reward'][t, env].cpu().numpy(), got), (rows, env, t)
      assert float(out['discount'][t, env]) == float(discount)
