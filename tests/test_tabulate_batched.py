"""User-written games tabulated MANY STATES PER CALL (campx_amd/lanes.py, tabulate_batched.py):
the game's own `update()` methods - arithmetic on [H, W] tensors, as the reference writes its
classes (examples/boat_race.py:40-57) - run on lane tensors, a whole frontier of the state graph
per frame.

CPU: `Lanes` keeps single-tensor semantics (values AND dtypes) for the operations such classes are
made of; the lane walker and the one-frame-per-play walker produce the SAME TracedGame, field for
field, on the library's boat race / wall world and a test-local two-crate warehouse
(tests/lanes_games.py), and on the reference's own examples/boat_race.py imported in place (against
the committed reference-made table); a 12x12 warehouse of 592 588 states - ten times what
MAX_PLAYS allows the old walker - tabulates in seconds and its table predicts the generic tier
frame by frame; games with Python branches on tensors, sprites or numpy fall back, and say why.
GPU: the warehouse at B = 65 536 through `Engine(batch=...)`, against the user's classes on the
generic tier.
"""

import os
import random
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from campx_amd import lanes, tabulate, tabulate_batched
from campx_amd.games import boat_race, wall_world
from conftest import GOLDEN_DIR, REPO
import lanes_games
import lanes_probes
import traced_games


def _first_difference(a, b):
  for k, v in vars(a).items():
    if k in ('n_plays', 'batched_frames'):
      continue
    w = getattr(b, k)
    if isinstance(v, np.ndarray):
      same = (np.array_equal(v.view(np.uint32), w.view(np.uint32)) if v.dtype == np.float32
              else np.array_equal(v, w))
    else:
      same = repr(v) == repr(w)
    if not same:
      return k
  return None


# ------------------------------------------------------------------------------ Lanes

def _scalar_and_lanes(fn, inputs, n=3):
  """fn on each of n plain input sets, and once on their lanes."""
  singles = [fn(*[x[i] for x in inputs]) for i in range(n)]
  batched = fn(*[lanes.wrap(torch.stack(list(x))) for x in inputs])
  return singles, batched


def test_lanes_keep_values_and_dtypes_of_the_single_tensor_operations():
  g = torch.Generator().manual_seed(1)
  boards = [(torch.rand(5, 7, generator=g) < 0.3).to(torch.uint8) for _ in range(3)]
  walls = torch.zeros(5, 7, dtype=torch.uint8)
  walls[:, 0] = 1
  act = torch.tensor([0, 0, 1, 0, 0]).float().byte()
  dct = torch.tensor([0., 0., 3., 1., 0.])

  def agent(b):
    left = torch.cat([b[:, 1:], b[:, :1]], dim=1)
    up = torch.cat([b[1:], b[:1]], dim=0)
    cand = (act[0] * left) + (act[2] * up) + (act[4] * b)
    gate = (cand * (1 - walls)).sum()                    # int64 0-d
    new = (gate * cand) + (b * (1 - gate))               # stays uint8: a 0-d operand does not widen
    reward = -0.25
    reward += (new * b).sum() * (dct * act.float()).sum()
    return new, gate, reward, new >= 1, (new.float() * 0.5).sum(), new.long().sum(), b[1:3, 2], -new.float()

  singles, batched = _scalar_and_lanes(agent, [boards])
  for j, out in enumerate(batched):
    assert isinstance(out, lanes.Lanes), j
    p = lanes.plain(out)
    for i in range(3):
      want = singles[i][j]
      assert p[i].dtype == want.dtype, (j, p.dtype, want.dtype)
      assert tuple(p[i].shape) == tuple(want.shape) and torch.equal(p[i], want), j
  # what a tensor says about itself is the single tensor's answer
  b = lanes.wrap(torch.stack(boards))
  assert b.shape == (5, 7) and b.dim() == 2 and len(b) == 5 and b.size(1) == 7 and b.numel() == 35
  assert b.dtype == torch.uint8 and b.sum().shape == () and [r.shape for r in b][0] == (7,)


def test_lanes_writes_go_to_the_lanes_own_storage():
  base = torch.arange(24, dtype=torch.uint8).reshape(2, 3, 4)
  b = lanes.wrap(base.clone())
  alias = b
  b.set_(lanes.wrap(base + 1))                            # rebinding, as curtain.set_(new) does
  assert alias is b and torch.equal(lanes.plain(b), base + 1)
  b.set_(torch.zeros(3, 4, dtype=torch.uint8))            # one tensor for every lane
  assert lanes.plain(b).shape == (2, 3, 4) and int(lanes.plain(b).sum()) == 0
  b += 2
  b[0, 1] = 9
  b[1] = lanes.wrap(torch.tensor([[5, 5, 5, 5], [6, 6, 6, 6]], dtype=torch.uint8))
  assert lanes.plain(b)[:, 0, 1].tolist() == [9, 9] and lanes.plain(b)[1, 1].tolist() == [6] * 4
  b.mul_(2)
  b.copy_(lanes.wrap(base))
  b.zero_()
  assert int(lanes.plain(b).sum()) == 0
  shared = torch.zeros(3, 4)
  with pytest.raises(lanes.CannotBatch, match='every state shares'):
    shared += lanes.wrap(torch.ones(2, 3, 4))
  with pytest.raises(lanes.CannotBatch, match='every state shares'):
    shared[0] = lanes.wrap(torch.ones(2, 4))


def test_python_level_reads_need_every_lane_to_agree():
  same = lanes.wrap(torch.tensor([3, 3, 3]))
  assert int(same) == 3 and bool(same) and same.item() == 3 and float(same) == 3.0
  if same == 3:                                           # a branch every state takes
    pass
  differs = lanes.wrap(torch.tensor([0, 1, 1]))
  for read in (bool, int, float, lambda x: x.item()):
    with pytest.raises(lanes.CannotBatch, match='differs between states'):
      read(differs)
  with pytest.raises(lanes.CannotBatch, match='differs between states'):
    assert differs == 1
  # the whole tensor as a numpy / Python value: every lane's where they agree (a read-only copy), a
  # `Diverged` that groups the lanes by their contents where they do not
  assert same.numpy().tolist() == 3 and same.tolist() == 3
  board = lanes.wrap(torch.tensor([[1, 0], [1, 0], [1, 0]]))
  assert board.numpy().tolist() == [1, 0] and not board.numpy().flags.writeable
  with pytest.raises(lanes.Diverged, match=r'numpy\(\) of a tensor that differs between states') as split:
    lanes.wrap(torch.tensor([[0, 1], [1, 0], [0, 1]])).numpy()
  groups = split.value.values.tolist()
  assert groups[0] == groups[2] != groups[1]
  with pytest.raises(lanes.Diverged) as split:
    bool(differs)
  assert split.value.values.tolist() == [0, 1, 1]
  # the general path (vmap of the very function): dim arguments keep their meaning
  x = lanes.wrap(torch.arange(24.).reshape(2, 3, 4))
  assert torch.equal(lanes.plain(torch.roll(x, 1, 1)), torch.roll(lanes.plain(x), 1, 2))
  assert torch.equal(lanes.plain(x.sum(dim=0)), lanes.plain(x).sum(dim=1))
  assert torch.equal(lanes.plain(x.t()), lanes.plain(x).transpose(1, 2))


# ------------------------------------------------ Lanes against single tensors, random programs

def rand_leaf(rng, n):
  kind = rng.choice(['u8', 'u8', 'f32', 'i64', 'bool', 'scalar0d_i64', 'scalar0d_f32', 'pyint', 'pyfloat', 'plain_u8', 'plain_f32'])
  g = torch.Generator().manual_seed(rng.randrange(1 << 30))
  if kind == 'u8': return [(torch.rand(4, 5, generator=g) < 0.4).to(torch.uint8) for _ in range(n)], True
  if kind == 'f32': return [torch.randint(-3, 4, (4, 5), generator=g).float() * 0.25 for _ in range(n)], True
  if kind == 'i64': return [torch.randint(-3, 4, (4, 5), generator=g) for _ in range(n)], True
  if kind == 'bool': return [torch.rand(4, 5, generator=g) < 0.5 for _ in range(n)], True
  if kind == 'scalar0d_i64': return [torch.randint(0, 3, (), generator=g) for _ in range(n)], True
  if kind == 'scalar0d_f32': return [torch.randint(0, 3, (), generator=g).float() * 0.5 for _ in range(n)], True
  if kind == 'pyint': v = rng.randrange(-2, 3); return [v] * n, False
  if kind == 'pyfloat': v = rng.choice([-0.25, 0.5, 1.0, 2.0]); return [v] * n, False
  if kind == 'plain_u8': t = (torch.rand(4, 5, generator=g) < 0.4).to(torch.uint8); return [t] * n, False
  t = torch.randint(-2, 3, (4, 5), generator=g).float(); return [t] * n, False

BIN = [lambda a, b: a + b, lambda a, b: a - b, lambda a, b: a * b, lambda a, b: a >= b, lambda a, b: a <= b,
       lambda a, b: a == b, lambda a, b: 1 - a if not isinstance(a, (int, float)) else b, lambda a, b: b - a]
UN = [lambda a: a.sum(), lambda a: a.float(), lambda a: a.byte(), lambda a: a.long(),
      lambda a: torch.cat([a[:, 1:], a[:, :1]], dim=1) if a.dim() == 2 else a,
      lambda a: torch.cat([a[-1:], a[:-1]], dim=0) if a.dim() == 2 else a,
      lambda a: a[1:3, 2] if a.dim() == 2 else a, lambda a: a[0] if a.dim() >= 1 else a,
      lambda a: (a * 2).sum() if a.dtype != torch.bool else a.sum()]

def _run_program(seed, n=3):
  rng = random.Random(seed)
  vals = []     # list of (per-lane list, is_lanes)
  for _ in range(3): vals.append(rand_leaf(rng, n))
  prog = []
  for step in range(6):
    if rng.random() < 0.6:
      i, j = rng.randrange(len(vals)), rng.randrange(len(vals)); op = rng.randrange(len(BIN)); prog.append(('b', op, i, j))
    else:
      i = rng.randrange(len(vals)); op = rng.randrange(len(UN)); prog.append(('u', op, i))
    vals.append(None)
  def execute(leaves):
    env = list(leaves)
    for ins in prog:
      if ins[0] == 'b':
        a, b = env[ins[2]], env[ins[3]]
        env.append(BIN[ins[1]](a, b))
      else:
        a = env[ins[2]]
        if isinstance(a, (int, float)): env.append(a); continue
        env.append(UN[ins[1]](a))
    return env[3:]
  leaves = vals[:3]
  try:
    singles = [execute([lv[0][k] for lv in leaves]) for k in range(n)]
  except Exception as e:
    return 'skip'
  batched_leaves = [lanes.wrap(torch.stack(list(lv[0]))) if lv[1] else lv[0][0] for lv in leaves]
  try:
    batched = execute(batched_leaves)
  except lanes.CannotBatch as e:
    return 'cannot:' + str(e)[:60]
  for step, out in enumerate(batched):
    for k in range(n):
      want = singles[k][step]
      if isinstance(out, lanes.Lanes):
        got = lanes.plain(out)[k]
      else:
        got = out
      if isinstance(want, (int, float)):
        if got != want: return 'MISMATCH py step %d' % step
        continue
      if not torch.is_tensor(got): return 'MISMATCH type step %d: %r vs %r' % (step, type(got), type(want))
      if got.dtype != want.dtype or tuple(got.shape) != tuple(want.shape) or not torch.equal(got, want):
        return 'MISMATCH seed %d step %d prog %r: got %s %s want %s %s' % (seed, step, prog[step], got.dtype, tuple(got.shape), want.dtype, tuple(want.shape))
  return 'ok'



def test_random_programs_of_the_supported_operations_match_lane_by_lane():
  """Six-instruction programs drawn from the operations arithmetic-only game classes use (+ - *,
  comparisons, `1 - x`, cat-shifts, slices, sum, dtype conversions) over uint8 / float32 / int64 /
  bool boards, 0-d tensors, Python numbers and tensors every lane shares: values, dtypes and
  shapes of EVERY intermediate equal what the same program gives on each lane's tensors alone."""
  outcomes = [_run_program(seed) for seed in range(400)]
  bad = [o for o in outcomes if o.startswith('MISMATCH')]
  assert not bad, bad[:3]
  assert sum(o == 'ok' for o in outcomes) > 200


# ------------------------------------------------------------------------- the two walkers

def _both(build, monkeypatch, actions=None):
  monkeypatch.setenv('CAMPX_TABULATE', 'walk')
  walked = tabulate.trace(build(), actions=actions, cache=False)
  monkeypatch.setenv('CAMPX_TABULATE', 'batch')
  batched = tabulate.trace(build(), actions=actions, cache=False)
  assert tabulate.LAST_WALK[0].startswith('lanes: ')
  return walked, batched


@pytest.mark.parametrize('build', [boat_race.build, wall_world.build, lanes_games.small_warehouse],
                         ids=['boat_race', 'wall_world', 'small_warehouse'])
def test_lane_walker_and_play_walker_produce_the_same_traced_game(build, monkeypatch):
  walked, batched = _both(build, monkeypatch)
  assert walked.n_states == batched.n_states and walked.n_states > 1
  assert _first_difference(walked, batched) is None
  # ... at a fraction of the frames (and the default, 'auto', is the lane walker)
  assert batched.n_plays < walked.n_plays
  monkeypatch.delenv('CAMPX_TABULATE')
  again = tabulate.trace(build(), cache=False)
  assert tabulate.LAST_WALK[0].startswith('lanes: ') and _first_difference(batched, again) is None


@pytest.mark.skipif(not os.path.exists('/root/reference/examples/boat_race.py'),
                    reason='reference tree not present (GPU box)')
def test_the_references_own_boat_race_classes_run_on_lanes():
  """examples/boat_race.py imported in place: AgentDrape / DirectionalHoverRewardDrape exactly as
  the reference has them (`assert sum(act) == 1`, `the_plot['prev_pos_A'] = layers['A']`), on lane
  tensors - to the reference-made transition table."""
  code = r'''
import sys, os
os.environ['CAMPX_TABULATE'] = 'batch'
sys.path.insert(0, %(repo)r)
sys.path.append('/root/reference/examples')
import numpy as np
import campx, boat_race
assert campx.__file__.startswith(%(repo)r) and boat_race.__file__.startswith('/root/reference')
from campx_amd import engine, tabulate
held = engine.Engine.its_showtime
engine.Engine.its_showtime = lambda self: (None, None, None)     # (no GPU here: held back)
eng, _, _, _ = boat_race.make_game()
engine.Engine.its_showtime = held
assert type(eng.things['A']).__module__ == 'boat_race'
traced = tabulate.trace(eng, cache=False)
assert tabulate.LAST_WALK[0].startswith('lanes: '), tabulate.LAST_WALK[0]
with np.load(%(fix)r) as f:
  fix = {k: f[k] for k in f.files}
assert traced.movers == ['A'] and traced.init_cells == (int(fix['cells'][0]),)
assert sorted({traced.cells_of(i)[0] for i in np.flatnonzero(traced.reached)}) == sorted(fix['cells'].tolist())
for s, cell in enumerate(fix['cells']):
  for a in range(5):
    i = traced.index_of((int(cell),), a)
    assert traced.reached[i] and traced.next_cells[0, i] == fix['next_cell'][s, a], (cell, a)
    assert np.array_equal(np.array([traced.reward[i]]).view(np.uint32),
                          np.array([fix['reward'][s, a]], np.float32).view(np.uint32)), (cell, a)
    assert traced.done[i] == fix['done'][s, a] and traced.visible[0, i] == fix['visible'][s, a]
    board = traced.model_board((int(traced.next_cells[0, i]),))
    assert np.array_equal(board, fix['board'][s, a].astype(np.uint8))
print('ok', traced.n_states, traced.n_plays)
''' % dict(repo=REPO, fix=os.path.join(GOLDEN_DIR, 'boat_race_table.npz'))
  out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  assert out.stdout.strip().startswith('ok 8 ')


def test_a_warehouse_of_half_a_million_states_tabulates_in_seconds_and_predicts_the_generic_tier(monkeypatch):
  """a user's own two-crate sokoban, plain Drapes on 12x12: 592 588 reachable
  states (the one-frame-per-play walker stops at 60 000 frames = 12 000 states)."""
  monkeypatch.setenv('CAMPX_TABULATE', 'batch')
  t0 = time.perf_counter()
  game = tabulate.trace(lanes_games.warehouse(), cache=False)
  took = time.perf_counter() - t0
  assert game.n_states == 592588 and game.movers == ['X', 'Y', 'P'] and game.dense_reason
  # (15-20 s on eight idle cores.  What is asserted is the structure that makes it so - a few hundred
  # frames of Python, each over a whole level of the state graph -, not the wall clock of whatever else
  # the machine is doing: four pytest workers x eight torch threads once read 894 s here)
  assert game.n_plays < 600
  from conftest import took_about
  took_about(took, 90.0, 'tabulating the 592 588-state warehouse')
  # the table against the user's classes on the generic tier: three random walks
  acts = tabulate.default_actions()
  rng = np.random.RandomState(5)
  for walk in range(3):
    eng = lanes_games.warehouse()
    obs, _, _ = eng.its_showtime()
    s = 0
    assert np.array_equal(obs.board.numpy().astype(np.uint8).reshape(-1), game.st_board[0])
    for t in range(120):
      a = int(rng.choice(5, p=[.24, .24, .24, .24, .04]))
      obs, reward, discount = eng.play(acts[a].clone())
      assert game.st_reached[s, a]
      want = np.float32(game.st_reward[s, a])
      assert np.float32(float(reward)) == want, (walk, t)
      s = int(game.st_next[s, a])
      assert np.array_equal(obs.board.numpy().astype(np.uint8).reshape(-1), game.st_board[s]), (walk, t)
  # and the wide tier takes it
  spec, blob = tabulate.to_wide_spec(game)[:2] if isinstance(tabulate.to_wide_spec(game), tuple) else (tabulate.to_wide_spec(game), None)
  assert spec is not None


def test_games_the_lane_walker_does_not_take_fall_back_and_say_why(monkeypatch):
  monkeypatch.delenv('CAMPX_TABULATE', raising=False)
  for build, why in ((traced_games.burrow, r'changes the z-order'),
                     (traced_games.time_limit, r'the_plot\.frame|something besides the curtains')):
    game = tabulate.trace(build(), cache=False)
    assert game.n_states > 1
    assert tabulate.LAST_WALK[0].startswith('one frame per play (lanes: ') and \
        __import__('re').search(why, tabulate.LAST_WALK[0]), tabulate.LAST_WALK[0]
  # a Python branch on a tensor that differs between states
  from campx import things
  from campx.ascii_art import ascii_art_to_game

  class Branchy(things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      shift, axis = ((-1, 1), (1, 1), (-1, 0), (1, 0), (0, 0))[int(actions.argmax())]
      moved = torch.roll(self.curtain, shift, axis)
      if (moved * layers['#']).sum() == 0:          # differs from state to state
        self.curtain.set_(moved)

  def build():
    return ascii_art_to_game(['#####', '#B  #', '#   #', '#####'], what_lies_beneath=' ',
                             drapes={'B': Branchy, '#': things.FixedDrape}, z_order='B#', update_schedule='B#')
  # (round 5, late: such a frame is run again for each group of states that branch the same way -
  # the lane walker takes the game, and gives what the one-frame walker gives)
  game = tabulate.trace(build(), cache=False)
  assert game.n_states == 6 and tabulate.LAST_WALK[0].startswith('lanes: '), tabulate.LAST_WALK[0]
  monkeypatch.setenv('CAMPX_TABULATE', 'walk')
  assert _first_difference(tabulate.trace(build(), cache=False), game) is None


def _predicts_live_play(table, build, frames=60, seed=3):
  """The table against the user's classes played live on the generic tier: board, reward bits, discount."""
  rng = np.random.RandomState(seed)
  eng = build()
  obs, _, _ = eng.its_showtime()
  one_hot = torch.eye(5)
  s = 0
  assert np.array_equal(obs.board.numpy().astype(np.uint8).reshape(-1), table.st_board[0])
  for t in range(frames):
    if eng.game_over:
      break
    a = int(rng.randint(5))
    obs, reward, discount = eng.play(one_hot[a].clone())
    assert table.st_reached[s, a], (t, s, a)
    want, got = np.float32(table.st_reward[s, a]), np.float32(float('nan') if reward is None else float(reward))
    assert want.view(np.uint32) == got.view(np.uint32) or (np.isnan(want) and np.isnan(got)), (t, s, a, want, got)
    s = int(table.st_next[s, a])
    if not eng.game_over:
      assert np.array_equal(obs.board.numpy().astype(np.uint8).reshape(-1), table.st_board[s]), (t, s, a)


def test_a_partner_kept_in_an_instance_attribute_is_followed_by_the_copies(monkeypatch):
  for mode in ('walk', 'auto'):
    monkeypatch.setenv('CAMPX_TABULATE', mode)
    table = tabulate.trace(lanes_probes.spy_through_its_own_attribute(), cache=False)
    assert table.n_states == 12
    _predicts_live_play(table, lanes_probes.spy_through_its_own_attribute)


@pytest.mark.parametrize('a,b,expect', lanes_probes.CASES, ids=[c[0].__name__ for c in lanes_probes.CASES])
def test_games_at_the_edges_are_tabulated_the_same_or_handed_to_the_walk(a, b, expect, monkeypatch):
  """tests/lanes_probes.py: whatever a game class does, the default walker's table equals the
  one-frame-per-play walker's - through lanes where the operations have a lane-by-lane form,
  through the fall-back where they do not - or both refuse the game with the same message."""
  outcome = {}
  for mode in ('walk', 'auto'):
    monkeypatch.setenv('CAMPX_TABULATE', mode)
    try:
      outcome[mode] = tabulate.trace(lanes_probes.game(a, b)(), cache=False)
      if mode == 'auto':
        how = tabulate.LAST_WALK[0]
    except tabulate.TabulationError as e:
      outcome[mode] = str(e)
  if expect.startswith('REFUSED: '):
    cut = lambda text: text.split(' answered action ')[0]      # (which action showed it: the random game's luck)
    assert isinstance(outcome['walk'], str) and cut(outcome['walk']) == cut(outcome['auto'])
    assert expect[len('REFUSED: '):] in outcome['walk']
    return
  assert not isinstance(outcome['walk'], str), outcome['walk']
  assert not isinstance(outcome['auto'], str), outcome['auto']
  assert how.startswith(expect), how
  assert outcome['walk'].n_states == outcome['auto'].n_states > 1
  assert _first_difference(outcome['walk'], outcome['auto']) is None
  _predicts_live_play(outcome['auto'], lanes_probes.game(a, b))


def test_a_sprite_agent_written_with_python_ints_and_branches_is_tabulated_on_lanes(monkeypatch):
  """PyColab's usual style - the agent a Sprite whose update() is integer arithmetic and `if`s on
  what it finds, a crate that reads where the agent stands: frames are split by where the sprite
  stands and by how the branches fall; same table as the one-frame walker, and it predicts live
  play.  On a larger board: thousands of states in seconds."""
  monkeypatch.setenv('CAMPX_TABULATE', 'walk')
  walked = tabulate.trace(lanes_probes.porter()(), cache=False)
  monkeypatch.setenv('CAMPX_TABULATE', 'batch')
  batched = tabulate.trace(lanes_probes.porter()(), cache=False)
  assert tabulate.LAST_WALK[0].startswith('lanes: '), tabulate.LAST_WALK[0]
  assert walked.n_states == batched.n_states > 100 and _first_difference(walked, batched) is None
  _predicts_live_play(batched, lanes_probes.porter(), frames=200)
  big = ['############', '#P         #', '# X    #   #', '#      #   #', '#   ####   #', '#          #',
         '#      G   #', '#   #      #', '#   #      #', '############']
  t0 = time.perf_counter()
  table = tabulate.trace(lanes_probes.porter(big)(), cache=False)
  assert tabulate.LAST_WALK[0].startswith('lanes: ') and table.n_states > 3000
  from conftest import took_about
  took_about(time.perf_counter() - t0, 60.0, 'tabulating the big porter board on lanes')
  _predicts_live_play(table, lanes_probes.porter(big), frames=300, seed=9)


@pytest.mark.parametrize('build,what', lanes_probes.SPIES, ids=[b.__name__ for b, _ in lanes_probes.SPIES])
def test_a_live_game_object_reached_behind_the_engine_is_refused_by_name(build, what, monkeypatch):
  """A class that reads another thing through a module global, a closure or a default argument
  (not through `all_things`) reads an object the tabulators' deep copies do not follow: both
  walkers would tabulate the game as if that thing stood still - silently wrong tables.  Refused
  statically, naming the route; the generic tier (batch=None) still plays such a game."""
  for mode in ('walk', 'auto'):
    monkeypatch.setenv('CAMPX_TABULATE', mode)
    with pytest.raises(tabulate.TabulationError) as e:
      tabulate.trace(build(), cache=False)
    assert what in str(e.value), str(e.value)
  eng = build()
  eng.its_showtime()
  one_hot = torch.eye(5)
  rewards = [float(eng.play(one_hot[a])[1]) for a in (1, 1, 3, 3)]
  assert len(set(rewards)) > 1          # (the live reference does matter to the game)


# ------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_user_written_warehouse_runs_batched_against_its_own_classes():
  from campx_amd import wide
  B, T = 65536, 60
  game = lanes_games.warehouse(batch=B, device='cuda')
  t0 = time.perf_counter()
  game.its_showtime()
  took = time.perf_counter() - t0
  assert isinstance(game.fused, wide.WideGame) and game.fused.traced.n_states == 592588
  assert took < 180.0, took       # (15 s of tabulation in the build container; 66 s on a 256-core GPU host)
  rng = np.random.RandomState(9)
  actions = rng.choice(5, size=(T, B), p=[.24, .24, .24, .24, .04]).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  board = out['board'].cpu().numpy()
  reward = out['reward'].cpu().numpy()
  acts = tabulate.default_actions()
  for env in (0, 777, B - 1):
    single = lanes_games.warehouse()
    single.its_showtime()
    for t in range(T):
      obs, r, d = single.play(acts[int(actions[t, env])].clone())
      assert np.array_equal(board[t, env], obs.board.numpy().astype(np.int8)), (env, t)
      assert np.float32(float(r)) == reward[t, env], (env, t)
