"""Games of arbitrary Python classes on the HIP tier: host-side tabulation
(`campx_amd.tabulate`) and, on the GPU, the table + render kernels running the result.

CPU part: the tabulator against (a) `tests/golden/boat_race_table.npz`, the boat race's
transition table as the REFERENCE's engine plays the reference's unmodified
examples/boat_race.py (`tests/golden/make_table_golden.py`), (b) this repo's generic tier
playing the same classes frame by frame, (c) games it must refuse.
GPU part (`-m gpu`): the interpreter-built table of the library boat race equals the
fixture; test-local games (tests/traced_games.py: classes of this test suite, not the rule
library; one mover, two movers with a sprite) at B = 65 536 against the generic tier and
against `oracle/table_replay.py` bit for bit, through rollout (update_table_kernel /
update_pair_kernel + render_kernel), the single fused kernel and play().
"""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from campx_amd import engine as engine_mod
from campx_amd import tabulate
from campx_amd.games import boat_race
from conftest import GOLDEN_DIR, REPO
import traced_games

REFERENCE_EXAMPLES = '/root/reference/examples'


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def _table_fixture():
  with np.load(os.path.join(GOLDEN_DIR, 'boat_race_table.npz')) as f:
    return {k: f[k] for k in f.files}


def _check_against_fixture(game, fix):
  """Every entry the reference could reach: next cell, reward, done, visibility, perf."""
  assert game.movers == ['A'] and game.init_cells == (int(fix['cells'][0]),)
  reached_cells = sorted({game.cells_of(i)[0] for i in np.flatnonzero(game.reached)})
  assert reached_cells == sorted(fix['cells'].tolist())
  for s, cell in enumerate(fix['cells']):
    for a in range(5):
      i = game.index_of((int(cell),), a)
      assert game.reached[i]
      assert game.next_cells[0, i] == fix['next_cell'][s, a], (cell, a)
      assert _same(game.reward[i], fix['reward'][s, a]), (cell, a)
      assert game.done[i] == fix['done'][s, a]
      assert game.visible[0, i] == fix['visible'][s, a]
      assert game.perf[i] == fix['perf'][s, a], (cell, a)
      board = game.model_board((int(game.next_cells[0, i]),))
      assert np.array_equal(board, fix['board'][s, a].astype(np.uint8))
  assert (fix['discount'] == 1.0).all() and (fix['done'] == 0).all()


# ------------------------------------------------------------------------------- CPU

@pytest.mark.skipif(not os.path.isdir(REFERENCE_EXAMPLES),
                    reason='reference tree not present (GPU box)')
def test_unmodified_reference_boat_race_tabulates_to_the_reference_table(tmp_path):
  """examples/boat_race.py imported IN PLACE, its zero-argument make_game() called as it is
  (its_showtime() held back so that the set-up game can be handed to the tabulator on a
  machine without a GPU), its classes run by this repo's generic tier over every reachable
  state: the table equals the one the reference's own engine produced."""
  code = r'''
import sys, numpy as np
sys.path.insert(0, %(repo)r)                       # this repo: `campx` -> campx_amd
sys.path.append(%(ref)r)                           # boat_race.py only
import campx, boat_race
assert campx.__file__.startswith(%(repo)r) and boat_race.__file__.startswith(%(ref)r)
from campx_amd import engine, tabulate
from campx_amd.games.boat_race import performance_masks
held = engine.Engine.its_showtime
engine.Engine.its_showtime = lambda self: (None, None, None)
game, _, _, _ = boat_race.make_game()
engine.Engine.its_showtime = held
assert type(game.things['A']).__module__ == 'boat_race'
game.set_hidden_performance('A', performance_masks())     # the driver's masks a, b, c, d
t = tabulate.trace(game)
%(save)s
save_table(t, %(out)r)
''' % dict(repo=REPO, ref=REFERENCE_EXAMPLES, save=_SAVE_TABLE, out=str(tmp_path / 'table.npz'))
  subprocess.run([sys.executable, '-c', code], check=True)
  game, got = _traced_from_npz(tmp_path / 'table.npz', 5, 5)
  # (round 5: the reference's classes run on lane tensors, a whole level of the state graph per
  # frame - 25 frames; the one-frame-per-play walker, CAMPX_TABULATE=walk, spends 8 x 5, each twice = 80)
  assert int(got['n_states']) == 8 and int(got['n_plays']) == 25
  _check_against_fixture(game, _table_fixture())


def _traced_from_npz(path, rows, cols):
  with np.load(path) as f:
    got = {k: f[k] for k in f.files}
  game = tabulate.TracedGame()
  game.rows, game.cols = rows, cols
  game.movers = [chr(c) for c in got['movers']]
  game.init_cells = tuple(int(c) for c in got['init'])
  game.z_order = [chr(c) for c in got['z']]
  game.mode_orders = [game.z_order]
  game.absent_cells = [set() for _ in game.movers]
  game.chars = [chr(c) for c in got['chars']]
  game.backdrop = got['backdrop']
  game.statics = [(chr(c), m) for c, m in zip(got['statics'], got['static_masks'])]
  for k in ('next_cells', 'visible', 'reward', 'done', 'discount', 'perf', 'reached'):
    setattr(game, k, got[k])
  return game, got


_SAVE_TABLE = r'''
def save_table(t, path):
  import numpy as np
  np.savez(path, next_cells=t.next_cells, visible=t.visible, reward=t.reward, done=t.done,
           discount=t.discount, perf=t.perf, reached=t.reached, init=np.array(t.init_cells),
           n_states=t.n_states, n_plays=t.n_plays, movers=np.array([ord(c) for c in t.movers]),
           backdrop=t.backdrop, statics=np.array([ord(c) for c, _ in t.statics]),
           static_masks=np.array([m for _, m in t.statics]).reshape(len(t.statics), t.rows, t.cols),
           z=np.array([ord(c) for c in t.z_order]), chars=np.array([ord(c) for c in t.chars]))
'''


@pytest.mark.skipif(not os.path.isdir(REFERENCE_EXAMPLES),
                    reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('name', ['demo1', 'demo2', 'demo3', 'demo4', 'demo5'])
def test_unmodified_notebook_classes_tabulate_to_their_goldens(name, golden, tmp_path):
  """The Demo 1-4 notebooks' own classes (cells exec'd unchanged from the .ipynb under
  /root/reference, against this repo's `campx` alias), tabulated - Demo 1-3 with the plain
  one-hot LISTS those notebooks pass to play() - and the table walked over the golden
  action streams: the frames the REFERENCE engine produced (tests/golden/demoN.npz)."""
  from oracle.table_replay import TableWalker
  from test_generic_golden import GOLDEN_OF, NOTEBOOK_GAMES
  notebook, cells, build = NOTEBOOK_GAMES[name]
  code = r'''
import json, sys, collections, itertools
import numpy as np, torch, six
sys.path.insert(0, %(repo)r)                       # this repo: `campx` -> campx_amd
from campx import things
from campx.ascii_art import ascii_art_to_game, Partial
from campx import engine
from campx_amd import tabulate
nb = json.load(open(%(ref)r + '/' + %(notebook)r))
ns = dict(globals())
for i in %(cells)r:
    exec(compile(''.join(nb['cells'][i]['source']), 'cell %%d' %% i, 'exec'), ns)
game = eval(%(build)r, ns)
assert type(game.things['A']).__module__ == '__main__'    # the notebook's own class
if %(name)r not in ('demo4', 'demo5'):     # Demo 1-3 call play([1, 0, 0, 0, 0])
    game.set_action_set([[int(i == a) for i in range(5)] for a in range(5)])
%(save)s
save_table(tabulate.trace(game, actions=game._action_set), %(out)r)
''' % dict(repo=REPO, ref=REFERENCE_EXAMPLES, notebook=notebook, cells=cells, build=build,
           name=name, save=_SAVE_TABLE, out=str(tmp_path / 'table.npz'))
  subprocess.run([sys.executable, '-c', code], check=True)
  gold = golden(GOLDEN_OF.get(name, name))
  H, W = gold['board'].shape[-2:]
  traced, _ = _traced_from_npz(tmp_path / 'table.npz', H, W)
  assert [ord(c) for c in traced.chars] == gold['chars'].tolist() and traced.movers == ['A']
  T, N = gold['actions'].shape
  walker = TableWalker(traced, N)
  want = walker.rollout(gold['actions'], reset_first=True)
  for k in ('reward', 'discount', 'done'):
    assert _same(want[k], gold[k]), k
  for t in range(T):
    board, layered = walker.render(want['cells'][:, t].astype(np.int64))
    assert np.array_equal(board, gold['board'][t + 1]), t
    assert np.array_equal(layered, gold['layered'][t + 1].astype(np.int8)), t


def test_library_boat_race_tabulates_to_the_reference_table():
  """The rule-library classes (ordinary Python update() bodies too) through the tabulator."""
  game = tabulate.trace(boat_race.build())
  assert game.n_states == 8 and game.any_reward and game.has_perf
  _check_against_fixture(game, _table_fixture())
  spec = tabulate.to_spec(game)
  assert spec.table_only == 1 and spec.table_valid == 1 and spec.n_rules == 0
  assert (spec.n_dyn, spec.n_static, spec.n_layers) == (1, 5, 7)
  assert (spec.dyn_row0[0], spec.dyn_col0[0]) == (1, 1)


def _walk_generic(build, traced, steps, seed):
  """Random walk on the generic tier, one environment; the table must predict every frame."""
  from oracle.table_replay import TableWalker
  rng = np.random.RandomState(seed)
  actions = rng.randint(0, 5, size=(steps, 1)).astype(np.int8)
  want = TableWalker(traced, 1).rollout(actions, reset_first=True)
  walker = TableWalker(traced, 1)
  game = build()
  obs, _, _ = game.its_showtime()
  onehot = tabulate.default_actions()
  ended = 0
  for t in range(steps):
    if game.game_over:
      game = build()
      game.its_showtime()
    obs, reward, discount = game.play(onehot[int(actions[t, 0])])
    board, layered = walker.render(want['cells'][:, t].astype(np.int64))
    assert np.array_equal(obs.board.numpy(), board[0].astype(np.uint8)), t
    assert np.array_equal(obs.layered_board.numpy(), layered[0]), t
    got = np.float32(np.nan) if reward is None else np.float32(float(reward))
    assert _same(got, want['reward'][t, 0]), t
    assert float(discount) == want['discount'][t, 0] and int(game.game_over) == want['done'][t, 0]
    ended += int(game.game_over)
  return ended


@pytest.mark.parametrize('name', sorted(traced_games.GAMES))
def test_test_local_games_tabulate_and_predict_the_generic_tier(name):
  build = traced_games.GAMES[name]
  traced = tabulate.trace(build())
  if name == 'ice_rink':
    assert traced.movers == ['A'] and [c for c, _ in traced.statics] == ['#', 'o', 'E']
    assert traced.done[traced.reached].sum() > 0
  elif name == 'toll_road':
    assert traced.movers == ['A'] and traced.discount_list == [1.0, 0.5, 0.25, 0.75]
  elif name == 'trio':
    assert traced.movers == ['A', 'L', 'T'] and traced.n == 35 ** 3 * 5   # 64-bit tuple entries
    both = traced.reached & (traced.next_cells[0] == traced.next_cells[1])
    assert both.any() and (traced.visible[0][both] == 0).all()   # the lift hides the walker
  elif name == 'vault':
    # walker, key, door, gem: four tracked things; three of them leave / enter the board
    assert traced.movers == ['A', 'k', 'D', '$'] and traced.absent_cells == [set(), {0}, {0}, {0}]
    assert traced.init_visible == [1, 1, 1, 0]               # the gem starts hidden
    gone = traced.reached & (traced.next_cells[1] == 0)      # key picked up
    assert gone.any() and (traced.visible[1][gone] == 0).all()
    shown = traced.reached & (traced.visible[3] == 1)        # the gem shows: the door is open
    assert shown.any() and (traced.next_cells[2][shown] == 0).all()
  elif name == 'burrow':
    assert traced.movers == ['A'] and len(traced.mode_orders) == 2    # above / under ground
    under = traced.reached & (traced.next_cells[1] == 1)
    assert under.any() and (traced.visible[0][under] == 0).any()
  else:
    assert traced.movers == ['A', 'G']                      # a drape and a sprite
    hidden = traced.reached & (traced.visible[0] == 0)      # the ghost stands on the walker
    assert hidden.any() and (traced.done[hidden] == 1).all()
    assert (traced.visible[1][traced.reached] == 1).all()
  assert len(set(traced.reward[traced.reached].tolist())) > (1 if name == 'toll_road' else 2)
  ended = _walk_generic(build, traced, 400, seed=5)
  assert ended > 0


def _traced_golden(name):
  with np.load(os.path.join(GOLDEN_DIR, 'traced_' + name + '.npz')) as f:
    return {k: f[k] for k in f.files}


@pytest.mark.parametrize('name', sorted(traced_games.GAMES))
def test_reference_engine_goldens_on_the_generic_tier_and_through_the_table(name):
  """tests/golden/traced_<name>.npz: the same test-local classes run by the REFERENCE's
  engine (make_traced_golden.py).  This repo's generic tier gives the same frames, and so
  does walking the table tabulated from them - discounts other than 0 / 1 included."""
  from oracle.table_replay import TableWalker
  gold = _traced_golden(name)
  T, N = gold['actions'].shape
  build = traced_games.GAMES[name]
  onehot = tabulate.default_actions()
  for n in range(4):
    game = build()
    obs, _, _ = game.its_showtime()
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      if game.game_over:
        game = build()
        game.its_showtime()
      obs, reward, discount = game.play(onehot[int(gold['actions'][t, n])])
      assert np.array_equal(obs.board.numpy(), gold['board'][t + 1, n].astype(np.uint8)), (n, t)
      assert np.array_equal(obs.layered_board.numpy(), gold['layered'][t + 1, n])
      assert _same(np.float32(np.nan if reward is None else float(reward)), gold['reward'][t, n])
      assert np.float32(discount) == gold['discount'][t, n]
      assert int(game.game_over) == gold['done'][t, n]
  traced = tabulate.trace(build())
  assert [ord(c) for c in traced.chars] == gold['chars'].tolist()
  walker = TableWalker(traced, N)
  want = walker.rollout(gold['actions'], reset_first=True)
  for k in ('reward', 'discount', 'done'):
    assert _same(want[k], gold[k]), k
  for t in range(T):
    board, layered = walker.render(want['cells'][:, t].astype(np.int64))
    assert np.array_equal(board, gold['board'][t + 1]), t
    assert np.array_equal(layered, gold['layered'][t + 1].astype(np.int8)), t
  if name == 'toll_road':
    assert traced.discount_list == [1.0, 0.5, 0.25, 0.75]
    assert set(np.unique(gold['discount']).tolist()) == {0.25, 0.5, 0.75, 1.0}


@pytest.mark.parametrize('cls,why', [
    (traced_games.Stepper, r"the_plot\['n'\] \(\d+ different values"),
    (traced_games.Discounter, 'more than 15 distinct discounts'),
])
def test_games_the_table_model_is_not_exact_for_are_refused(cls, why):
  with pytest.raises(tabulate.TabulationError, match=why):
    tabulate.trace(traced_games.refused(cls))


def test_a_drape_that_grows_is_tracked_cell_by_cell_until_that_is_too_many():
  """Until round 6 a moving drape on two cells was refused ("covers 2 cells"); now every cell it
  ever covers is a tracked thing of its own - the four cells of this one's row - and the refusal
  comes when they are more than the kernels track (tests/lanes_probes.py's trail: twelve)."""
  traced = tabulate.trace(traced_games.refused(traced_games.Grower), cache=False)
  assert traced.movers == ['A'] * 4 and traced.piece_cell == [0, 1, 2, 3] and traced.n_states == 4
  import lanes_probes
  with pytest.raises(tabulate.TabulationError, match=r"'A' cover several cells that come and go - 12 tracked cells"):
    tabulate.trace(lanes_probes.game(lanes_probes.Grower)(), cache=False)


def test_a_z_order_change_that_reorders_the_scenery_is_refused():
  """Moving things may change places in the z-order (the burrow game above); two overlapping
  STATIC drapes swapping would change the one scenery row the render kernels lay down."""
  with pytest.raises(tabulate.TabulationError, match='re-orders overlapping static things'):
    tabulate.trace(traced_games.refused_scenery())


def test_z_order_modes_are_tabulated_as_one_more_tracked_value():
  traced = tabulate.trace(traced_games.burrow())
  assert traced.movers == ['A'] and traced.n_tracked == 2
  assert [''.join(z) for z in traced.mode_orders] == ['du$=A#', 'Adu$=#']
  spec = tabulate.to_spec(traced)
  assert spec.n_dyn == 2 and spec.dyn_z[1] == 0 and spec.table_only == 1
  assert traced.visible[1].max() == 0           # the mode is never painted
  # under ground on the lawn: the entry says "not the character its cell shows"
  lawn = 1 * 9 + 4
  i = traced.index_of((lawn, 1), 4)
  assert traced.reached[i] and traced.visible[0, i] == 0 and traced.next_cells[1, i] == 1
  j = traced.index_of((lawn, 0), 4)
  assert traced.reached[j] and traced.visible[0, j] == 1


def test_a_second_make_game_of_the_same_game_reuses_the_tabulation():
  """The reference's driver calls make_game() per episode (examples/reinforce.py:122): the
  tabulation is keyed by a fingerprint of the set-up engine (classes by the code of their
  methods, every attribute, curtains, groups, z-order, action set) and reused."""
  import time
  tabulate._CACHE.clear()
  first = tabulate.trace(traced_games.mirror())
  t0 = time.perf_counter()
  again = tabulate.trace(traced_games.mirror())
  from conftest import took_about
  assert again is first
  took_about(time.perf_counter() - t0, 0.5, 'a tabulation found in the cache')
  assert tabulate.trace(traced_games.mirror(), cache=False) is not first
  other = traced_games.toll_road()
  other.things['A'].bonus = 3                        # any attribute that differs: another key
  assert (tabulate.fingerprint(other, tabulate.default_actions()) !=
          tabulate.fingerprint(traced_games.toll_road(), tabulate.default_actions()))
  # different action objects are a different game as far as the table goes
  ints = [torch.eye(5)[a] * 1.0 for a in range(5)]
  assert tabulate.fingerprint(other, ints) == tabulate.fingerprint(other, tabulate.default_actions())
  assert tabulate.fingerprint(other, list(range(5))) != tabulate.fingerprint(other, ints)

  class Opaque(traced_games.Walker):                 # state the fingerprint cannot read
    def __init__(self, curtain, character):
      super(Opaque, self).__init__(curtain, character)
      self.gen = iter(())

  game = traced_games.ascii_art_to_game(['#####', '#A  #', '#####'], what_lies_beneath=' ',
                                        drapes={'A': Opaque, '#': traced_games.things.FixedDrape})
  assert tabulate.fingerprint(game, tabulate.default_actions()) is None


def test_too_large_a_state_space_is_refused_with_a_pointer_to_the_rule_library(monkeypatch):
  # (the one-frame-per-play walker's budget; the lane walker, which takes this game since round 5,
  # spends 65 frames on its 388 states)
  monkeypatch.setenv('CAMPX_TABULATE', 'walk')
  with pytest.raises(tabulate.TabulationError, match='campx_amd.rules'):
    tabulate.trace(traced_games.mirror(), max_plays=50)


def test_default_batch_makes_a_zero_argument_make_game_batched_and_loud_without_a_gpu():
  """`set_default_batch` (or CAMPX_BATCH): set-up code written for the reference builds a
  batched engine; with no HIP device that fails loudly - no CPU fallback."""
  assert engine_mod.get_default_batch() == (None, None)
  engine_mod.set_default_batch(8)
  try:
    game = traced_games.ice_rink()
    assert game.batch == 8
    if not torch.cuda.is_available():
      with pytest.raises(RuntimeError, match='needs a HIP device'):
        traced_games.make_game()
    assert traced_games.ice_rink(batch=None).batch is None      # explicit wins
  finally:
    engine_mod.set_default_batch(None)
  assert traced_games.ice_rink().batch is None
  out = subprocess.run(
      [sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from campx_amd import engine; '
       'print(engine.get_default_batch())' % REPO],
      env=dict(os.environ, CAMPX_BATCH='4096', CAMPX_DEVICE='cuda:0'), check=True,
      capture_output=True, text=True)
  assert out.stdout.strip() == "(4096, 'cuda:0')"


# ------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_interpreter_built_table_equals_the_reference_table():
  """campx_spec_compile() runs the rule interpreter KERNEL over every (cell, action) of the
  library boat race: the entries the reference can reach equal the reference's."""
  game = boat_race.build(batch=64, device='cuda')
  game.its_showtime()
  spec, fix = game.fused.spec, _table_fixture()
  assert spec.table_valid == 1 and spec.table_only == 0
  for s, cell in enumerate(fix['cells']):
    for a in range(5):
      tr = spec.table[int(cell) * 5 + a]
      assert tr.next_cell == fix['next_cell'][s, a], (cell, a)
      assert _same(np.float32(tr.reward), fix['reward'][s, a])
      assert tr.done == fix['done'][s, a] and tr.perf == fix['perf'][s, a]
      assert (0 if tr.paint & 0x80 else 1) == fix['visible'][s, a]
  # ... and the host tabulation of the same game fills the same table
  host = tabulate.to_spec(tabulate.trace(boat_race.build()))
  for s, cell in enumerate(fix['cells']):
    for a in range(5):
      x, y = spec.table[int(cell) * 5 + a], host.table[int(cell) * 5 + a]
      assert (x.next_cell, x.done, x.perf, x.paint) == (y.next_cell, y.done, y.perf, y.paint)
      assert _same(np.float32(x.reward), np.float32(y.reward))


def _generic_frames(build, actions):
  """The generic tier over one environment's action stream: boards, rewards, done."""
  onehot = tabulate.default_actions()
  game = build()
  game.its_showtime()
  boards, layered, rewards, dones = [], [], [], []
  for a in actions:
    if game.game_over:
      game = build()
      game.its_showtime()
    obs, reward, _ = game.play(onehot[int(a)])
    boards.append(obs.board.numpy().astype(np.int8))
    layered.append(obs.layered_board.numpy().astype(np.int8))
    rewards.append(np.float32(np.nan) if reward is None else np.float32(float(reward)))
    dones.append(int(game.game_over))
  return np.array(boards), np.array(layered), np.array(rewards, np.float32), np.array(dones, np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(traced_games.GAMES))
def test_test_local_games_at_full_batch_through_the_table_and_render_kernels(name):
  from oracle.table_replay import TableWalker
  from campx_amd import fused
  B, T = 65536, 100
  build = traced_games.GAMES[name]
  game = build(batch=B, device='cuda')
  first, reward0, discount0 = game.its_showtime()
  f = game.fused
  assert isinstance(f, fused.FusedGame) and f.traced is not None and f.spec.table_only == 1
  assert f.uses_table and reward0 is None and discount0 == 1.0
  traced = f.traced
  walker = TableWalker(traced, B)
  board0, layered0 = walker.render(walker.cells[:, :4])
  assert np.array_equal(first.board[:4].cpu().numpy(), board0)
  assert np.array_equal(first.layered_board[:4].cpu().numpy(), layered0)

  rng = np.random.RandomState(17)
  actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions), want_board=True)
  want = walker.rollout(actions)
  assert out['trace'] is not None                      # the two-kernel path
  trace = out['trace'].cpu().numpy()
  assert np.array_equal(trace & 0x7f, want['cells'])
  assert np.array_equal(trace >> 7, want['visible'])
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), want[k]), k
  assert want['done'].sum() > B // 10
  if name == 'toll_road':     # discounts that are neither 0 nor 1, from the table's codes
    assert set(np.unique(want['discount']).tolist()) == {0.25, 0.5, 0.75, 1.0}
  assert _same(f.ret.cpu().numpy(), walker.ret)
  # every environment's observation at a few frames, a strided sample at every frame
  for t in (0, 1, T // 2, T - 1):
    board, layered = walker.render(want['cells'][:, t].astype(np.int64))
    assert np.array_equal(out['obs'][t].cpu().numpy(), layered), t
    assert np.array_equal(out['board'][t].cpu().numpy(), board), t
  sample = np.arange(0, B, 1024)
  obs_sample = out['obs'][:, sample].cpu().numpy()
  for t in range(T):
    _, layered = walker.render(want['cells'][:, t][:, sample].astype(np.int64))
    assert np.array_equal(obs_sample[t], layered), t
  sums = out['obs'].sum(dim=2, dtype=torch.int32)
  assert int(sums.min()) == 1 and int(sums.max()) == 1      # each cell shows one character
  # the user's classes themselves, on the generic tier, for a few environments
  for env in (0, 1, 4097, B - 1):
    boards, layered, rewards, dones = _generic_frames(build, actions[:, env])
    assert np.array_equal(out['board'][:, env].cpu().numpy(), boards), env
    assert np.array_equal(out['obs'][:, env].cpu().numpy(), layered), env
    assert _same(out['reward'][:, env].cpu().numpy(), rewards), env
    assert np.array_equal(out['done'][:, env].cpu().numpy(), dones), env

  # the same frames one play() at a time (step_table_kernel / step_pair_kernel) ...
  game2 = build(batch=B, device='cuda')
  game2.its_showtime()
  for t in range(12):
    obs, reward, discount = game2.play(torch.from_numpy(actions[t]))
    assert torch.equal(obs.layered_board, out['obs'][t]), t
    assert torch.equal(obs.board, out['board'][t]), t
    assert _same(reward.cpu().numpy(), want['reward'][t])
    assert _same(discount.cpu().numpy(), want['discount'][t])
  # ... and, for the one-mover game, in the single fused kernel (no trace buffer)
  if f.n_dyn == 1:
    fused.SPLIT_ROLLOUT = False
    try:
      game3 = build(batch=4096, device='cuda')
      game3.its_showtime()
      alone = game3.rollout(torch.from_numpy(actions[:, :4096].copy()))
      assert alone['trace'] is None
      assert torch.equal(alone['obs'], out['obs'][:, :4096])
      assert _same(alone['reward'].cpu().numpy(), want['reward'][:, :4096])
    finally:
      fused.SPLIT_ROLLOUT = True


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(traced_games.GAMES))
def test_reference_engine_goldens_on_the_gpu(name):
  """What the reference's engine did with the test-local classes (traced_<name>.npz) against
  the HIP path running the table tabulated from them: rollout, then play() frame by frame."""
  gold = _traced_golden(name)
  T, N = gold['actions'].shape
  build = traced_games.GAMES[name]
  game = build(batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
  assert np.array_equal(first.board.cpu().numpy(), gold['board'][0])
  assert np.array_equal(first.layered_board.cpu().numpy(), gold['layered'][0].astype(np.int8))
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert np.array_equal(out['obs'].cpu().numpy(), gold['layered'][1:].astype(np.int8))
  assert np.array_equal(out['board'].cpu().numpy(), gold['board'][1:])
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), gold[k]), k
  game = build(batch=N, device='cuda')
  game.its_showtime()
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert np.array_equal(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert np.array_equal(obs.layered_board.cpu().numpy(), gold['layered'][t + 1].astype(np.int8))
    assert _same(reward.cpu().numpy(), gold['reward'][t]), t
    assert _same(discount.cpu().numpy(), gold['discount'][t]), t
    assert np.array_equal(game.fused.done.cpu().numpy(), gold['done'][t])


@pytest.mark.gpu
def test_zero_argument_make_game_runs_batched_with_a_default_batch():
  """What `north_star` asks of examples/boat_race.py (whose file cannot travel to the GPU
  box): set-up code written for ONE environment - a zero-argument make_game() that calls
  its_showtime() itself, play() with ONE one-hot action - runs unchanged on the HIP path."""
  engine_mod.set_default_batch(4096, 'cuda')
  try:
    game, board, reward, discount = traced_games.make_game()
  finally:
    engine_mod.set_default_batch(None)
  assert game.batch == 4096 and game.fused.traced is not None
  assert tuple(board.layered_board.shape) == (4096, 5, 6, 9) and reward is None
  single = traced_games.ice_rink()
  single.its_showtime()
  for a in (0, 3, 2, 1, 3):                              # slides; the last one stops on the exit
    onehot = torch.zeros(5)
    onehot[a] = 1
    obs, reward, discount = game.play(onehot)            # one action for every environment
    ref_obs, ref_reward, ref_discount = single.play(onehot)
    assert np.array_equal(obs.board[0].cpu().numpy(), ref_obs.board.numpy())
    assert np.array_equal(obs.board[4095].cpu().numpy(), ref_obs.board.numpy())
    assert float(reward[7]) == float(ref_reward) and float(discount[7]) == float(ref_discount)
  assert single.game_over and bool(game.fused.done.all())
  fresh = traced_games.ice_rink()                        # (every environment starts over)
  fresh.its_showtime()
  _, want, _ = fresh.play(torch.tensor([0., 1., 0., 0., 0.]))
  obs, reward, _ = game.play([0, 1, 0, 0, 0])            # the notebooks' plain one-hot list
  assert float(reward[0]) == float(want) and bool((reward == reward[0]).all())
  obs, reward, _ = game.play(4)                          # ... or one id
  assert bool((reward == -0.125).all())


@pytest.mark.gpu
def test_a_two_mover_game_cannot_run_without_its_table():
  """No rules to interpret: the single fused kernel (two movers) is refused, loudly."""
  from campx_amd import _hip, fused
  fused.SPLIT_ROLLOUT = False
  try:
    game = traced_games.mirror(batch=256, device='cuda')
    game.its_showtime()
    with pytest.raises(RuntimeError, match='GameSpec failed validation'):
      game.rollout(torch.zeros((4, 256), dtype=torch.int8))
  finally:
    fused.SPLIT_ROLLOUT = True
