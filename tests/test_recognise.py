"""User classes of the Hello World kind - rigidly translating multi-cell things - reach the
shape tier without being re-typed (`campx_amd/recognise.py`).

CPU: (a) cells 3-4 of the reference's examples/Hello World Example.ipynb exec'd unchanged
(skipped where /root/reference is absent) recognise to a CampxShapeSpec BYTE-EQUAL to the one
the library's re-typed classes lower to, which is also a committed fixture
(`tests/golden/hello_world_spec.npz`); (b) a test-local game of plain-Python multi-cell things
(tests/shape_local.py) is recognised, and the recognised description replays the frames the
REFERENCE engine produced with the same classes (`parade.npz`, make_golden.py) through the C
oracle; (c) games that are not shape games are refused, with the frame that shows it.
GPU (`-m gpu`): the test-local game through `shape_rollout_kernel` against that golden and
against the C oracle on random streams; Hello World itself as a user's own classes
(examples/hello_world_batched.py, written independently of the notebook) through
`set_default_batch()` and recognition, against `hello_world.npz`.
(d) round 5: recognition is a proof - every thing alone, at every position it can reach, under
every action, with recording stand-ins for everything else update() is handed: a one-cell
exception, a reward that differs on one row and a look at `layers` are all refused.
"""

import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from campx_amd import gamespec, recognise, tabulate
from campx_amd import engine as engine_mod
from campx_amd.games import hello_world
from conftest import GOLDEN_DIR, REPO
import shape_local

NOTEBOOK = '/root/reference/examples/Hello World Example.ipynb'


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def _spec_bytes(spec):
  return ctypes.string_at(ctypes.addressof(spec), ctypes.sizeof(spec))


def _golden(name):
  with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as f:
    return {k: f[k] for k in f.files}


# ------------------------------------------------------------------------------- CPU

def test_library_hello_world_lowers_to_the_committed_spec():
  spec = gamespec.lower_shapes(gamespec.describe(hello_world.build()))
  assert _spec_bytes(spec) == _golden('hello_world_spec')['spec'].tobytes()


@pytest.mark.skipif(not os.path.exists(NOTEBOOK), reason='reference tree not present (GPU box)')
def test_the_notebooks_own_hello_world_classes_recognise_to_the_same_spec(tmp_path):
  """Cells 3 and 4 exec'd as they are against this repo's `campx` alias; `make_game()` is the
  notebook's; nothing tells the engine what the classes do or that actions are integers."""
  code = r'''
import json, sys, ctypes
import numpy as np, torch, six, itertools, collections
sys.path.insert(0, %(repo)r)
from campx import things
from campx.ascii_art import ascii_art_to_game, Partial
from campx import engine
from campx_amd import recognise, gamespec
nb = json.load(open(%(nb)r))
ns = dict(globals())
for i in (3, 4):
    exec(compile(''.join(nb['cells'][i]['source']), 'cell %%d' %% i, 'exec'), ns)
game = ns['make_game']()
assert type(game.things['@']).__module__ == '__main__' and game._action_set is None
assert not gamespec.is_rule_game(game)
actions = recognise.detect_actions(game)
assert actions == [0, 1, 2, 3, 4] and recognise.looks_like_shapes(game, actions)
spec = gamespec.lower_shapes(recognise.shapes(game, actions))
open(%(out)r, 'wb').write(ctypes.string_at(ctypes.addressof(spec), ctypes.sizeof(spec)))
''' % dict(repo=REPO, nb=NOTEBOOK, out=str(tmp_path / 'spec.bin'))
  subprocess.run([sys.executable, '-c', code], check=True)
  got = open(tmp_path / 'spec.bin', 'rb').read()
  assert got == _spec_bytes(gamespec.lower_shapes(gamespec.describe(hello_world.build())))
  assert got == _golden('hello_world_spec')['spec'].tobytes()


def test_test_local_parade_is_recognised():
  game = shape_local.parade()
  assert not gamespec.is_rule_game(game)
  actions = recognise.detect_actions(game)
  assert actions == list(range(5)) and recognise.looks_like_shapes(game, actions)
  desc = recognise.shapes(game, actions)
  by = {e.char: e for e in desc.entities}
  assert [e.char for e in desc.entities] == list('ZbW#h') and desc.z_order == list('h#WbZ')
  assert by['#'].kind == 'fixed' and all(by[c].kind == 'shape' for c in 'ZbWh')
  assert by['W'].params['drow'] == [-1, 2, 0, 0, 0] and by['W'].params['dcol'] == [0, 0, -1, 3, 0]
  assert by['W'].params['rewards'] == [0.25, 0.25, 0.25, 0.25, 0.5]
  assert by['Z'].params['drow'] == [1, -1, 2, -2, 0] and by['Z'].params['dcol'] == [2, -2, -1, 1, 0]
  assert by['Z'].params['rewards'] == [None, None, None, -1.0, None]
  assert by['Z'].params['quit_actions'] == [4] and by['W'].params['quit_actions'] == []
  assert by['b'].params['dcol'] == [1, 1, 1, 1, 0] and by['b'].params['sprite']
  assert by['h'].params['drow'] == [-1, 1, 0, 0, 0]
  spec = gamespec.lower_shapes(desc)
  assert spec.first_drape == 1 and spec.n_things == 5       # the hiker leaves a trail


def test_recognised_parade_replays_the_reference_engines_frames_through_the_oracle():
  """parade.npz: shape_local's classes on the REFERENCE engine.  The description recognised
  from the same classes on this repo's generic tier, run by the C oracle's shape path."""
  from oracle import cpu
  gold = _golden('parade')
  desc = recognise.shapes(shape_local.parade())
  assert [ord(c) for c in desc.chars] == gold['chars'].tolist()
  og = cpu.OracleGame.from_description(desc)
  ref = og.rollout(gold['actions'], reset_first=True)
  assert _same(ref['obs'], gold['layered'][1:].astype(np.int8))
  assert _same(ref['board'], gold['board'][1:])
  for k in ('reward', 'discount', 'done'):
    assert _same(ref[k], gold[k]), k
  assert gold['done'].sum() > 10
  # the trail: by the last frame of the always-up environment the hiker's column is full
  col = gold['board'][-1, 2, :, 3]
  assert (col == ord('h')).sum() >= 6        # (walls and the zigzag are in front of it)


def test_games_that_are_not_shape_games_are_refused_with_the_frame_that_shows_it():
  with pytest.raises(recognise.RecogniseError, match=r"'L' looks at / touches things\['#'\]"):
    recognise.shapes(shape_local.not_a_shape(), list(range(5)))
  # a sprite that stops at the right edge instead of wrapping: 33 columns away from where it
  # starts - found by the enumeration of every position it can reach
  with pytest.raises(recognise.RecogniseError,
                     match=r"'c' moved by \(rows 0, cols 33\) from its start, action 1: it is not moved"):
    recognise.shapes(shape_local.clamps_at_the_edge(), list(range(5)))
  import traced_games
  # a one-cell walker stopped by walls: the tabulator's game, not this module's
  game = traced_games.mirror()
  assert not recognise.looks_like_shapes(game, tabulate.default_actions())
  with pytest.raises(recognise.RecogniseError):
    recognise.shapes(game)
  # state outside the curtains
  from campx import things
  from campx.ascii_art import ascii_art_to_game

  class Counting(things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      self.curtain.set_(torch.roll(self.curtain, 1, 1))
      the_plot['n'] = the_plot.get('n', 0) + 1

  game = ascii_art_to_game(['CC   ', '     '], what_lies_beneath=' ', drapes={'C': Counting},
                           z_order='C', update_schedule='C')
  with pytest.raises(recognise.RecogniseError, match=r"'C' looks at / touches the_plot\['n'\]"):
    recognise.shapes(game, list(range(5)))

  class Tired(things.Drape):
    """State in an attribute of its own: after three moves it stops."""
    def __init__(self, curtain, character):
      super(Tired, self).__init__(curtain, character)
      self.moves = 0

    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None or self.moves >= 3:
        return
      self.moves += 1
      self.curtain.set_(torch.roll(self.curtain, 1, 1))

  game = ascii_art_to_game(['TT   ', '     '], what_lies_beneath=' ', drapes={'T': Tired},
                           z_order='T', update_schedule='T')
  with pytest.raises(recognise.RecogniseError, match=r"state outside its curtain / position changed"):
    recognise.shapes(game, list(range(5)))


def test_action_format_is_detected_from_the_classes():
  import traced_games
  assert all(torch.is_tensor(a) for a in recognise.detect_actions(traced_games.ice_rink()))
  assert recognise.detect_actions(shape_local.parade()) == list(range(5))
  game = shape_local.parade()
  game.set_action_set([4, 3, 2, 1, 0])
  assert recognise.detect_actions(game) == [4, 3, 2, 1, 0]


# ------------------------------------------------------- recognition is a proof (round 5)

def _hello(**classes):
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import hello_world_batched as ex
  return ex, ex.make_game(**classes)


def test_hello_world_with_one_trap_door_cell_is_refused():
  """sprite '3' jumps to (0, 0) when it stands on (2, 3) - a cell the sampled walks
  of round 4 never reached, so the game was accepted and would have run wrong on the device.
  The enumeration visits every cell a thing can reach: refused, naming thing and place."""
  ex, _ = _hello()

  class TrapDoor(ex.Bishop):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is not None and self.character == '3' and tuple(self.position) == (2, 3):
        self._position = self.Position(0, 0)
        return
      ex.Bishop.update(self, actions, board, layers, backdrop, all_things, the_plot)

  game = ex.make_game(bishop=TrapDoor)
  assert recognise.looks_like_shapes(game, list(range(5)))
  with pytest.raises(recognise.RecogniseError, match=r"'3' moved by \(rows \d+, cols \d+\) from its "
                     r"start, action \d: it is not moved by the action's offset"):
    recognise.shapes(game)
  # and through the engine's own front door: the tabulator cannot take it either
  game = ex.make_game(bishop=TrapDoor)
  game._batch, game._device = 4, 'cpu'
  with pytest.raises(ValueError):
    with pytest.MonkeyPatch.context() as mp:
      mp.setattr(torch.cuda, 'is_available', lambda: True)
      game.its_showtime()


def test_a_drape_whose_reward_differs_on_one_row_is_refused():
  ex, _ = _hello()

  class Generous(ex.Scroller):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      ex.Scroller.update(self, actions, board, layers, backdrop, all_things, the_plot)
      # the banner's top row (13 cells) starts on row 7; one point more whenever it stands on row 2
      if actions is not None and actions != ex.QUIT and int(self.curtain[2].sum()) == 13 \
          and int(self.curtain[1].sum()) == 0:
        the_plot.add_reward(1)

  game = ex.make_game(scroller=Generous)
  with pytest.raises(recognise.RecogniseError, match=r"'@' moved by .* its reward / termination is not"):
    recognise.shapes(game)


def test_a_sprite_that_reads_layers_board_things_or_the_plot_is_refused():
  ex, _ = _hello()

  def bishop_that(look):
    class Peeking(ex.Bishop):
      def update(self, actions, board, layers, backdrop, all_things, the_plot):
        if actions is not None:
          look(self, board, layers, backdrop, all_things, the_plot)
        ex.Bishop.update(self, actions, board, layers, backdrop, all_things, the_plot)
    return Peeking

  cases = [
      (lambda self, b, l, bd, th, p: bool(l['@'][0, 0]), r"layers\['@'\]"),
      (lambda self, b, l, bd, th, p: int(b.sum()), r"board \(sum\)"),
      (lambda self, b, l, bd, th, p: bd.curtain.numpy(), r"backdrop\.curtain"),
      (lambda self, b, l, bd, th, p: th['@'].curtain, r"things\['@'\]"),
      (lambda self, b, l, bd, th, p: p.get('anything'), r"the_plot\['anything'\] \(read\)"),
      (lambda self, b, l, bd, th, p: p.__setitem__('n', 1), r"the_plot\['n'\] \(written\)"),
      (lambda self, b, l, bd, th, p: p.frame, r"reads the_plot\.frame"),
      (lambda self, b, l, bd, th, p: p.change_z_order('1', '4'), r"z-order"),
  ]
  for look, pattern in cases:
    with pytest.raises(recognise.RecogniseError, match=pattern):
      recognise.shapes(ex.make_game(bishop=bishop_that(look)))
  # what is NOT a look: the thing's own entry of `things`, shapes and dtypes, the palette, the log
  harmless = bishop_that(lambda self, b, l, bd, th, p: (
      th[self.character].position, b.shape, l['@'].dtype, len(l), bd.palette, p.log('moved')))
  desc = recognise.shapes(ex.make_game(bishop=harmless))
  assert _spec_bytes(gamespec.lower_shapes(desc)) == _golden('hello_world_spec')['spec'].tobytes()


_PEERS = {}


def test_a_sprite_that_reaches_another_through_a_global_or_a_closure_is_refused():
  """The recording stand-ins see what update() is HANDED; a live Sprite kept in a module global
  or a closure is read without them knowing, and the per-thing proof - on a deep copy - would see
  it stand still.  Refused statically, by the name of the route (tabulate.reached_behind_the_engine)."""
  ex, _ = _hello()

  class ThroughGlobal(ex.Bishop):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      ex.Bishop.update(self, actions, board, layers, backdrop, all_things, the_plot)
      if actions is not None and _PEERS['4'].position == (2, 3):
        self._teleport((0, 0))

  game = ex.make_game(bishop=ThroughGlobal)
  _PEERS['4'] = game.things['4']
  with pytest.raises(recognise.RecogniseError, match=r"a live \w+ through the module global '_PEERS'"):
    recognise.shapes(game)

  peers = []

  class ThroughClosure(ex.Bishop):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      ex.Bishop.update(self, actions, board, layers, backdrop, all_things, the_plot)
      if actions is not None and peers[0].things['4'].position == (2, 3):
        self._teleport((0, 0))

  game = ex.make_game(bishop=ThroughClosure)
  peers.append(game)
  with pytest.raises(recognise.RecogniseError, match=r"a live Engine through the closure variable 'peers'"):
    recognise.shapes(game)


def test_a_backdrop_that_changes_or_looks_is_refused():
  from campx import things
  from campx.ascii_art import ascii_art_to_game
  C = shape_local.bind(things)

  class Tide(things.Backdrop):
    def update(self, actions, board, layers, things_, the_plot):
      if actions is not None and int(actions) == 2:
        self.curtain[0, 0] = ord('~')

  class Nosy(things.Backdrop):
    def update(self, actions, board, layers, things_, the_plot):
      if actions is not None:
        things_['W'].curtain.sum()

  art = ['        ', '  WW    ', '  W     ', '        ']
  for backdrop, pattern in ((Tide, r'the Backdrop, action 2: it changes'),
                            (Nosy, r"things\['W'\]")):
    game = ascii_art_to_game(art, what_lies_beneath=' ', drapes={'W': C.Wave}, backdrop=backdrop,
                             z_order='W', update_schedule='W')
    with pytest.raises(recognise.RecogniseError, match=pattern):
      recognise.shapes(game, list(range(5)))


def test_recognising_hello_world_takes_seconds():
  import time
  ex, game = _hello()
  t0 = time.perf_counter()
  desc = recognise.shapes(game)
  from conftest import took_about
  took_about(time.perf_counter() - t0, 10.0, 'recognising Hello World')
  assert _spec_bytes(gamespec.lower_shapes(desc)) == _golden('hello_world_spec')['spec'].tobytes()


# ------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_parade_on_the_shape_kernel_against_the_reference_engines_golden():
  from campx_amd import shapes
  gold = _golden('parade')
  T, N = gold['actions'].shape
  game = shape_local.parade(batch=N, device='cuda')
  obs, reward, discount = game.its_showtime()
  assert isinstance(game.fused, shapes.ShapeGame) and reward is None and discount == 1.0
  assert [ord(c) for c in game.fused.chars] == gold['chars'].tolist()
  assert _same(obs.layered_board.cpu().numpy(), gold['layered'][0].astype(np.int8))
  assert _same(obs.board.cpu().numpy(), gold['board'][0])
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  for key, want in (('obs', gold['layered'][1:].astype(np.int8)), ('board', gold['board'][1:]),
                    ('reward', gold['reward']), ('discount', gold['discount']),
                    ('done', gold['done'])):
    assert _same(out[key].cpu().numpy(), want), key
  game = shape_local.parade(batch=N, device='cuda')
  game.its_showtime()
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert _same(obs.layered_board.cpu().numpy(), gold['layered'][t + 1].astype(np.int8)), t
    assert _same(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert _same(reward.cpu().numpy(), gold['reward'][t]), t
    assert _same(discount.cpu().numpy(), gold['discount'][t]), t


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [7, 4096, 32768])
def test_parade_random_streams_against_the_oracle_and_the_generic_tier(batch):
  from oracle import cpu
  game = shape_local.parade(batch=batch, device='cuda')
  game.its_showtime()
  og = cpu.OracleGame.from_description(recognise.shapes(shape_local.parade()))
  rng = np.random.RandomState(batch)
  streams = []
  for launch, T in enumerate([1, 70, 33]):
    actions = rng.choice(5, size=(T, batch), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    streams.append(actions)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    for k in ('obs', 'board', 'reward', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), ref[k]), (launch, k)
  # the user's classes themselves on the generic tier, two environments of the long launch
  whole = np.concatenate(streams)
  game = shape_local.parade(batch=batch, device='cuda')
  game.its_showtime()
  out = game.rollout(torch.from_numpy(whole), want_board=True)
  for env in (0, batch - 1):
    single = shape_local.parade()
    single.its_showtime()
    for t in range(whole.shape[0]):
      if single.game_over:
        single = shape_local.parade()
        single.its_showtime()
      obs, reward, discount = single.play(int(whole[t, env]))
      assert np.array_equal(out['board'][t, env].cpu().numpy(), obs.board.numpy().astype(np.int8)), (env, t)
      want = np.float32(np.nan) if reward is None else np.float32(float(reward))
      assert _same(out['reward'][t, env].cpu().numpy(), want), (env, t)
      assert int(out['done'][t, env]) == int(single.game_over)


@pytest.mark.gpu
def test_user_written_hello_world_runs_batched_through_recognition():
  """The notebook's cells cannot travel to the GPU box; examples/hello_world_batched.py holds
  the same GAME written independently (integers as actions, a zero-argument make_game());
  `set_default_batch` is the only addition.  Same spec as the notebook's own cells recognise to
  (the CPU test above), same frames as `hello_world.npz`, which make_golden.py made with the
  notebook's cells on the reference engine."""
  from campx_amd import shapes
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import hello_world_batched as ex

  gold = _golden('hello_world')
  T, N = gold['actions'].shape
  engine_mod.set_default_batch(N, 'cuda')
  try:
    game = ex.make_game()
  finally:
    engine_mod.set_default_batch(None)
  board, reward, discount = game.its_showtime()
  assert isinstance(game.fused, shapes.ShapeGame) and reward is None
  assert _spec_bytes(game.fused.spec) == _golden('hello_world_spec')['spec'].tobytes()
  assert _same(board.board.cpu().numpy(), gold['board'][0])
  for t in range(T):
    board, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert _same(board.board.cpu().numpy(), gold['board'][t + 1]), t
    assert _same(board.layered_board.cpu().numpy(), gold['layered'][t + 1].astype(np.int8)), t
    assert _same(reward.cpu().numpy(), gold['reward'][t]), t
    assert _same(discount.cpu().numpy(), gold['discount'][t]), t
  # `game.play(0)` as in cell 6: one integer for every environment
  board, reward, discount = game.play(0)
  assert bool((reward == 1.0).all())
