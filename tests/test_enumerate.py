"""Multi-mover rule games above 128 cells: the state table is enumerated ON THE DEVICE by the
rules themselves (campx_amd/enumerate_states.py, csrc/k_wide.hip wide_enumerate_kernel) and run
by the wide tier.  The case: the build's sokoban rules on a 16x16 board with two boxes
(games/sokoban.py level 3; campx/engine.py:31 sets no board size limit) - 4.4 million reachable
states, out of reach of the host tabulator's frame of Python per state and action.

CPU: the wide lowering equals the one-cell lowering where both exist; `sokoban_l3.npz` (the
REFERENCE engine running the reference's own AgentDrape with the build's Box / Goal rules,
make_golden.py) is reproduced by this repo's generic tier and by the C oracle at 256 cells.
GPU: its_showtime() under 10 s; B = 65 536 bit-exact against the oracle and the golden; the
enumerate kernel against the rule interpreter kernel on a board both can take.
"""

import ctypes
import os
import time

import numpy as np
import pytest
import torch

from campx_amd import gamespec
from campx_amd.games import sokoban
from conftest import GOLDEN_DIR
from oracle import cpu


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f':
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
  return np.array_equal(a, b)


def _golden():
  with np.load(os.path.join(GOLDEN_DIR, 'sokoban_l3.npz')) as f:
    return {k: f[k] for k in f.files}


# ------------------------------------------------------------------------------- CPU

@pytest.mark.parametrize('level', [0, 1, 2])
def test_wide_lowering_is_the_one_cell_lowering(level):
  desc = gamespec.describe(sokoban.build(level=level))
  narrow, wide = gamespec.lower(desc), gamespec.lower(desc, wide=True)
  HW = desc.rows * desc.cols
  assert ctypes.string_at(ctypes.addressof(narrow.rules), ctypes.sizeof(narrow.rules)) == \
      ctypes.string_at(ctypes.addressof(wide.rules), ctypes.sizeof(wide.rules))
  for name in ('static_top_layer', 'static_top_z', 'static_cover', 'cell_class'):
    assert np.array_equal(np.array(getattr(narrow, name)[:HW]), getattr(wide, name)), name
  for name in ('n_dyn', 'n_rules', 'any_reward', 'perf_dyn', 'perf_mode', 'perf_mask', 'perf_scale',
               'perf_offset'):
    assert getattr(narrow, name) == getattr(wide, name), name
  assert list(narrow.dyn_z[:wide.n_dyn]) == wide.dyn_z[:wide.n_dyn].tolist()


def test_a_board_above_1024_cells_is_refused_by_the_wide_lowering():
  from campx_amd import rules
  from campx_amd.ascii_art import ascii_art_to_game, Partial
  art = ['#' * 40] + ['#A' + ' ' * 37 + '#'] + ['#' + ' ' * 38 + '#'] * 24 + ['#' * 40]
  game = ascii_art_to_game(art, what_lies_beneath=' ',
                           drapes={'#': rules.FixedDrape, 'A': Partial(rules.AgentDrape, blocking_chars='#')},
                           z_order='A#', update_schedule='A#')
  with pytest.raises(ValueError, match='more than 1024 cells'):
    gamespec.lower(gamespec.describe(game), wide=True)


def test_level_3_golden_on_the_generic_tier_and_the_oracle():
  """What the REFERENCE engine did with the 16x16 level (its own AgentDrape, the build's
  Box / Goal rules): this repo's generic tier for three environments, the C oracle for all."""
  gold = _golden()
  T, N = gold['actions'].shape
  assert gold['board'].shape[-2:] == (16, 16) and gold['done'].sum() >= 1
  onehot = [torch.eye(5)[a] for a in range(5)]
  for n in range(3):
    game = sokoban.build(level=3)
    obs, _, _ = game.its_showtime()
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      if game.game_over:
        game = sokoban.build(level=3)
        game.its_showtime()
      obs, reward, discount = game.play(onehot[int(gold['actions'][t, n])])
      assert np.array_equal(obs.board.numpy(), gold['board'][t + 1, n].astype(np.uint8)), (n, t)
      assert _same(np.float32(float(reward)), gold['reward'][t, n])
      assert int(game.game_over) == gold['done'][t, n]
  og = cpu.OracleGame.from_description(gamespec.describe(sokoban.build(level=3)))
  ref = og.rollout(gold['actions'], reset_first=True)
  assert _same(ref['obs'], gold['layered'][1:].astype(np.int8))
  assert _same(ref['board'], gold['board'][1:])
  for k in ('reward', 'discount', 'done'):
    assert _same(ref[k], gold[k]), k


# ------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_sokoban_16x16_is_enumerated_on_the_device_and_matches_the_golden():
  from campx_amd import wide
  gold = _golden()
  T, N = gold['actions'].shape
  # (a fresh box pages torch's sort / unique / searchsorted code objects in on first use - a
  # minute, once per process, nothing to do with the enumeration: touch them before the clock)
  warm = torch.randint(0, 1000, (4096,), device='cuda')
  torch.searchsorted(torch.sort(torch.unique(warm)).values, warm)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  game = sokoban.build(batch=N, device='cuda', level=3)
  first, reward0, _ = game.its_showtime()
  torch.cuda.synchronize()
  took = time.perf_counter() - t0
  f = game.fused
  assert isinstance(f, wide.WideGame) and reward0 is None
  assert f.traced.n_states > 1 << 20 and f.traced.movers == ['X', 'Y', 'A']
  assert took < 10.0, took
  assert _same(first.board.cpu().numpy(), gold['board'][0])
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert _same(out['obs'].cpu().numpy(), gold['layered'][1:].astype(np.int8))
  assert _same(out['board'].cpu().numpy(), gold['board'][1:])
  for k in ('reward', 'discount', 'done'):
    assert _same(out[k].cpu().numpy(), gold[k]), k
  game = sokoban.build(batch=N, device='cuda', level=3)
  game.its_showtime()
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert _same(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert _same(reward.cpu().numpy(), gold['reward'][t]), t
    assert _same(discount.cpu().numpy(), gold['discount'][t]), t


@pytest.mark.gpu
def test_sokoban_16x16_at_full_batch_against_the_oracle():
  """B = 65 536: every byte of a 24-frame launch and of a second one (state carried over), the
  hidden side-effects penalty included; then a 100-frame launch, every scalar of every
  environment against the oracle's and the observations of every 16th environment byte for
  byte (a second oracle instance that follows those 4 096 environments through all three
  launches; the whole launch's observations would be 10 GB on the host)."""
  B = 65536
  game = sokoban.build(batch=B, device='cuda', level=3)
  game.its_showtime()
  desc = gamespec.describe(sokoban.build(level=3))
  og, og_sub = cpu.OracleGame.from_description(desc), cpu.OracleGame.from_description(desc)
  rng = np.random.RandomState(31)
  sub = np.arange(0, B, 16)
  for launch, T in enumerate([24, 16]):
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    out = game.rollout(torch.from_numpy(actions), want_board=True)
    ref = og.rollout(actions, reset_first=(launch == 0))
    og_sub.rollout(np.ascontiguousarray(actions[:, sub]), reset_first=(launch == 0), keep_obs=False)
    for k in ('obs', 'board', 'reward', 'discount', 'done', 'perf'):
      assert _same(out[k].cpu().numpy(), ref[k]), (launch, k)
    del out, ref
  actions = rng.randint(0, 5, size=(100, B)).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions))
  ref = og.rollout(actions, keep_obs=False, want_board=False)         # scalars of everybody
  for k in ('reward', 'discount', 'done', 'perf'):
    assert _same(out[k].cpu().numpy(), ref[k]), k
  ref = og_sub.rollout(np.ascontiguousarray(actions[:, sub]), want_board=False)
  assert _same(out['obs'][:, torch.from_numpy(sub).cuda()].cpu().numpy(), ref['obs'])
  assert int(out['obs'].sum(dim=2, dtype=torch.int32).min()) == 1     # one character per cell


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['boat_race', 'wall_world', 'demo3', 'demo4'])
def test_enumerate_kernel_on_the_other_rule_kinds(name):
  """Directional hover rewards, reward characters, the hidden-performance cycle: one-mover
  library games (never enumerated in production: the host tabulator takes them) put through
  the enumerate kernel and the wide tier, against the one-cell tier on the same actions."""
  from campx_amd import enumerate_states, wide
  from games_under_test import FUSED_GAMES
  build = FUSED_GAMES[name]
  B, T = 1024, 150
  a_game = build(batch=B, device='cuda')
  a_game.its_showtime()
  traced = enumerate_states.enumerate_rule_game(build(), 'cuda')
  assert traced.n_states >= 4
  b = wide.WideGame(build(batch=B, device='cuda'), B, 'cuda', traced)
  b.showtime()
  actions = torch.from_numpy(np.random.RandomState(7).randint(0, 5, size=(T, B)).astype(np.int8))
  x = a_game.rollout(actions, want_board=True)
  y = b.rollout(actions, want_board=True)
  for k in ('obs', 'board', 'reward', 'discount', 'done', 'perf'):
    if x[k] is None:
      assert y[k] is None or k == 'perf'
      continue
    assert _same(x[k].cpu().numpy(), y[k].cpu().numpy()), k


@pytest.mark.gpu
@pytest.mark.parametrize('level', [0, 1, 2])
def test_enumerate_kernel_against_the_rule_interpreter(level):
  """A board both tiers can take: the state table the enumerate kernel finds, run by the wide
  tier, against the one-cell tier (tables built by the rule interpreter kernel) on the same
  actions - two kernels that restate the rules, neither of them the oracle."""
  from campx_amd import enumerate_states, fused, wide
  B, T = 2048, 120
  a_game = sokoban.build(batch=B, device='cuda', level=level)
  a_game.its_showtime()
  assert type(a_game.fused) is fused.FusedGame and a_game.fused.traced is None
  traced = enumerate_states.enumerate_rule_game(sokoban.build(level=level), 'cuda')
  b_engine = sokoban.build(batch=B, device='cuda', level=level)
  b = wide.WideGame(b_engine, B, 'cuda', traced)
  b.showtime()
  actions = torch.from_numpy(np.random.RandomState(level).randint(0, 5, size=(T, B)).astype(np.int8))
  x = a_game.rollout(actions, want_board=True)
  y = b.rollout(actions, want_board=True)
  for k in ('obs', 'board', 'reward', 'discount', 'done', 'perf'):
    assert _same(x[k].cpu().numpy(), y[k].cpu().numpy()), k
  assert int(x['done'].sum()) > 0


def test_enumerate_entry_point_checks_its_arguments_without_a_gpu():
  """campx_wide_enumerate_launch() refuses bad arguments before it touches the device (the
  reference raises ValueError / RuntimeError for malformed set-up: engine.py:47-53, 332-350)."""
  from campx_amd import _hip
  r = gamespec.CampxWideRules()
  fake = ctypes.c_void_p(16)                   # never dereferenced: every call below is refused
  call = lambda rules, n=1: _hip.lib.campx_wide_enumerate_launch(
      ctypes.byref(rules), fake, n, fake, fake, fake, fake, None, None)
  assert ctypes.sizeof(r) == _hip.lib.campx_wide_rules_size()
  assert call(r) != 0                          # no magic
  r.magic, r.version = gamespec.SPEC_MAGIC, gamespec.SPEC_VERSION
  r.rows, r.cols, r.n_dyn = 16, 16, 3
  assert call(r) != 0                          # no scenery tables
  r.top_layer = r.top_z = r.cover = r.cell_class = 16
  r.rows = 200
  assert call(r) != 0                          # more than 127 rows
  r.rows, r.n_dyn = 16, 9
  assert call(r) != 0                          # more than four moving things
  r.n_dyn, r.n_rules = 3, 1
  r.rules[0].op, r.rules[0].dyn = gamespec.OP_BOX, 7
  assert call(r) != 0                          # a rule about a thing that does not exist
  r.rules[0].dyn, r.rules[0].aux = 0, 1
  assert call(r, n=0) == 0                     # nothing to do: accepted, nothing launched
  assert call(r, n=-1) != 0
