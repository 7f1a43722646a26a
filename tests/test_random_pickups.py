"""A seeded family of 9 games whose drapes cover SEVERAL cells that come and go
(tests/random_pickups.py: coins taken one by one, coins that come back, ice that breaks behind the
walker - tests/traced_games.py's Coins / ReturningCoins / ThinIce) against what the REFERENCE's
engine, renderer and Plot did with the very same classes (tests/golden/random_pickups.npz,
make_random_golden.py pickups).  The reference's Drape has no one-cell limit
(campx/things.py:161-262); VERDICT r5 named such drapes as the first thing the batched tiers
refuse that PyColab games do.  The tabulator now tracks one thing per cell the drape ever covers
(`TracedGame.piece_cell`), two to seven of them here, with and without an episode end, with a
hidden Plot entry on top (the ice).  And the second thing the verdict named: a `Backdrop.update()`
that changes the scenery (`Lamps`, campx/things.py:103-148) - every (cell, character) the
backdrop ever shows beyond its first picture is a piece painted behind every thing
(`TracedGame.in_backdrop`), three games of it - and when its pictures differ in more cells than
there are tracked things to spare (a switch turns the WHOLE floor: `Tide`, `Seasons`), the pictures
become variants of the scenery that the state names (`TracedGame.variants`,
`CampxWideSpec.n_variants`: the render kernel lays each environment's own variant); three games.

Per game: (a) the generator still makes the fixture's game; (b) this repo's generic tier gives
the reference engine's frames; (c) so does the table tabulated from the classes, walked on the
host - by cells where the one-cell tier takes the game (three tracked things), by states
otherwise; (d, GPU) so does the HIP path, rollout() and play(), and a rollout of 4 096
environments agrees with the host walker."""

import json
import os

import numpy as np
import pytest
import torch

from campx_amd import tabulate
from conftest import GOLDEN_DIR
import random_pickups

DEFS = random_pickups.definitions()
IDS = ['pickup{}-{}'.format(k, d['kind']) for k, d in enumerate(DEFS)]


def _gold(k):
  with np.load(os.path.join(GOLDEN_DIR, 'random_pickups.npz')) as f:
    pre = 'k{}_'.format(k)
    return {name[len(pre):]: f[name] for name in f.files if name.startswith(pre)}


_TRACED = {}


def _traced(k):
  """Game k tabulated (once per test process: the one-frame-per-play walker takes 1-4 s a game)."""
  if k not in _TRACED:
    _TRACED[k] = tabulate.trace(random_pickups.builder(DEFS[k])(), cache=False)
  return _TRACED[k]


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f' or b.dtype.kind == 'f':
    return np.array_equal(a.astype(np.float32), b.astype(np.float32), equal_nan=True)
  return np.array_equal(a, b)


def test_the_generator_still_makes_the_games_of_the_fixture():
  assert len(DEFS) == random_pickups.N_GAMES == 15
  for k, d in enumerate(DEFS):
    gold = _gold(k)
    assert [''.join(chr(c) for c in row) for row in gold['art']] == d['art'], k
    assert json.loads(str(gold['meta'])) == dict(kind=d['kind']), k


def test_several_cell_drapes_and_changing_backdrops_become_pieces_or_variants():
  """Few pieces - at most three tracked things, none in the Backdrop - stay pieces, on the cell-indexed
  tables; everything else becomes VARIANTS of the scenery (the Backdrop's picture and the several-cell
  drapes' curtains together), one tracked value beside the walker."""
  kinds = {}
  for k, d in enumerate(DEFS):
    traced = _traced(k)
    W = len(d['art'][0])
    where = lambda ch: [r * W + c for r, row in enumerate(d['art']) for c, x in enumerate(row) if x == ch]
    ch = {'ice': '~', 'coins': 'o', 'returning': 'o'}.get(d['kind'])
    n = len(where(ch)) if ch else 0
    if ch and n <= 2 and d['kind'] != 'ice':
      # pieces: one tracked thing per cell the drape ever covers, all of its character
      assert traced.movers == ['A'] + [ch] * n and traced.piece_cell == [None] + where(ch), (k, traced.movers)
      assert len(traced.variants) == 1 and traced.dense_reason is None and not any(traced.in_backdrop)
      kinds['pieces'] = kinds.get('pieces', 0) + 1
      continue
    assert traced.movers == ['A'] and traced.piece_cell == [None], (k, traced.movers)
    V = len(traced.variants)
    assert 2 <= V <= 256 and set(traced.st_variant.tolist()) == set(range(V))
    assert traced.dense_reason.startswith('the scenery changes')
    if ch:                       # the drape's curtains are part of the pictures; the Backdrop stays
      assert all(sorted(m) == [ch] for m in traced.variant_masks)
      assert all((v == traced.variants[0]).all() for v in traced.variants)
      assert traced.variant_masks[0][ch].reshape(-1).nonzero()[0].tolist() == where(ch)
      assert V <= 2 ** n and not any(name == ch for name, _ in traced.statics)
    else:                        # the Backdrop's own pictures
      assert all(m == {} for m in traced.variant_masks)
      assert V == {'tide': 2, 'seasons': 3}.get(d['kind'], V)
      if d['kind'] == 'lamps':
        assert V == 2 ** len(where(':') + where('*'))
    kinds[d['kind']] = kinds.get(d['kind'], 0) + 1
    spec, arrays = tabulate.to_wide_spec(traced)
    assert spec.n_variants == V and arrays['variant_top_layer'].shape == (V, len(d['art']) * W)
    from campx_amd import _hip
    import ctypes
    assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == 0
    arrays['state_variant'][min(3, traced.n_states - 1)] = V          # a variant that is not there
    assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == -2
  assert kinds == {'pieces': 3, 'ice': 3, 'coins': 2, 'returning': 1, 'lamps': 3, 'tide': 2, 'seasons': 1}, kinds


def test_past_the_variants_the_pieces_and_past_the_pieces_a_refusal(monkeypatch):
  """More pictures than the kernels hold sets of rows for (here: the bound lowered to one): pieces
  again - a tracked thing per cell, the Backdrop's painted behind every thing - up to eight."""
  from campx_amd import gamespec
  monkeypatch.setattr(gamespec, 'WIDE_MAX_VARIANTS', 1)
  lamps = tabulate.trace(random_pickups.builder(DEFS[10])(), cache=False)       # four lamps in the Backdrop
  assert lamps.movers[0] == 'A' and lamps.in_backdrop == [False] + [True] * 4 and len(lamps.variants) == 1
  assert sorted(zip(lamps.piece_cell[1:], lamps.movers[1:])) == list(zip(lamps.piece_cell[1:], lamps.movers[1:]))
  coins = tabulate.trace(random_pickups.builder(DEFS[3])(), cache=False)        # seven coins
  assert coins.movers == ['A'] + ['o'] * 7 and len(coins.variants) == 1
  import traced_games as tg
  art = ['##########', '#Aooooooo#', '#o       #', '##########']
  game = tg.ascii_art_to_game(art, what_lies_beneath=' ', drapes={'A': tg.Forager, 'o': tg.Coins, '#': tg.things.FixedDrape},
                              z_order='oA#', update_schedule='Ao#')
  with pytest.raises(tabulate.TabulationError, match=r"'o' cover several cells that come and go - 9 tracked cells"):
    tabulate.trace(game, cache=False)


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_generic_tier_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_pickups.builder(DEFS[k])
  onehot = tabulate.default_actions()
  for n in range(N):
    game = build()
    obs, _, _ = game.its_showtime()
    assert np.array_equal(obs.board.numpy(), gold['board'][0, n].astype(np.uint8))
    for t in range(T):
      if game.game_over:
        game = build()
        game.its_showtime()
      obs, reward, discount = game.play(onehot[int(gold['actions'][t, n])])
      assert np.array_equal(obs.board.numpy(), gold['board'][t + 1, n].astype(np.uint8)), (n, t)
      assert np.array_equal(obs.layered_board.numpy(), gold['layered'][t + 1, n]), (n, t)
      assert _same(np.float32(np.nan if reward is None else float(reward)), gold['reward'][t, n]), (n, t)
      assert np.float32(discount) == gold['discount'][t, n], (n, t)
      assert int(game.game_over) == gold['done'][t, n]


@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_the_table_tabulated_from_the_classes_gives_them_too(k):
  from oracle.table_replay import StateWalker, TableWalker
  gold = _gold(k)
  T, N = gold['actions'].shape
  traced = _traced(k)
  assert [ord(c) for c in traced.chars] == gold['chars'].tolist()
  if traced.dense_reason is not None:
    walker = StateWalker(traced, N)
    want = walker.rollout(gold['actions'], reset_first=True)
    render = lambda t: walker.render(want['state'][t])
  else:
    walker = TableWalker(traced, N)
    want = walker.rollout(gold['actions'], reset_first=True)
    render = lambda t: walker.render(want['cells'][:, t].astype(np.int64))
  for name in ('reward', 'discount', 'done'):
    assert _same(want[name], gold[name]), name
  for t in range(T):
    board, layered = render(t)
    assert np.array_equal(board, gold['board'][t + 1]), t
    assert np.array_equal(layered, gold['layered'][t + 1].astype(np.int8)), t


@pytest.mark.gpu
@pytest.mark.parametrize('k', range(len(DEFS)), ids=IDS)
def test_hip_path_gives_the_reference_engines_frames(k):
  gold = _gold(k)
  T, N = gold['actions'].shape
  build = random_pickups.builder(DEFS[k])
  game = build(batch=N, device='cuda')
  first, _, _ = game.its_showtime()
  assert game.fused is not None and game.fused.traced is not None
  assert np.array_equal(first.board.cpu().numpy(), gold['board'][0])
  assert np.array_equal(first.layered_board.cpu().numpy(), gold['layered'][0].astype(np.int8))
  out = game.rollout(torch.from_numpy(gold['actions']), want_board=True)
  assert np.array_equal(out['obs'].cpu().numpy(), gold['layered'][1:].astype(np.int8))
  assert np.array_equal(out['board'].cpu().numpy(), gold['board'][1:])
  for name in ('reward', 'discount', 'done'):
    assert _same(out[name].cpu().numpy(), gold[name]), name
  game = build(batch=N, device='cuda')
  game.its_showtime()
  for t in range(T):
    obs, reward, discount = game.play(torch.from_numpy(gold['actions'][t]))
    assert np.array_equal(obs.board.cpu().numpy(), gold['board'][t + 1]), t
    assert np.array_equal(obs.layered_board.cpu().numpy(), gold['layered'][t + 1].astype(np.int8)), t
    assert _same(reward.cpu().numpy(), gold['reward'][t]), t
    assert _same(discount.cpu().numpy(), gold['discount'][t]), t
    assert np.array_equal(game.fused.done.cpu().numpy(), gold['done'][t])
  # ... and at a size the kernels' workgroups fill: 4 096 environments against the host's walker
  from oracle.table_replay import StateWalker, TableWalker
  B = 4096
  traced = _traced(k)
  acts = np.random.RandomState(40 + k).randint(0, 5, size=(60, B)).astype(np.int8)
  game = build(batch=B, device='cuda')
  game.its_showtime()
  out = game.rollout(torch.from_numpy(acts), want_board=True)
  if traced.dense_reason is not None:
    walker = StateWalker(traced, B)
    want = walker.rollout(acts, reset_first=True)
    render = lambda t: walker.render(want['state'][t])
  else:
    walker = TableWalker(traced, B)
    want = walker.rollout(acts, reset_first=True)
    render = lambda t: walker.render(want['cells'][:, t].astype(np.int64))
  for name in ('reward', 'discount', 'done'):
    assert _same(out[name].cpu().numpy(), want[name]), name
  for t in (0, 1, 17, 59):
    board, layered = render(t)
    assert np.array_equal(out['board'][t].cpu().numpy(), board), t
    assert np.array_equal(out['obs'][t].cpu().numpy(), layered), t
