"""A seeded family of 16 games whose SCENERY changes (tests/random_pickups.py) against what the
REFERENCE's engine, renderer and Plot did with the very same classes
(tests/golden/random_pickups.npz, make_random_golden.py pickups): drapes that cover SEVERAL cells
which come and go - coins taken one by one, coins that come back, ice that breaks behind the walker
(tests/traced_games.py's Coins / ReturningCoins / ThinIce; the reference's Drape has no one-cell
limit, campx/things.py:161-262) - and a `Backdrop.update()` that repaints the scenery (`Lamps`,
`Tide`, `Seasons`; campx/things.py:103-148): the two things VERDICT r5 named as what the batched
tiers refuse that PyColab games do.

The tabulator describes such cells as PIECES (`TracedGame.piece_cell` / `in_backdrop`: one per cell
a drape ever covers, one per (cell, character) a Backdrop shows beyond its first picture) and hands
them to the kernels in one of three ways, all three held against the reference's frames here:
  * up to sixteen pieces beside an ordinary mover: a MASK of the pieces that show, one 16-bit value
    per state (`CampxWideSpec.n_pieces`; the render kernel patches them from one trace entry a row);
  * more cells than that - a switch turns the WHOLE floor: `Tide`, `Seasons` - as VARIANTS of the
    scenery the state names (`TracedGame.variants`, `CampxWideSpec.n_variants`).
`ROUTES` below runs some games down the road they would not take by themselves (the bounds
lowered), so that every road sees drapes and Backdrops both.

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
# (game, road): every game down its own road, some down the others
ROUTES = ([(k, 'own') for k in range(len(DEFS))] + [(k, 'variants') for k in (3, 5, 10)] +
          [(k, 'things') for k in (3, 4, 9, 10)] + [(k, 'dense') for k in (0, 1, 7)])
ROUTE_IDS = ['{}-{}'.format(IDS[k], road) for k, road in ROUTES]


def _road(monkeypatch, road):
  """The bounds that decide how pieces reach the kernels, lowered: 'variants' = no mask;
  'things' = no mask and one picture only (round 6's first form: a tracked thing per piece);
  'dense' = a walker and two coins stay three things of the cell-indexed tables."""
  from campx_amd import gamespec
  if road == 'dense':            # three tracked things: pieces as things of the one-cell tier's tables
    monkeypatch.setattr(gamespec, 'PIECES_AS_THINGS_MAX', 3)
  if road in ('variants', 'things'):
    monkeypatch.setattr(gamespec, 'WIDE_MAX_PIECES', 0)
  if road == 'things':
    monkeypatch.setattr(gamespec, 'WIDE_MAX_VARIANTS', 1)


def _traced(k, road='own'):
  """Game k tabulated (once per test process: the one-frame-per-play walker takes 1-4 s a game);
  call under `_road()` for a road that is not the game's own."""
  if (k, road) not in _TRACED:
    _TRACED[k, road] = tabulate.trace(random_pickups.builder(DEFS[k])(), cache=False)
  return _TRACED[k, road]


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f' or b.dtype.kind == 'f':
    return np.array_equal(a.astype(np.float32), b.astype(np.float32), equal_nan=True)
  return np.array_equal(a, b)


def test_the_generator_still_makes_the_games_of_the_fixture():
  assert len(DEFS) == random_pickups.N_GAMES == 16
  for k, d in enumerate(DEFS):
    gold = _gold(k)
    assert [''.join(chr(c) for c in row) for row in gold['art']] == d['art'], k
    assert json.loads(str(gold['meta'])) == dict(kind=d['kind']), k


def test_how_the_pieces_of_each_game_reach_the_kernels():
  import ctypes
  from campx_amd import _hip
  kinds = {}
  for k, d in enumerate(DEFS):
    traced = _traced(k)
    W = len(d['art'][0])
    where = lambda ch: [r * W + c for r, row in enumerate(d['art']) for c, x in enumerate(row) if x == ch]
    ch = {'ice': '~', 'coins': 'o', 'returning': 'o'}.get(d['kind'])
    if ch:
      # one piece per cell the drape ever covers, all of its character
      n = len(where(ch))
      assert traced.movers == ['A'] + [ch] * n and traced.piece_cell == [None] + where(ch), (k, traced.movers)
      assert len(traced.variants) == 1 and not any(traced.in_backdrop)
    elif d['kind'] == 'lamps':
      # one piece per (cell, character) the Backdrop shows beyond its first picture
      lamps = where(':') + where('*')
      assert traced.movers[0] == 'A' and traced.in_backdrop == [False] + [True] * len(lamps)
      assert sorted(traced.piece_cell[1:]) == sorted(lamps) and len(traced.variants) == 1
    else:                        # the whole floor turns: the Backdrop's own pictures
      assert traced.movers == ['A'] and traced.piece_cell == [None] and not traced.pieces_as_mask
      V = len(traced.variants)
      assert V == {'tide': 2, 'seasons': 3}[d['kind']] and set(traced.st_variant.tolist()) == set(range(V))
      assert all(m == {} for m in traced.variant_masks)
      spec, arrays = tabulate.to_wide_spec(traced)
      assert (spec.n_dyn, spec.n_pieces, spec.n_variants) == (1, 0, V)
      assert arrays['variant_top_layer'].shape == (V, len(d['art']) * W)
      assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == 0
      arrays['state_variant'][min(3, traced.n_states - 1)] = V          # a variant that is not there
      assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == -2
      kinds[d['kind']] = kinds.get(d['kind'], 0) + 1
      continue
    # a mask: the walker is the kernels' one thing, the pieces are bits
    assert traced.pieces_as_mask and traced.dense_reason
    spec, arrays = tabulate.to_wide_spec(traced)
    P = len(traced.movers) - 1
    assert (spec.n_dyn, spec.n_pieces, spec.n_variants) == (1, P, 0) and arrays['state_cells'].shape[1] == 1
    assert list(spec.piece_cell[:P]) == traced.piece_cell[1:]
    assert [traced.chars[i] for i in spec.piece_layer[:P]] == traced.movers[1:]
    shown = arrays['state_pieces']
    assert shown.dtype == np.uint16 and int(shown.max()) < (1 << P)
    for p in range(P):
      assert np.array_equal((shown >> p) & 1, traced.st_shows[:, 1 + p])
    assert len(set(shown.tolist())) > P                                   # (many pictures, one plane)
    if k == 15:                  # nine tiles of ice: the mask's second byte
      assert P == 9 and int(shown.max()) == 0x1ff and traced.n_states == 1097
    assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == 0
    shown[min(3, traced.n_states - 1)] = 1 << P                         # a piece that is not there
    assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == -2
    shown[min(3, traced.n_states - 1)] = 0
    spec.piece_cell[0] = len(d['art']) * W                                # off the board
    assert _hip.lib.campx_wide_spec_validate(ctypes.byref(spec)) == -2
    kinds[d['kind']] = kinds.get(d['kind'], 0) + 1
  assert kinds == {'ice': 4, 'coins': 3, 'returning': 3, 'lamps': 3, 'tide': 2, 'seasons': 1}, kinds


def test_past_the_mask_the_variants_then_things_then_a_refusal(monkeypatch):
  """More pieces than the mask has bits (here: the bound lowered to none): the scenery's pictures
  as variants, the several-cell drapes' curtains part of them; more pictures than the kernels hold
  sets of rows for (the bound lowered to one): a tracked thing per piece, up to eight."""
  _road(monkeypatch, 'dense')      # a walker and two coins: three things of the cell-indexed tables
  few = _traced(0, 'dense')
  assert few.movers == ['A', 'o', 'o'] and few.dense_reason is None and not few.pieces_as_mask
  _road(monkeypatch, 'variants')
  coins = _traced(3, 'variants')                                             # seven coins: 51 pictures
  assert coins.movers == ['A'] and coins.piece_cell == [None] and not coins.pieces_as_mask
  V = len(coins.variants)
  assert 7 < V <= 2 ** 7 and set(coins.st_variant.tolist()) == set(range(V))
  assert all(sorted(m) == ['o'] for m in coins.variant_masks)
  assert all((v == coins.variants[0]).all() for v in coins.variants)         # (the Backdrop stays)
  assert not any(name == 'o' for name, _ in coins.statics)
  lamps = _traced(10, 'variants')                                            # four lamps: 16 pictures
  assert lamps.movers == ['A'] and len(lamps.variants) == 16 and all(m == {} for m in lamps.variant_masks)
  _road(monkeypatch, 'things')
  lamps = _traced(10, 'things')
  assert lamps.movers[0] == 'A' and lamps.in_backdrop == [False] + [True] * 4 and len(lamps.variants) == 1
  assert not lamps.pieces_as_mask and tabulate.to_wide_spec(lamps)[0].n_dyn == 5
  coins = _traced(3, 'things')
  assert coins.movers == ['A'] + ['o'] * 7 and len(coins.variants) == 1 and not coins.pieces_as_mask
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


@pytest.mark.parametrize('k,road', ROUTES, ids=ROUTE_IDS)
def test_the_table_tabulated_from_the_classes_gives_them_too(k, road, monkeypatch):
  from oracle.table_replay import StateWalker, TableWalker
  gold = _gold(k)
  T, N = gold['actions'].shape
  _road(monkeypatch, road)
  traced = _traced(k, road)
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
@pytest.mark.parametrize('k,road', ROUTES, ids=ROUTE_IDS)
def test_hip_path_gives_the_reference_engines_frames(k, road, monkeypatch):
  gold = _gold(k)
  T, N = gold['actions'].shape
  _road(monkeypatch, road)
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
  traced = _traced(k, road)
  for_road = {'own': None, 'variants': lambda f: f.spec.n_variants > 1, 'things': lambda f: f.spec.n_dyn > 1,
              'dense': lambda f: not hasattr(f, 'spec') or not hasattr(f.spec, 'n_pieces')}[road]
  acts = np.random.RandomState(40 + k).randint(0, 5, size=(60, B)).astype(np.int8)
  game = build(batch=B, device='cuda')
  game.its_showtime()
  assert for_road is None or for_road(game.fused), road
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
