"""Random SEQUENCES of calls on one batched engine - play(), rollout() with and without a reset,
with a flat board, keeping only the last frame, 16-bit observations, reused buffers, the two-stream
form, deferred rollouts over own and shared observation buffers, flush() - against the C oracle
fed the same actions in the same order (state carried from call to call: positions, game-over
flags, episode returns; an environment whose episode ended is rebuilt before its next action).
Every tier: one-cell (one, two, three movers; batches on both sides of the one-launch rollout's
bound), state table, shapes.  Seeded."""

import os

import numpy as np
import pytest
import torch

from campx_amd import gamespec
from campx_amd.games import boat_race, hello_world, maze, sokoban, wall_world
from oracle import cpu

pytestmark = pytest.mark.gpu


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f' or b.dtype.kind == 'f':
    return np.array_equal(a.astype(np.float32).view(np.uint32), b.astype(np.float32).view(np.uint32)) or \
        np.array_equal(a.astype(np.float32), b.astype(np.float32), equal_nan=True)
  return np.array_equal(a, b)


GAMES = {
    'boat_race': (boat_race.build, {}, 5),
    'wall_world': (wall_world.build, {}, 5),
    'sokoban': (sokoban.build, {}, 5),
    'sokoban_l1': (sokoban.build, dict(level=1), 5),
    'maze_16x16': (lambda **kw: maze.build(16, 16, **kw), {}, 5),
    'hello_world': (hello_world.build, {}, 5),
}
# more of the shape tier (eight things and quits; a 16x24 board with a static drape; two drapes on
# 13x36 from the random family) and a two-box sokoban on 16x16 (state table enumerated on the device)
import shape_zoo  # noqa: E402
import random_hellos  # noqa: E402
GAMES['zoo1'] = (shape_zoo.library_builders()['shape_zoo1'], {}, 5)
GAMES['zoo3'] = (shape_zoo.library_builders()['shape_zoo3'], {}, 5)
GAMES['random_hello3'] = (random_hellos.library_builder(random_hellos.definitions()[3]), {}, 5)
GAMES['sokoban_l3'] = (sokoban.build, dict(level=3), 5)


class TracedOracle(object):
  """A host-tabulated game's STATE table walked on the host (oracle/table_replay.py StateWalker),
  dressed as `oracle.cpu.OracleGame` for this fuzz: the checker of games made of arbitrary Python
  classes, which have no rule description for the C oracle.  (That the table is right is pinned
  elsewhere: tests/test_random_pickups.py, against the reference engine's frames.)"""

  def __init__(self, traced):
    self.traced, self.walker = traced, None

  def _frames(self, states):
    from oracle.table_replay import StateWalker
    board, layered = StateWalker(self.traced, 1).render(np.asarray(states).reshape(-1))
    return layered, board

  def first_frame(self):
    layered, board = self._frames([0])
    return layered[0], board[0]

  def rollout(self, actions, reset_first=False):
    from oracle.table_replay import StateWalker
    T, B = actions.shape
    if self.walker is None:
      self.walker = StateWalker(self.traced, B)
    want = self.walker.rollout(actions, reset_first=reset_first)
    layered, board = self._frames(want['state'])
    g = self.traced
    return dict(obs=layered.reshape(T, B, len(g.chars), g.rows, g.cols), board=board.reshape(T, B, g.rows, g.cols),
                reward=want['reward'], discount=want['discount'], done=want['done'], perf=None)


# games of arbitrary Python classes whose drapes cover several cells that come and go, or whose
# Backdrop changes (round 6, tests/random_pickups.py): seven coins (state table), coins that come
# back (three tracked things: the cell-indexed tables), thin ice with a hidden Plot entry, floor lamps
import random_pickups  # noqa: E402
from campx_amd import tabulate  # noqa: E402
for _k in (1, 3, 5, 10, 13, 14):      # (13, 14: a scenery of three / two VARIANTS - a whole floor that turns)
  _d = random_pickups.definitions()[_k]
  GAMES['pickup%d_%s' % (_k, _d['kind'])] = (random_pickups.builder(_d), {}, 5, 'traced')


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_SEQ_SEEDS', '6'))))
@pytest.mark.parametrize('name', sorted(GAMES))
def test_a_random_sequence_of_calls_matches_the_oracle(name, seed):
  build, kw, n_actions = GAMES[name][:3]
  rng = np.random.RandomState(1000 * seed + len(name))
  B = int(rng.choice([7, 8, 64, 1000, 1002, 4096, 9000, 9008]))
  game = build(batch=B, device='cuda', **kw)
  first, _, _ = game.its_showtime()
  f = game.fused
  one_cell = type(f).__name__ == 'FusedGame'
  if len(GAMES[name]) > 3:
    og = TracedOracle(tabulate.trace(build(**kw)))
  else:
    og = cpu.OracleGame.from_description(gamespec.describe(build(**kw)))
  obs0, board0 = og.first_frame()
  assert np.array_equal(first.layered_board.cpu().numpy()[0], obs0.astype(np.int8))
  kept = {}                      # T -> buffers reused across calls
  fresh = True                   # nothing played yet: the first action needs no reset either way
  log = []

  def actions(T):
    return rng.randint(0, n_actions, size=(T, B)).astype(np.int8)

  # the episode return every environment carries (CampxState.ret: what bench.py's return log
  # gathers), accumulated here from the oracle's rewards: restarted by the frame that follows an
  # episode's end, NaN (the reference's None) counted as 0
  carried = dict(ret=np.zeros(B, np.float32), over=np.ones(B, bool))

  oracle_rollout = og.rollout

  def accounted(actions, reset_first=False):
    if reset_first:
      carried['over'][:] = True
    ref = oracle_rollout(actions, reset_first=reset_first)
    pending.append(ref)
    return ref

  pending = []
  og.rollout = accounted

  def check(out, ref, what, obs=True, board=False, last_only=False):
    log.append(what)
    if out.get('perf') is not None and ref.get('perf') is not None:
      assert _same(out['perf'].cpu().numpy(), ref['perf']), (log, 'perf')
    if obs and not last_only:
      assert _same(out['obs'].cpu().numpy(), ref['obs']), log
    if last_only:
      assert _same(out['obs'].cpu().numpy().reshape(ref['obs'][-1].shape), ref['obs'][-1]), log
    if board:
      assert _same(out['board'].cpu().numpy(), ref['board']), log
    for k in ('reward', 'discount', 'done'):
      assert _same(out[k].cpu().numpy(), ref[k]), (log, k)

  def settle():
    torch.cuda.synchronize()
    while pending:
      ref = pending.pop(0)
      last = not pending
      for t in range(ref['reward'].shape[0]):
        r = np.where(np.isnan(ref['reward'][t]), np.float32(0), ref['reward'][t]).astype(np.float32)
        carried['ret'] = (np.where(carried['over'], np.float32(0), carried['ret']) + r).astype(np.float32)
        carried['over'] = ref['done'][t] != 0
      if last and hasattr(f, 'ret') and f.ret is not None:
        assert _same(f.ret.cpu().numpy(), carried['ret']), (log, 'episode returns')
        assert np.array_equal(f.done.cpu().numpy() != 0, carried['over']), (log, 'game-over flags')

  for step in range(14):
    settle()
    op = rng.choice(['play', 'play', 'play-forms', 'rollout', 'rollout', 'rollout-reset', 'rollout-board', 'rollout-out',
                     'rollout-last', 'rollout-f16', 'pipelined', 'deferred', 'deferred-shared'])
    if op == 'play':
      for _ in range(int(rng.randint(1, 4))):
        a = actions(1)
        obs, reward, discount = game.play(torch.from_numpy(a[0]))
        ref = og.rollout(a, reset_first=False)
        log.append('play')
        assert _same(obs.layered_board.cpu().numpy(), ref['obs'][0]), log
        assert _same(obs.board.cpu().numpy(), ref['board'][0]), log
        assert _same(reward.cpu().numpy(), ref['reward'][0]), log
        assert _same(discount.cpu().numpy(), ref['discount'][0]), log
      continue
    if op == 'play-forms':
      # the other forms an action may take: one-hot floats [B, 5] (the reference's format,
      # examples/boat_race.py:154-184), int64 ids, a Python list of ids, ONE one-hot vector or ONE
      # id for every environment, and - unchecked - ids outside 0..4, which act as "stay"
      form = rng.choice(['onehot', 'int64', 'list', 'one-vector', 'one-id', 'outside'])
      if type(f).__name__ == 'ShapeGame' and form in ('onehot', 'one-vector'):
        form = 'int64'           # (the Hello World kind takes integer ids, as its notebook passes them)
      a = actions(1)
      sent = torch.from_numpy(a[0])
      if form == 'onehot':
        sent = torch.eye(5)[torch.from_numpy(a[0]).long()]
      elif form == 'int64':
        sent = torch.from_numpy(a[0]).long().cuda()
      elif form == 'list':
        sent = [int(x) for x in a[0]]
        if B == 5:
          sent = torch.tensor(sent)
      elif form == 'one-vector':
        a[0, :] = a[0, 0]
        sent = torch.eye(5)[int(a[0, 0])]
      elif form == 'one-id':
        a[0, :] = a[0, 0]
        sent = int(a[0, 0])
      else:
        if n_actions != 5 or not one_cell:
          continue
        raw = a[0].copy()
        raw[rng.rand(B) < 0.3] = int(rng.choice([5, 9, 100, -1, -128]))
        a[0] = np.where((raw < 0) | (raw > 4), 4, raw)
        sent = torch.from_numpy(raw)
        f.validate_actions = False
      obs, reward, discount = game.play(sent)
      f.validate_actions = True
      if form == 'outside':            # (counted: the next look would raise; take the count back)
        torch.cuda.synchronize()
        f._bad.zero_()
        if hasattr(f, '_bad_flag') and f._bad_flag is not None:
          f._bad_flag.zero_()
      ref = og.rollout(a, reset_first=False)
      log.append('play ' + form)
      assert _same(obs.layered_board.cpu().numpy(), ref['obs'][0]), log
      assert _same(reward.cpu().numpy(), ref['reward'][0]), log
      assert _same(discount.cpu().numpy(), ref['discount'][0]), log
      continue
    T = int(rng.choice([1, 5, 16, 23, 40]))
    a = actions(T)
    dev = torch.from_numpy(a).cuda()
    if op in ('rollout', 'rollout-reset', 'rollout-board'):
      reset = op == 'rollout-reset'
      out = game.rollout(dev, reset_first=reset, want_board=(op == 'rollout-board'))
      check(out, og.rollout(a, reset_first=reset), '%s T=%d' % (op, T), board=(op == 'rollout-board'))
    elif op == 'rollout-out':
      out = kept.setdefault(T, game.rollout_buffers(T))
      got = game.rollout(dev, out=out)
      assert got['obs'] is out['obs']
      check(out, og.rollout(a, reset_first=False), 'rollout out= T=%d' % T)
    elif op == 'rollout-last':
      out = game.rollout(dev, keep_obs=False)
      check(out, og.rollout(a, reset_first=False), 'rollout keep_obs=False T=%d' % T, obs=False, last_only=True)
    elif op == 'rollout-f16':
      dt = torch.float16 if rng.rand() < 0.5 else torch.bfloat16
      out = game.rollout(dev, obs_dtype=dt)
      assert out['obs'].dtype == dt
      ref = og.rollout(a, reset_first=False)
      log.append('rollout %s T=%d' % (dt, T))
      assert _same(out['obs'].float().cpu().numpy(), ref['obs'].astype(np.float32)), log
      assert _same(out['reward'].cpu().numpy(), ref['reward']), log
    elif op == 'pipelined':
      if not one_cell:
        continue
      bufs = [f.rollout_buffers(T), f.rollout_buffers(T)]
      refs = []
      streams = [actions(T) for _ in range(3)]
      for i, s in enumerate(streams):
        f.rollout(torch.from_numpy(s).cuda(), out=bufs[i & 1], pipelined=True)
        refs.append(og.rollout(s, reset_first=False))
        if i >= 1:                # (a buffer set is the caller's again once the NEXT call was issued)
          check(bufs[(i - 1) & 1], refs[i - 1], 'pipelined %d T=%d' % (i - 1, T))
      torch.cuda.synchronize()
      check(bufs[(len(streams) - 1) & 1], refs[-1], 'pipelined last T=%d' % T)
    else:
      shared = op == 'deferred-shared' and one_cell
      one = game.rollout_buffers(T)
      bufs = [one, game.rollout_buffers(T, share=one) if shared else game.rollout_buffers(T)]
      refs = []
      n = int(rng.randint(1, 5))
      for i in range(n):
        s = actions(T)
        prev = game.rollout_deferred(torch.from_numpy(s).cuda(), bufs[i & 1])
        refs.append(og.rollout(s, reset_first=False))
        assert (prev is None) == (i == 0), log
        assert _same(bufs[i & 1]['reward'].cpu().numpy(), refs[i]['reward']), (log, 'deferred scalars', i)
        if prev is not None:
          assert prev is bufs[(i - 1) & 1]
          check(prev, refs[i - 1], 'deferred%s %d/%d T=%d' % (' shared' if shared else '', i - 1, n, T))
      if rng.rand() < 0.5 or not one_cell:
        check(game.flush(), refs[-1], 'flush T=%d' % T)
      else:
        # not flushed: the next call of any kind must leave the owed observations right, or
        # render them first - play() reads and writes the engine's own frame buffer only
        a1 = actions(1)
        obs, reward, _ = game.play(torch.from_numpy(a1[0]))
        ref1 = og.rollout(a1, reset_first=False)
        log.append('play after deferred')
        assert _same(obs.layered_board.cpu().numpy(), ref1['obs'][0]), log
        check(game.flush(), refs[-1], 'late flush T=%d' % T)

  settle()
  assert not pending and len(log) >= 10

  # ... and the per-frame hand-off in the policy network's dtype (one-cell tier, table games)
  if one_cell and f.uses_table:
    dt = torch.float16 if rng.rand() < 0.5 else torch.bfloat16
    f.set_play_obs_dtype(dt)
    for _ in range(3):
      a = actions(1)
      obs, reward, _ = game.play(torch.from_numpy(a[0]))
      ref = og.rollout(a, reset_first=False)
      log.append('play %s' % dt)
      assert obs.layered_board.dtype == dt
      assert _same(obs.layered_board.float().cpu().numpy(), ref['obs'][0].astype(np.float32)), log
      assert _same(reward.cpu().numpy(), ref['reward'][0]), log
    f.set_play_obs_dtype(torch.int8)
    a = actions(1)
    obs, _, _ = game.play(torch.from_numpy(a[0]))
    assert _same(obs.layered_board.cpu().numpy(), og.rollout(a, reset_first=False)['obs'][0]), log


def test_a_rollout_of_no_frames_is_refused_in_words():
  for name in ('boat_race', 'sokoban_l1', 'maze_16x16', 'hello_world'):
    build, kw = GAMES[name][:2]
    game = build(batch=16, device='cuda', **kw)
    game.its_showtime()
    with pytest.raises(ValueError, match='at least one frame'):
      game.rollout(torch.zeros((0, 16), dtype=torch.int8))
