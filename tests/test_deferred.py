"""Rollouts pipelined across calls (FusedGame.rollout_deferred / flush: the update pass of one
rollout and the render pass of the one before it in one launch, csrc/k_update.hip
pipe_table_kernel behind campx_update_render_launch).  Every byte of every frame against the C
oracle, over several calls with the state carried from one to the next: batch sizes whose last
update workgroup is partial, whose frames are not whole 16-byte chunks or that are too big for
the shared launch (the rollout is then run whole, at once), rollouts of changing length, a
two-mover game (no shared launch at all), and the C entry point's own fallback - the two passes
one after the other - through the torch op."""

import ctypes

import numpy as np
import pytest
import torch

from campx_amd import _hip
from campx_amd import gamespec
from campx_amd.games import boat_race, sokoban, wall_world
from oracle import cpu

import traced_games

pytestmark = pytest.mark.gpu


def _check(build, B, Ts, seed, build_kwargs=None, actions_ready=False):
  kw = build_kwargs or {}
  game = build(batch=B, device='cuda', **kw)
  game.its_showtime()
  fused = game.fused
  og = cpu.OracleGame.from_description(gamespec.describe(build(**kw)))
  rng = np.random.RandomState(seed)
  refs, dicts = [], []
  bufs = {}
  for call, T in enumerate(Ts):
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    refs.append(og.rollout(actions, reset_first=(call == 0)))
    key = (T, sum(1 for d in dicts if d['obs'].shape[0] == T) & 1)
    if key not in bufs:
      bufs[key] = fused.rollout_buffers(T)
    out = bufs[key]
    out['obs'].fill_(-7)
    prev = fused.rollout_deferred(torch.from_numpy(actions).cuda(), out, reset_first=(call == 0),
                                  actions_ready=actions_ready)
    assert prev is (dicts[-1] if dicts else None)
    # this call's scalars are ready, the previous call's observations too
    for k in ('reward', 'discount'):
      if out[k] is not None:
        assert np.array_equal(out[k].cpu().numpy().view(np.uint32), refs[-1][k].view(np.uint32)), (B, call, k)
    assert np.array_equal(out['done'].cpu().numpy(), refs[-1]['done']), (B, call)
    if prev is not None:
      assert np.array_equal(prev['obs'].cpu().numpy(), refs[-2]['obs']), (B, call - 1)
    dicts.append(out)
  last = fused.flush()
  assert last is dicts[-1]
  assert np.array_equal(last['obs'].cpu().numpy(), refs[-1]['obs']), (B, 'flush')
  assert fused.flush() is None
  # ... and the engine carries on from there on the ordinary path
  actions = rng.randint(0, 5, size=(20, B)).astype(np.int8)
  out = game.rollout(torch.from_numpy(actions))
  ref = og.rollout(actions)
  assert np.array_equal(out['obs'].cpu().numpy(), ref['obs'])


@pytest.mark.parametrize('B', [16, 1000, 1024, 4096, 5008, 16384])
def test_boat_race_deferred_matches_oracle(B):
  _check(boat_race.build, B, [100, 100, 33, 100, 33], seed=B)


@pytest.mark.parametrize('B,T', [(32768, 20), (24000, 30), (33024, 10)])
def test_big_batches(B, T):
  # 512 and 375 update workgroups in front of the render ones; 33 024 environments are past
  # what the shared launch takes (32 768): each rollout whole, at once
  _check(boat_race.build, B, [T, T, T], seed=B)


def test_wall_world_deferred_matches_oracle():
  _check(wall_world.build, 2000, [64, 64, 64], seed=5)


def _kernels_of(fn):
  from torch.profiler import ProfilerActivity, profile
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fn()
    torch.cuda.synchronize()
  return [e.key for e in prof.key_averages() if 'campx_impl' in e.key]


@pytest.mark.parametrize('level,B,Ts', [(0, 512, [50, 50, 50]), (0, 4096, [100, 100, 33, 100]),
                                        (0, 5008, [40, 40, 40]), (0, 16384, [30, 30, 30]),
                                        (1, 1024, [64, 64, 17, 64]), (1, 8192, [40, 40, 40]),
                                        (2, 1024, [64, 64, 64]), (2, 4096, [50, 50, 20, 50])])
def test_games_of_two_to_four_movers_share_the_launch(level, B, Ts):
  """Round 5: pipe_multi_kernel - the update role walks the pair (sokoban: two movers) or tuple
  table (levels 1 and 2: three and four), the render role patches up to 2K bytes per row from K
  planes of the previous rollout's trace.  Every byte against the oracle, state carried over."""
  _check(sokoban.build, B, Ts, seed=31 * level + B, build_kwargs=dict(level=level))
  game = sokoban.build(level=level, batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  assert f.n_dyn == level + 2
  T = Ts[0]
  sets = [f.rollout_buffers(T), f.rollout_buffers(T)]
  acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
  f.rollout_deferred(acts, sets[0], reset_first=True)
  names = _kernels_of(lambda: f.rollout_deferred(acts, sets[1]))
  assert len(names) == 1 and 'pipe_multi_kernel<%d' % (level + 2) in names[0], names
  f.flush()


def test_multi_mover_games_past_the_shared_launch_pipeline_over_two_streams():
  # 33 024 environments of the two-mover game: each rollout whole, in order.  20 000 of the two-
  # and of the three-mover one, 16 384 of the four-mover one: past what the shared launch takes
  # (16 384 / 8 192), rollout_deferred(actions_ready=True) runs the update pass on the high-priority
  # side stream under the previous rollout's render (campx::rollout_pipelined: two kernels, two
  # streams, one op), complete a call early; without the promise (the default) the same rollouts
  # run in order on the caller's stream
  for ready in (True, False):
    _check(sokoban.build, 33024, [10, 10, 10], seed=6, actions_ready=ready)
    _check(sokoban.build, 20000, [30, 30, 30], seed=9, actions_ready=ready)
    _check(sokoban.build, 16384, [20, 20, 7, 20], seed=7, build_kwargs=dict(level=2), actions_ready=ready)
    _check(sokoban.build, 20000, [16, 16, 16], seed=8, build_kwargs=dict(level=1), actions_ready=ready)


@pytest.mark.parametrize('ready', [False, True])
def test_deferred_rollouts_come_after_the_work_that_makes_their_actions(ready):
  """Round 5 advice: actions produced by kernels queued on the caller's stream BEHIND a long render
  (a `torch.randint`, the int64 -> int8 narrowing of `_action_ids`) must have been written before
  the update pass reads them - whichever route the call takes.  With `actions_ready=True` and ids
  that are NOT the caller's own int8 tensor the side stream waits for the caller's stream."""
  B, T, level = 16384, 40, 2
  game = sokoban.build(level=level, batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  og = cpu.OracleGame.from_description(gamespec.describe(sokoban.build(level=level)))
  sets = [f.rollout_buffers(T), f.rollout_buffers(T)]
  big = torch.empty((1 << 28,), dtype=torch.int32, device='cuda')    # 1 GiB: a fill takes ~0.15 ms
  gen = torch.Generator(device='cuda').manual_seed(77)
  scratch = torch.empty((T, B), dtype=torch.int64, device='cuda')
  got, fed = [], []
  for call in range(6):
    for _ in range(4):
      big.fill_(call)                     # the stream is busy when the actions are made ...
    scratch.fill_(4)                      # ... in a buffer that held "stay" until this instant
    scratch.random_(0, 5, generator=gen)
    prev = f.rollout_deferred(scratch, sets[call & 1], reset_first=(call == 0), actions_ready=ready)
    fed.append(scratch.cpu().numpy().astype(np.int8))
    if prev is not None:
      got.append({k: prev[k].cpu().numpy().copy() for k in ('obs', 'reward', 'done')})
  last = f.flush()
  got.append({k: last[k].cpu().numpy().copy() for k in ('obs', 'reward', 'done')})
  for call, (a, g) in enumerate(zip(fed, got)):
    ref = og.rollout(a, reset_first=(call == 0))
    assert np.array_equal(g['reward'].view(np.uint32), ref['reward'].view(np.uint32)), call
    assert np.array_equal(g['done'], ref['done']), call
    assert np.array_equal(g['obs'], ref['obs']), call


def test_deferred_rollouts_refuse_what_they_cannot_do():
  game = boat_race.build(batch=64, device='cuda')
  game.its_showtime()
  fused = game.fused
  a = torch.zeros((10, 64), dtype=torch.int8, device='cuda')
  with pytest.raises(ValueError, match='deferred rollouts need'):
    fused.rollout_deferred(a, fused.rollout_buffers(10, want_board=True))
  with pytest.raises(ValueError, match='deferred rollouts need'):
    fused.rollout_deferred(a, fused.rollout_buffers(10, keep_obs=False))
  out = fused.rollout_buffers(10)
  assert fused.rollout_deferred(a, out) is None
  with pytest.raises(ValueError, match='still to be rendered'):
    fused.rollout_deferred(a, out)
  assert fused.flush() is out


def test_two_buffer_sets_may_share_their_observations():
  B, T = 1024, 50
  game = boat_race.build(batch=B, device='cuda')
  game.its_showtime()
  fused = game.fused
  og = cpu.OracleGame.from_description(gamespec.describe(boat_race.build()))
  rng = np.random.RandomState(3)
  first = fused.rollout_buffers(T)
  sets = [first, fused.rollout_buffers(T, share=first)]
  ref_prev = None
  for call in range(5):
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    ref = og.rollout(actions, reset_first=(call == 0))
    prev = fused.rollout_deferred(torch.from_numpy(actions).cuda(), sets[call & 1],
                                  reset_first=(call == 0))
    if prev is not None:
      assert np.array_equal(prev['obs'].cpu().numpy(), ref_prev['obs']), call
    ref_prev = ref
  assert np.array_equal(fused.flush()['obs'].cpu().numpy(), ref_prev['obs'])


@pytest.mark.parametrize('build,kw,B', [(boat_race.build, {}, 1000), (boat_race.build, {}, 40000),
                                        (sokoban.build, dict(level=2), 20000), (sokoban.build, {}, 40000)],
                         ids=['ragged-frames', 'past-the-shared-launch', 'two-streams', 'two-movers-in-order'])
def test_shared_observations_where_nothing_is_deferred(build, kw, B):
  """Sizes and games `rollout_deferred()` has no shared launch for (frames that are not whole
  16-byte chunks, more than 32 768 environments, the multi-mover games past their bounds): with
  two dicts over ONE observation buffer the dict a call hands back must still hold the PREVIOUS
  rollout's observations.  Until round 5 these cases rendered the new rollout at once - into the
  buffer the caller was about to read (found by tests/test_random_warehouses.py)."""
  T = 12
  game = build(batch=B, device='cuda', **kw)
  game.its_showtime()
  fused = game.fused
  og = cpu.OracleGame.from_description(gamespec.describe(build(**kw)))
  rng = np.random.RandomState(B)
  first = fused.rollout_buffers(T)
  sets = [first, fused.rollout_buffers(T, share=first)]
  ref_prev = None
  for call in range(5):
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    ref = og.rollout(actions, reset_first=(call == 0))
    prev = fused.rollout_deferred(torch.from_numpy(actions).cuda(), sets[call & 1], reset_first=(call == 0))
    # this call's scalars are there at once, the previous call's observations now
    assert np.array_equal(sets[call & 1]['reward'].cpu().numpy().view(np.uint32), ref['reward'].view(np.uint32))
    if prev is not None:
      assert prev is sets[(call - 1) & 1]
      assert np.array_equal(prev['obs'].cpu().numpy(), ref_prev['obs']), call
    ref_prev = ref
  assert np.array_equal(fused.flush()['obs'].cpu().numpy(), ref_prev['obs'])


def test_tiers_that_run_deferred_rollouts_whole_refuse_a_shared_observation_buffer():
  from campx_amd.games import hello_world, maze
  for game, n_actions in ((maze.build(16, 16, batch=256, device='cuda'), 5),
                          (hello_world.make_game(batch=256, device='cuda')[0], 4)):
    if game.fused is None:
      game.its_showtime()
    T = 8
    first = game.rollout_buffers(T)
    acts = torch.randint(0, n_actions, (T, 256), dtype=torch.int8, device='cuda')
    assert game.rollout_deferred(acts, first, reset_first=True) is None
    with pytest.raises(ValueError, match='would hand back the previous'):
      game.rollout_deferred(acts, game.rollout_buffers(T, share=first))
    own = game.rollout_buffers(T)
    kept = first['obs'].clone()
    assert game.rollout_deferred(acts, own) is first and torch.equal(first['obs'], kept)
    assert game.flush() is own


def test_the_entry_point_runs_the_passes_one_after_the_other_when_it_must():
  # B = 1000: frames of 175 000 bytes, not whole 16-byte chunks - rollout_deferred() would not
  # defer at all; campx_update_render_launch itself (C callers) issues update, then render
  B, T = 1000, 40
  game = boat_race.build(batch=B, device='cuda')
  game.its_showtime()
  fused = game.fused
  assert not _hip.lib.campx_update_render_shared(ctypes.byref(fused.spec), B, T)
  assert _hip.lib.campx_update_render_shared(ctypes.byref(fused.spec), 1024, T)
  og = cpu.OracleGame.from_description(gamespec.describe(boat_race.build()))
  rng = np.random.RandomState(11)
  sets = [fused.rollout_buffers(T), fused.rollout_buffers(T)]
  acts = [rng.randint(0, 5, size=(T, B)).astype(np.int8) for _ in range(3)]
  refs = [og.rollout(a, reset_first=(i == 0)) for i, a in enumerate(acts)]

  def head(i):
    o = sets[i & 1]
    return (fused._spec_host, fused._spec_dev, fused.pos, fused.done, fused.ret, fused._pair_table,
            torch.from_numpy(acts[i]).cuda(), o['reward'], o['discount'], o['done'], o['perf'],
            o['trace'], None, None, i == 0)
  fused._update(*head(0))
  for i in (1, 2):
    prev = sets[(i - 1) & 1]
    prev['obs'].fill_(-7)
    fused._update_render(*(head(i) + (prev['trace'], prev['obs'])))
    assert np.array_equal(prev['obs'].cpu().numpy(), refs[i - 1]['obs']), i
    assert np.array_equal(sets[i & 1]['reward'].cpu().numpy(), refs[i]['reward']), i
  with pytest.raises(RuntimeError, match='trace buffer each'):
    fused._update_render(*(head(2) + (sets[0]['trace'], sets[0]['obs'])))


def test_host_tabulated_game_deferred_matches_its_ordinary_rollouts():
  # a game of arbitrary Python classes (tabulated on the host, `table_only`): the same table
  # kernel body, so the same shared launch - twin engines, one deferred, one not
  B, T = 2048, 60
  twins = [traced_games.ice_rink(batch=B, device='cuda') for _ in range(2)]
  for g in twins:
    g.its_showtime()
  a, b = twins[0].fused, twins[1].fused
  assert a.traced is not None and a.n_dyn == 1
  assert _hip.lib.campx_update_render_shared(ctypes.byref(a.spec), B, T)
  rng = np.random.RandomState(2)
  sets = [a.rollout_buffers(T), a.rollout_buffers(T)]
  want_prev = None
  for call in range(4):
    actions = torch.from_numpy(rng.randint(0, 5, size=(T, B)).astype(np.int8)).cuda()
    want = b.rollout(actions, reset_first=(call == 2))
    prev = a.rollout_deferred(actions, sets[call & 1], reset_first=(call == 2))
    for k in ('reward', 'discount', 'done'):
      if want[k] is not None:
        assert torch.equal(sets[call & 1][k], want[k]), (call, k)
    if prev is not None:
      assert torch.equal(prev['obs'], want_prev), call
    want_prev = want['obs']
  assert torch.equal(a.flush()['obs'], want_prev)
  assert torch.equal(a.pos, b.pos) and torch.equal(a.ret, b.ret)


def test_calls_queued_back_to_back_over_one_observation_buffer():
  """tools/deferred_stress.py: 1 500 calls without a pause, two buffer sets sharing their
  observations, every rollout compared on the device with the two kernels of a twin engine."""
  import os, subprocess, sys
  from conftest import REPO
  out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'deferred_stress.py'), '16384', '1500', '100'],
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  assert out.stdout.strip().endswith('ok B=16384 calls=1500')


@pytest.mark.parametrize('tier', ['wide', 'shape'])
def test_engine_level_calls_on_tiers_without_a_shared_launch(tier):
  """Engine.rollout_deferred / flush on the state-table tier (16x16 maze) and the shape tier
  (Hello World): the rollout is run whole, the calling convention is the same."""
  from campx_amd.games import hello_world, maze
  B, T = 512, 30
  build = (lambda **kw: maze.build(16, 16, **kw)) if tier == 'wide' else hello_world.build
  a, b = build(batch=B, device='cuda'), build(batch=B, device='cuda')
  a.its_showtime()
  b.its_showtime()
  assert type(a.fused).__name__ == ('WideGame' if tier == 'wide' else 'ShapeGame')
  rng = np.random.RandomState(8)
  sets = [a.rollout_buffers(T), a.rollout_buffers(T)]
  want_prev = None
  for call in range(3):
    actions = torch.from_numpy(rng.randint(0, 5, size=(T, B)).astype(np.int8)).cuda()
    want = b.rollout(actions, reset_first=(call == 0))
    prev = a.rollout_deferred(actions, sets[call & 1], reset_first=(call == 0))
    assert prev is (None if call == 0 else sets[(call - 1) & 1])
    if prev is not None:
      assert torch.equal(prev['obs'], want_prev)
    want_prev = want['obs'].clone()
  assert torch.equal(a.flush()['obs'], want_prev)
  assert a.flush() is None


def test_the_two_stream_form_with_new_buffers_for_every_rollout():
  """campx::rollout_pipelined keeps an event per trace buffer it has seen (the update pass may
  not overwrite a trace its last render still reads); a caller that allocates new buffers for
  every rollout used to grow that table without bound - now it is let go of, behind one wait,
  once it holds 64.  150 rollouts into fresh buffers, against the oracle."""
  B, T = 20000, 6
  game = sokoban.build(level=1, batch=B, device='cuda')
  game.its_showtime()
  og = cpu.OracleGame.from_description(gamespec.describe(sokoban.build(level=1)))
  rng = np.random.RandomState(0)
  keep = []
  for i in range(150):
    a = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    out = game.fused.rollout_buffers(T)
    game.fused.rollout(torch.from_numpy(a).cuda(), out=out, pipelined=True)
    ref = og.rollout(a, reset_first=False)
    keep = (keep + [out])[-3:]
    if i % 10 == 0 or i > 140:
      assert np.array_equal(out['obs'].cpu().numpy(), ref['obs']), i
