"""The torch custom-op boundary (csrc/campx_torch.cpp): registration, schemas, loud
failure without a HIP device (CPU part); behaviour through the ops, opcheck and
torch.compile on the GPU (gpu part)."""

import ctypes

import numpy as np
import pytest
import torch

from campx_amd import _hip, gamespec
from campx_amd.games import boat_race


def _spec_tensor():
  spec = gamespec.lower(gamespec.describe(boat_race.build()))
  return torch.frombuffer(bytearray(gamespec.spec_bytes(spec)), dtype=torch.uint8)


def test_ops_are_registered_with_mutable_schemas():
  for name in _hip.OP_NAMES:
    op = getattr(torch.ops.campx, name).default
    assert op._schema.returns == []            # everything is written in place
  s = str(torch.ops.campx.rollout.default._schema)
  for written in ('Tensor(a!) pos', 'Tensor(b!) done', 'Tensor(d!) obs', 'Tensor(f!)? reward',
                  'Tensor(j!)? trace', 'bool reset_first'):
    assert written in s, s
  s = str(torch.ops.campx.step.default._schema)
  assert 'Tensor actions' in s and 'Tensor(k!)? bad_flag' in s


def test_no_cpu_kernel_so_cpu_tensors_fail_loudly():
  spec = _spec_tensor()
  B = 8
  pos = torch.zeros((2, B), dtype=torch.int8)
  done = torch.zeros((B,), dtype=torch.uint8)
  obs = torch.zeros((B, 7, 5, 5), dtype=torch.int8)
  acts = torch.zeros((B,), dtype=torch.int8)
  with pytest.raises((NotImplementedError, RuntimeError)) as e:
    torch.ops.campx.step(spec, spec, pos, done, None, None, acts, obs, None, None, None,
                         None, None, None, None)
  assert 'CPU' in str(e.value)
  with pytest.raises((NotImplementedError, RuntimeError)):
    torch.ops.campx.reset(spec, spec, pos, done, None, None, obs, None)


def test_meta_kernels_trace_without_a_device():
  """The Meta registration is the fake-tensor implementation: tracing needs no GPU."""
  spec = _spec_tensor().to('meta')
  B = 8
  pos = torch.zeros((2, B), dtype=torch.int8, device='meta')
  done = torch.zeros((B,), dtype=torch.uint8, device='meta')
  obs = torch.zeros((3, B, 7, 5, 5), dtype=torch.int8, device='meta')
  acts = torch.zeros((3, B), dtype=torch.int8, device='meta')
  assert torch.ops.campx.rollout(spec, spec, pos, done, None, None, acts, obs, None, None,
                                 None, None, None, None, None, None, True) is None


# ----------------------------------------------------------------------- GPU

def _game(batch=256):
  game = boat_race.build(batch=batch, device='cuda')
  game.its_showtime()
  return game


@pytest.mark.gpu
def test_play_and_rollout_go_through_the_ops(golden, monkeypatch):
  gold = golden('boat_race')
  T, N = gold['actions'].shape
  game = _game(N)
  calls = []
  real_step, real_rollout = game.fused._step, game.fused._rollout
  monkeypatch.setattr(game.fused, '_step', lambda *a: (calls.append('step'), real_step(*a))[1])
  monkeypatch.setattr(game.fused, '_rollout',
                      lambda *a: (calls.append('rollout'), real_rollout(*a))[1])
  obs, reward, discount = game.play(torch.from_numpy(gold['actions'][0]))
  assert np.array_equal(obs.layered_board.cpu().numpy(), gold['layered'][1])
  out = game.rollout(torch.from_numpy(gold['actions'][1:]))
  assert np.array_equal(out['obs'].cpu().numpy(), gold['layered'][2:])
  assert calls == ['step', 'rollout']


@pytest.mark.gpu
def test_ops_run_on_the_current_stream():
  """Work is enqueued on torch's current stream: a rollout issued on a side stream is
  ordered after what that stream already holds, with no synchronisation in the op."""
  game = _game(4096)
  acts = torch.randint(0, 5, (30, 4096), dtype=torch.int8, device='cuda')
  want = game.rollout(acts, reset_first=True)
  torch.cuda.synchronize()
  side = torch.cuda.Stream()
  bufs = game.fused.rollout_buffers(30)
  with torch.cuda.stream(side):
    torch.cuda._sleep(20_000_000)              # ~10 ms of work ahead of the op on `side`
    got = game.rollout(acts, out=bufs, reset_first=True)
    done_early = side.query()
  side.synchronize()
  assert not done_early
  assert torch.equal(got['obs'], want['obs']) and torch.equal(got['reward'], want['reward'])


@pytest.mark.gpu
def test_opcheck():
  game = _game(128)
  f = game.fused
  acts = torch.randint(0, 5, (128,), dtype=torch.int8, device='cuda')
  args = (f._spec_host, f._spec_dev, f.pos, f.done, f.ret, None, acts, f._obs, f._board,
          f._reward, f._discount, f._step_done, f.perf, f._bad, None)
  torch.library.opcheck(torch.ops.campx.step.default, args)      # all four checks
  acts = torch.randint(0, 5, (5, 128), dtype=torch.int8, device='cuda')
  b = f.rollout_buffers(5, want_board=True)
  args = (f._spec_host, f._spec_dev, f.pos, f.done, f.ret, None, acts, b['obs'], b['board'],
          b['reward'], b['discount'], b['done'], b['perf'], b['trace'], f._bad, None, True)
  torch.library.opcheck(torch.ops.campx.rollout.default, args)


@pytest.mark.gpu
def test_torch_compile_traces_through_the_op_without_a_graph_break(golden):
  """A policy-in-the-loop step - observation -> action ids -> campx::step - compiles as
  ONE graph (fullgraph=True raises on any graph break) and matches eager."""
  game_c, game_e = _game(512), _game(512)
  w = torch.randn(7 * 25, 5, device='cuda')

  def loop(f, steps):
    total = torch.zeros((), device='cuda')
    for _ in range(steps):
      logits = f._obs.view(f.batch, -1).float() @ w
      ids = logits.argmax(dim=1).to(torch.int8)
      torch.ops.campx.step(f._spec_host, f._spec_dev, f.pos, f.done, f.ret, None, ids,
                           f._obs, f._board, f._reward, f._discount, f._step_done, f.perf,
                           None, None)
      total = total + f._reward.sum()
    return total

  compiled = torch.compile(loop, fullgraph=True, backend='aot_eager')
  a = compiled(game_c.fused, 3)
  b = loop(game_e.fused, 3)
  torch.cuda.synchronize()
  assert torch.equal(a, b)
  assert torch.equal(game_c.fused._obs, game_e.fused._obs)
  assert torch.equal(game_c.fused.pos, game_e.fused.pos)


@pytest.mark.gpu
def test_play_is_capturable_in_a_hip_graph():
  """campx::step only enqueues kernels on the current stream (no allocation, no
  synchronisation), so a policy-in-the-loop stretch of play() calls can be captured in a
  HIP graph and replayed: one graph launch for many frames."""
  game_g, game_e = _game(1024), _game(1024)
  for g in (game_g, game_e):
    g.fused.validate_actions = False
  w = torch.randn(7 * 25, 5, device='cuda')

  def frames(game, n):
    f = game.fused
    for _ in range(n):
      ids = (f._obs.view(f.batch, -1).float() @ w).argmax(dim=1).to(torch.int8)
      game.play(ids)

  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    frames(game_g, 2)                          # warm up allocations outside the capture
  torch.cuda.current_stream().wait_stream(side)
  frames(game_e, 2)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    frames(game_g, 6)
  for _ in range(3):
    graph.replay()
  frames(game_e, 18)
  torch.cuda.synchronize()
  assert torch.equal(game_g.fused._obs, game_e.fused._obs)
  assert torch.equal(game_g.fused.pos, game_e.fused.pos)
  assert torch.equal(game_g.fused.ret, game_e.fused.ret)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['boat_race', 'sokoban', 'wall_world', 'sokoban_l2'])
def test_pipelined_rollouts_match_in_order_rollouts(name):
  """Update pass on a side stream, overlapping the previous launch's render: same bits."""
  import sys, os
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from games_under_test import FUSED_GAMES
  if True:
    B, T = 4096, 40
    a = FUSED_GAMES[name](batch=B, device='cuda'); a.its_showtime()
    b = FUSED_GAMES[name](batch=B, device='cuda'); b.its_showtime()
    gen = torch.Generator().manual_seed(3)
    acts = [torch.randint(0, 5, (T, B), generator=gen, dtype=torch.int8).cuda() for _ in range(5)]
    torch.cuda.synchronize()
    bufs = [b.fused.rollout_buffers(T)]
    bufs.append(b.fused.rollout_buffers(T, share=bufs[0]))
    for i, x in enumerate(acts):
      want = a.rollout(x, reset_first=(i % 2 == 0))
      got = b.rollout(x, out=bufs[i & 1], reset_first=(i % 2 == 0), pipelined=True)
      for k in ('obs', 'reward', 'discount', 'done', 'perf', 'trace'):
        if want[k] is not None:
          assert torch.equal(want[k], got[k]), (i, k)
      if i == 2:                                 # an in-order call in between
        o1, r1, _ = a.play(x[0]); o2, r2, _ = b.play(x[0])
        assert torch.equal(o1.layered_board, o2.layered_board) and torch.equal(r1, r2)
    assert torch.equal(a.fused.pos, b.fused.pos) and torch.equal(a.fused.ret, b.fused.ret)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['boat_race', 'sokoban'])
def test_pipelined_rollouts_do_not_overwrite_a_trace_still_being_rendered(name):
  """Six pipelined calls with different actions, two alternating buffer sets with their own
  observation buffers, ONE synchronisation at the end: the side stream must not start the
  update pass of call i + 2 before the render of call i has read the trace they share.
  Long renders (T = 100 at B = 65 536) against update passes a tenth as long make the
  overrun certain without that ordering."""
  import sys, os
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from games_under_test import FUSED_GAMES
  B, T, N = 65536, 100, 6
  a = FUSED_GAMES[name](batch=B, device='cuda'); a.its_showtime()
  b = FUSED_GAMES[name](batch=B, device='cuda'); b.its_showtime()
  gen = torch.Generator().manual_seed(11)
  acts = [torch.randint(0, 5, (T, B), generator=gen, dtype=torch.int8).cuda() for _ in range(N)]
  # what each call must show at a sample of frames, from in-order rollouts
  frames = [0, T // 2, T - 1]
  want = []
  for i, x in enumerate(acts):
    out = a.rollout(x, reset_first=(i == 0))
    want.append([out['obs'][f].clone() for f in frames])
  torch.cuda.synchronize()
  bufs = [b.fused.rollout_buffers(T), b.fused.rollout_buffers(T)]
  kept = []
  for i, x in enumerate(acts):
    got = b.rollout(x, out=bufs[i & 1], reset_first=(i == 0), pipelined=True)
    # keep the sampled frames by a copy queued on the main stream (ordered after the render)
    kept.append([got['obs'][f].clone() for f in frames])
  torch.cuda.synchronize()
  for i in range(N):
    for j, f in enumerate(frames):
      assert torch.equal(kept[i][j], want[i][j]), (i, f)
  assert torch.equal(a.fused.pos, b.fused.pos)
