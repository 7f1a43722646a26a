"""The C ABI (include/campx_hip.h) called directly through ctypes with RANDOM output
configurations - any of reward / discount / done / perf / trace / board absent, every frame kept or
only the last one (strides 0), rows of the per-frame streams B apart or padded, int8 / f16 / bf16
observations, with and without the one-launch scratch, two calls in a row with state carried over
- each accepted call compared with the C oracle, each refused one required to be a clean
CAMPX_EINVAL (never a crash, never silence).  What the Python front end never does: it always
hands over the full set of outputs."""

import ctypes
import os

import numpy as np
import pytest
import torch

from campx_amd import _hip, gamespec
from campx_amd.games import boat_race, sokoban, wall_world
from oracle import cpu

pytestmark = pytest.mark.gpu

GAMES = [(boat_race.build, {}), (wall_world.build, {}), (sokoban.build, {}), (sokoban.build, dict(level=1))]


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  if a.dtype.kind == 'f' or b.dtype.kind == 'f':
    a, b = a.astype(np.float32), b.astype(np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)
  return np.array_equal(a, b)


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_ABI_SEEDS', '40'))))
def test_random_output_configurations(seed):
  rng = np.random.RandomState(4200 + seed)
  build, kw = GAMES[int(rng.randint(len(GAMES)))]
  B = int(rng.choice([1, 3, 17, 64, 250, 1000, 1008, 2049, 4096, 8200]))
  game = build(batch=B, device='cuda', **kw)
  game.its_showtime()
  f = game.fused
  og = cpu.OracleGame.from_description(gamespec.describe(build(**kw)))
  L, H, W, K = f.n_layers, f.rows, f.cols, f.n_dyn
  R, HW = L * H * W, H * W
  dev = torch.device('cuda')
  p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
  st = _hip.CampxState(p(f.pos), p(f.done), p(f.ret), p(f._pair_table))
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  accepted = refused = 0
  fresh = True
  for call in range(3):
    T = int(rng.choice([1, 2, 15, 16, 17, 40, 64, 70]))
    pitch = int(rng.choice([B, (B + 15) // 16 * 16, (B + 15) // 16 * 16 + 16]))
    every = rng.rand() < 0.75
    fmt = int(rng.choice([0, 0, 0, 1, 2]))
    want = {k: rng.rand() < 0.7 for k in ('reward', 'discount', 'done', 'perf', 'trace', 'board')}
    want['perf'] = want['perf'] and f.has_perf
    board_every = every if rng.rand() < 0.8 else not every
    scratch = rng.rand() < 0.5
    reset = fresh or rng.rand() < 0.3
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    tdt = {0: torch.int8, 1: torch.float16, 2: torch.bfloat16}[fmt]
    obs = torch.full((T if every else 1, B, L, H, W), 7, dtype=tdt, device=dev)
    board = torch.full((T if board_every else 1, B, H, W), 7, dtype=torch.int8, device=dev) if want['board'] else None
    rows = lambda dt: torch.full((T, pitch), 7, dtype=dt, device=dev)
    reward = rows(torch.float32) if want['reward'] else None
    discount = rows(torch.float32) if want['discount'] else None
    done = rows(torch.uint8) if want['done'] else None
    perf = rows(torch.int8) if want['perf'] else None
    trace = torch.zeros((K, T, pitch), dtype=torch.uint8, device=dev) if want['trace'] else None
    o = _hip.CampxOutputs()
    o.obs, o.obs_t_stride = p(obs), (B * R if every else 0)
    if board is not None:
      o.board, o.board_t_stride = p(board), (B * HW if board_every else 0)
    o.reward, o.discount, o.done, o.perf, o.trace = p(reward), p(discount), p(done), p(perf), p(trace)
    o.obs_format, o.scalar_pitch = fmt, (0 if pitch == B and rng.rand() < 0.5 else pitch)
    keep = []
    if scratch:
      block = torch.zeros((_hip.lib.campx_flow_scratch_bytes(B, T) + 3) // 4, dtype=torch.int32, device=dev)
      state = _hip.CampxFlowState()
      flag = torch.zeros(1, dtype=torch.int32, device=dev)
      o.overlap_ctl, o.overlap_ctl_bytes = p(block), block.numel() * 4
      o.flow_state, o.error_flag = ctypes.cast(ctypes.pointer(state), ctypes.c_void_p), p(flag)
      keep = [block, state, flag]
    # the state as it is, in case the call is refused
    before = (f.pos.clone(), f.done.clone(), f.ret.clone())
    rc = _hip.lib.campx_rollout_launch(ctypes.byref(f.spec), p(f._spec_dev), st, p(torch.from_numpy(actions).to(dev)),
                                       o, B, T, int(reset), stream)
    torch.cuda.synchronize()
    what = dict(seed=seed, call=call, game=(build.__module__, kw), B=B, T=T, pitch=pitch, every=every, fmt=fmt,
                board_every=board_every, scratch=scratch, reset=reset, **want)
    if rc != 0:
      # a refusal: a clean CAMPX_EINVAL, nothing written to the state
      assert rc == -1, (rc, what)
      assert torch.equal(before[0], f.pos) and torch.equal(before[1], f.done) and torch.equal(before[2], f.ret), what
      # (what the header says is refused: 16-bit observations without the two-kernel path,
      # a board kept differently from the observations)
      assert fmt != 0 or (want['board'] and board_every != every), what
      refused += 1
      continue
    accepted += 1
    fresh = False
    ref = og.rollout(actions, reset_first=reset)
    got = obs.float().cpu().numpy() if fmt else obs.cpu().numpy()
    exp = ref['obs'] if every else ref['obs'][-1:]
    assert _same(got, exp.astype(got.dtype)), what
    if board is not None:
      assert _same(board.cpu().numpy(), ref['board'] if board_every else ref['board'][-1:]), what
    for name, t in (('reward', reward), ('discount', discount), ('done', done), ('perf', perf)):
      if t is not None:
        assert _same(t[:, :B].cpu().numpy(), ref[name]), (name, what)
    if scratch:
      assert int(keep[2].item()) == 0, what
    assert torch.equal(f.done.cpu(), torch.from_numpy(ref['done'][-1])), what
  assert accepted + refused == 3


# ----------------------------------------------------------------- the other two tiers' entry points

def _rows(T, pitch, dt, want, dev):
  return torch.full((T, pitch), 7, dtype=dt, device=dev) if want else None


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_ABI_SEEDS', '40')) // 2))
def test_random_output_configurations_of_the_shape_tier(seed):
  """campx_shape_rollout_launch: with and without the frame-major path's tables / scratch, a flat
  board, every frame or the last one, 16-bit observations, absent per-frame streams, emit_first."""
  from campx_amd.games import hello_world
  import random_hellos
  rng = np.random.RandomState(5200 + seed)
  builders = [hello_world.build] + [random_hellos.library_builder(d) for d in random_hellos.definitions()[:6]]
  build = builders[int(rng.randint(len(builders)))]
  B = int(rng.choice([1, 3, 16, 64, 250, 1000, 2048]))
  game = build(batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  og = cpu.OracleGame.from_description(gamespec.describe(build()))
  L, H, W = f.n_layers, f.rows, f.cols
  R, HW = L * H * W, H * W
  dev = torch.device('cuda')
  p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
  st = _hip.CampxState(p(f.pos), p(f.done), p(f.ret), None)
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  accepted = 0
  for call in range(3):
    T = int(rng.choice([1, 3, 4, 5, 17, 40]))
    every = rng.rand() < 0.75
    fmt = int(rng.choice([0, 0, 0, 1, 2]))
    want = {k: rng.rand() < 0.7 for k in ('reward', 'discount', 'done', 'board', 'fast')}
    board_every = every
    reset = call == 0 or rng.rand() < 0.3
    actions = rng.choice(5, size=(T, B), p=[.24, .24, .24, .24, .04]).astype(np.int8)
    tdt = {0: torch.int8, 1: torch.float16, 2: torch.bfloat16}[fmt]
    obs = torch.full((T if every else 1, B, L, H, W), 7, dtype=tdt, device=dev)
    board = torch.full((T if board_every else 1, B, H, W), 7, dtype=torch.int8, device=dev) if want['board'] else None
    reward = _rows(T, B, torch.float32, want['reward'], dev)
    discount = _rows(T, B, torch.float32, want['discount'], dev)
    done = _rows(T, B, torch.uint8, want['done'], dev)
    o = _hip.CampxOutputs()
    o.obs, o.obs_t_stride = p(obs), (B * R if every else 0)
    if board is not None:
      o.board, o.board_t_stride = p(board), (B * HW if board_every else 0)
    o.reward, o.discount, o.done, o.obs_format = p(reward), p(discount), p(done), fmt
    tables = scratch = None
    if want['fast'] and f._tables is not None:
      need = int(_hip.lib.campx_shape_scratch_bytes(ctypes.byref(f.spec), B, T))
      scratch = torch.zeros((need + 7) // 8, dtype=torch.int64, device=dev)
      tables = f._tables
      o.trace = p(scratch)
    rc = _hip.lib.campx_shape_rollout_launch(ctypes.byref(f.spec), p(f._spec_dev), p(tables), st, p(f.backdrop),
                                             p(torch.from_numpy(actions).to(dev)), o, B, T, int(reset), 0, stream)
    torch.cuda.synchronize()
    what = dict(seed=seed, call=call, B=B, T=T, every=every, fmt=fmt, reset=reset, HW=(H, W), **want)
    assert rc == 0, (rc, what)
    accepted += 1
    ref = og.rollout(actions, reset_first=reset)
    got = obs.float().cpu().numpy() if fmt else obs.cpu().numpy()
    exp = ref['obs'] if every else ref['obs'][-1:]
    assert _same(got, exp.astype(got.dtype)), what
    if board is not None:
      assert _same(board.cpu().numpy(), ref['board'] if board_every else ref['board'][-1:]), what
    for name, t in (('reward', reward), ('discount', discount), ('done', done)):
      if t is not None:
        assert _same(t.cpu().numpy(), ref[name]), (name, what)
    assert torch.equal(f.done.cpu(), torch.from_numpy(ref['done'][-1])), what
  assert accepted == 3


@pytest.mark.parametrize('seed', range(int(os.environ.get('CAMPX_ABI_SEEDS', '40')) // 2))
def test_random_output_configurations_of_the_state_table_tier(seed):
  """campx_wide_rollout_launch on a 16x16 maze (one mover) and the 16x16 two-box sokoban."""
  from campx_amd.games import maze
  rng = np.random.RandomState(6200 + seed)
  build, kw = [(lambda **k: maze.build(16, 16, **k), {}), (sokoban.build, dict(level=3))][int(rng.randint(2))]
  B = int(rng.choice([1, 3, 17, 64, 250, 1000, 1008, 4096]))
  game = build(batch=B, device='cuda', **kw)
  game.its_showtime()
  f = game.fused
  og = cpu.OracleGame.from_description(gamespec.describe(build(**kw)))
  L, H, W, K = f.n_layers, f.rows, f.cols, f.n_dyn
  R, HW = L * H * W, H * W
  dev = torch.device('cuda')
  p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
  st = _hip.CampxState(p(f.state), p(f.done), p(f.ret), None)
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  accepted = refused = 0
  for call in range(3):
    T = int(rng.choice([1, 2, 15, 16, 17, 40]))
    pitch = int(rng.choice([B, (B + 15) // 16 * 16, (B + 15) // 16 * 16 + 16]))
    every = rng.rand() < 0.75
    fmt = int(rng.choice([0, 0, 0, 1, 2]))
    want = {k: rng.rand() < 0.7 for k in ('reward', 'discount', 'done', 'perf', 'board')}
    want['perf'] = want['perf'] and f.has_perf
    reset = call == 0 or rng.rand() < 0.3
    actions = rng.randint(0, 5, size=(T, B)).astype(np.int8)
    tdt = {0: torch.int8, 1: torch.float16, 2: torch.bfloat16}[fmt]
    obs = torch.full((T if every else 1, B, L, H, W), 7, dtype=tdt, device=dev)
    board = torch.full((T if every else 1, B, H, W), 7, dtype=torch.int8, device=dev) if want['board'] else None
    reward = _rows(T, pitch, torch.float32, want['reward'], dev)
    discount = _rows(T, pitch, torch.float32, want['discount'], dev)
    done = _rows(T, pitch, torch.uint8, want['done'], dev)
    perf = _rows(T, pitch, torch.int8, want['perf'], dev)
    trace = torch.zeros((K, T, pitch), dtype=torch.int16, device=dev)
    o = _hip.CampxOutputs()
    o.obs, o.obs_t_stride = p(obs), (B * R if every else 0)
    if board is not None:
      o.board, o.board_t_stride = p(board), (B * HW if every else 0)
    o.reward, o.discount, o.done, o.perf, o.trace = p(reward), p(discount), p(done), p(perf), p(trace)
    o.obs_format, o.scalar_pitch = fmt, (0 if pitch == B and rng.rand() < 0.5 else pitch)
    before = (f.state.clone(), f.done.clone())
    rc = _hip.lib.campx_wide_rollout_launch(ctypes.byref(f.spec), p(f._tables), st, p(torch.from_numpy(actions).to(dev)),
                                            o, B, T, int(reset), stream)
    torch.cuda.synchronize()
    what = dict(seed=seed, call=call, B=B, T=T, pitch=pitch, every=every, fmt=fmt, reset=reset, **want)
    if rc != 0:
      assert rc == -1 and fmt != 0, (rc, what)
      assert torch.equal(before[0], f.state) and torch.equal(before[1], f.done), what
      refused += 1
      continue
    accepted += 1
    ref = og.rollout(actions, reset_first=reset)
    got = obs.float().cpu().numpy() if fmt else obs.cpu().numpy()
    exp = ref['obs'] if every else ref['obs'][-1:]
    assert _same(got, exp.astype(got.dtype)), what
    if board is not None:
      assert _same(board.cpu().numpy(), ref['board'] if every else ref['board'][-1:]), what
    for name, t in (('reward', reward), ('discount', discount), ('done', done), ('perf', perf)):
      if t is not None:
        assert _same(t[:, :B].cpu().numpy(), ref[name]), (name, what)
  assert accepted + refused == 3
