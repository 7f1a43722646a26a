"""The shape tier: Hello-World-style games on the HIP device.

Games made only of rigidly translated things that interact with nothing
(`rules.RollingDrape`, `rules.SlidingSprite`, `FixedDrape`; the reference's
examples/Hello World Example.ipynb) are lowered to a `CampxShapeSpec`
(`gamespec.lower_shapes`) and stepped by `campx::shape_rollout`
(csrc/campx_torch.cpp -> campx_shape_rollout_launch, one wavefront per environment).
Same surface as `fused.FusedGame`: `showtime()`, `play(actions)`, `rollout(actions)`;
actions are the game's integer ids `[B]` / `[T, B]` (Hello World: 0..3 move, 4 quits).

State per environment: each thing's cyclic (row, col) offset from its art position,
the game-over latch, the running return, and - because sprites painted before the
first drape write into the backdrop on the reference's renderer -
the backdrop itself.
"""

import ctypes
import os

import torch

from . import _hip
from . import gamespec
from .rendering import Observation


# Rollouts that keep every frame run FRAME-MAJOR (round 5, csrc/k_shape.hip): an update pass, then
# a render pass of one-shot waves with memory-aligned 2 KiB windows that computes every row of the
# observation arithmetically from 64-bit row words (the things' masks in every column rotation,
# the per-environment trail words of sprites painted before the first drape) - the one-cell
# tier's store pattern instead of "every wave streams its own row".  Boards with rows of 16 to 64
# cells; False (or the library setting shape_split=0): always the one-wave-per-environment kernel.
FRAME_MAJOR = True


class ShapeGame(object):

  def __init__(self, engine, batch, device=None, description=None):
    if not torch.cuda.is_available():
      raise RuntimeError(
          'the fused tier needs a HIP device (torch.cuda.is_available() is '
          'False) and has no CPU fallback; use batch=None for the '
          'single-environment generic tier')
    self.device = torch.device('cuda' if device is None else device)
    if self.device.type != 'cuda':
      raise ValueError('fused tier: device must be a HIP/cuda device, got {}'
                       .format(self.device))
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    self.batch = int(batch)
    if self.batch < 1:
      raise ValueError('batch must be >= 1')
    self.description = description or gamespec.describe(engine)
    self.spec = gamespec.lower_shapes(self.description)
    _hip.check(_hip.lib.campx_shape_spec_validate(ctypes.byref(self.spec)),
               'campx_shape_spec_validate')
    self.chars = list(self.description.chars)
    self.rows, self.cols = engine.rows, engine.cols
    self.n_layers = len(self.chars)
    self.n_dyn = self.spec.n_things
    self.any_reward = bool(self.spec.any_reward)
    self.has_perf = False
    self.uses_table = False
    # no visible sprite is painted before the first drape: the backdrop never changes, and
    # rollouts can take the two-kernel path (csrc/k_shape.hip shape_render_kernel)
    self.trail_free = not any(self.spec.things[k].visible for k in range(self.spec.first_drape))
    # the frame-major path's row tables (campx_shape_tables_build: pure host code), if the game
    # is one for it
    self._tables = None
    n = int(_hip.lib.campx_shape_tables_bytes(ctypes.byref(self.spec)))
    if n > 0 and FRAME_MAJOR:
      host = torch.zeros((n + 7) // 8, dtype=torch.int64)
      _hip.check(_hip.lib.campx_shape_tables_build(ctypes.byref(self.spec),
                                                   ctypes.c_void_p(host.data_ptr()), host.numel() * 8),
                 'campx_shape_tables_build')
      self._tables = host.to(self.device)
    B, dev = self.batch, self.device
    blob = ctypes.string_at(ctypes.addressof(self.spec), ctypes.sizeof(self.spec))
    self._spec_host = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    self._spec_dev = self._spec_host.to(dev)
    self.pos = torch.zeros((2 * self.n_dyn, B), dtype=torch.int8, device=dev)
    self.done = torch.zeros((B,), dtype=torch.uint8, device=dev)
    self.ret = torch.zeros((B,), dtype=torch.float32, device=dev)
    self.backdrop = torch.zeros((B, self.rows * self.cols), dtype=torch.int8, device=dev)
    self._obs = torch.empty((B, self.n_layers, self.rows, self.cols), dtype=torch.int8,
                            device=dev)
    self._board = torch.empty((B, self.rows, self.cols), dtype=torch.int8, device=dev)
    self._reward = torch.empty((B,), dtype=torch.float32, device=dev)
    self._discount = torch.empty((B,), dtype=torch.float32, device=dev)
    self._step_done = torch.empty((B,), dtype=torch.uint8, device=dev)
    self._bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    self._bad_flag = torch.zeros((1,), dtype=torch.int32).pin_memory()
    self._bad_flag_view = self._bad_flag.numpy()
    self.validate_actions = True
    self.frame = -1
    layers = {ch: self._obs[:, i] for i, ch in enumerate(self.chars)}
    self._observation_cache = Observation(board=self._board, layers=layers,
                                          layered_board=self._obs)
    self._op = _hip.ops.shape_rollout.default

  # same bookkeeping as FusedGame
  def _raise_bad(self):
    n = int(self._bad.item())
    self._bad.zero_()
    self._bad_flag_view[0] = 0
    if n:
      raise ValueError('{} action ids are outside 0..{}'.format(n, gamespec.N_ACTIONS - 1))

  def check_actions(self):
    torch.cuda.synchronize(self.device)
    self._raise_bad()

  def _after_launch(self):
    if self.validate_actions == 'sync':
      self._raise_bad()
    elif self.validate_actions and self._bad_flag_view[0]:
      self._raise_bad()

  def _ids(self, actions, expect):
    if not torch.is_tensor(actions):
      actions = torch.as_tensor(actions)
    if actions.is_floating_point():
      raise ValueError('shape games take integer action ids, not one-hot vectors')
    if actions.dim() == 0 and len(expect) == 1:
      # ONE action for every environment: `game.play(0)`, Hello World notebook cell 6
      actions = actions.to(torch.int64).clamp(-1, gamespec.N_ACTIONS).to(torch.int8).expand(expect)
    actions = actions.to(self.device)
    if tuple(actions.shape) != tuple(expect):
      raise ValueError('action ids must have shape {}, got {}'.format(
          tuple(expect), tuple(actions.shape)))
    if actions.dtype != torch.int8:
      actions = actions.clamp(-1, gamespec.N_ACTIONS).to(torch.int8)
    return actions.contiguous()

  def showtime(self):
    self._op(self._spec_host, self._spec_dev, self.pos, self.done, self.ret, self.backdrop,
             None, self._obs, self._board, None, None, None, None, None, True, True)   # (in play()'s format)
    self.frame = 0
    return self._observation_cache, None, 1.0

  def reset(self):
    """A new episode for every environment (see `FusedGame.reset`)."""
    return self.showtime()

  def set_play_obs_dtype(self, dtype):
    """Make `play()` return `layered_board` in `dtype` (torch.int8, float16 or bfloat16): the
    policy-network input of examples/reinforce.py:149, written by the kernel itself."""
    if dtype not in (torch.int8, torch.float16, torch.bfloat16):
      raise ValueError('obs dtype must be torch.int8, float16 or bfloat16')
    if dtype != self._obs.dtype:
      self._obs = self._obs.to(dtype)        # keep the frame currently shown
      layers = {ch: self._obs[:, i] for i, ch in enumerate(self.chars)}
      self._observation_cache = Observation(board=self._board, layers=layers,
                                            layered_board=self._obs)

  def play(self, actions):
    ids = self._ids(actions, (self.batch,))
    validate = self.validate_actions
    self._op(self._spec_host, self._spec_dev, self.pos, self.done, self.ret, self.backdrop,
             ids, self._obs, self._board, self._reward, self._discount, self._step_done,
             self._bad if validate else None, self._bad_flag if validate else None,
             False, False)
    self.frame += 1
    if validate:
      self._after_launch()
    return (self._observation_cache, (self._reward if self.any_reward else None),
            self._discount)

  def capture_play(self, n_frames, policy=None, record_obs=False):
    """`n_frames` consecutive `play()` calls - and, with `policy(observation, t) -> ids [B]`, the
    policy's forward pass and sampling in front of each - captured once in a HIP graph;
    `.replay()` then runs them with one launch of the host's (campx_amd/play_graph.py)."""
    from .play_graph import PlayGraph
    return PlayGraph(self, n_frames, policy=policy, record_obs=record_obs)

  def rollout_buffers(self, T, keep_obs=True, want_board=False, obs_dtype=torch.int8,
                      share=None):
    if obs_dtype not in (torch.int8, torch.float16, torch.bfloat16):
      raise ValueError('obs_dtype must be torch.int8, float16 or bfloat16')
    if obs_dtype != torch.int8 and not keep_obs:
      raise ValueError('16-bit observations need keep_obs=True')
    B, L, H, W, dev = self.batch, self.n_layers, self.rows, self.cols, self.device
    if share is not None:
      obs, board = share['obs'], share['board']
    else:
      obs = (torch.empty((T, B, L, H, W), dtype=obs_dtype, device=dev) if keep_obs
             else self._obs)
      board = None
      if want_board:
        board = (torch.empty((T, B, H, W), dtype=torch.int8, device=dev) if keep_obs
                 else self._board)
    # the scratch of the frame-major path (the things' offsets per frame, the trail words every
    # fourth frame); the library ignores it for the calls that path does not take
    trace = None
    if (self._tables is not None and keep_obs and T > 0 and obs_dtype == torch.int8
        and board is None):
      need = int(_hip.lib.campx_shape_scratch_bytes(ctypes.byref(self.spec), B, T))
      if need > 0:
        trace = torch.empty((need + 7) // 8, dtype=torch.int64, device=dev)
    return dict(obs=obs, board=board,
                reward=(torch.empty((T, B), dtype=torch.float32, device=dev)
                        if self.any_reward else None),
                discount=torch.empty((T, B), dtype=torch.float32, device=dev),
                done=torch.empty((T, B), dtype=torch.uint8, device=dev),
                perf=None, trace=trace)

  def rollout_deferred(self, actions, out, reset_first=False, actions_ready=False):
    """`FusedGame.rollout_deferred` for this tier, which has no shared launch (one kernel does
    update and render): the rollout is run whole, at once, and `out` is simply complete a call
    early; returns the previous call's dict (None on the first)."""
    prev = getattr(self, '_deferred', None)
    if prev is not None and prev['obs'].data_ptr() == out['obs'].data_ptr():
      raise ValueError('the shape tier runs a deferred rollout whole, at once: two dicts over ONE '
                       'observation buffer (rollout_buffers(T, share=...)) would hand back the previous '
                       'rollout\'s dict with this rollout\'s observations in it - give each dict its own '
                       'buffers (rollout_buffers(T) twice)')
    self.rollout(actions, out=out, reset_first=reset_first)
    self._deferred = out
    return prev

  def flush(self):
    prev, self._deferred = getattr(self, '_deferred', None), None
    return prev

  def rollout(self, actions, obs=None, board=None, keep_obs=True, reset_first=False,
              want_board=False, obs_dtype=torch.int8, out=None, pipelined=False):
    """T frames in one launch; arguments and result as `FusedGame.rollout`."""
    if pipelined:
      raise ValueError('the shape tier is a single kernel: nothing to pipeline')
    T = int(actions.shape[0])
    if T < 1:
      raise ValueError('a rollout needs at least one frame: actions [T, B] with T >= 1')
    ids = self._ids(actions, (T, self.batch))
    if out is None:
      out = self.rollout_buffers(T, keep_obs, want_board or board is not None, obs_dtype)
      if obs is not None and keep_obs:
        out['obs'] = obs
      if board is not None:
        out['board'] = board
    validate = self.validate_actions
    self._op(self._spec_host, self._spec_dev, self.pos, self.done, self.ret, self.backdrop,
             ids, out['obs'], out['board'], out['reward'], out['discount'], out['done'],
             self._bad if validate else None, self._bad_flag if validate else None,
             bool(reset_first), False, out.get('trace'),
             self._tables if out.get('trace') is not None else None)
    self.frame = T if reset_first else self.frame + T
    if validate:
      self._after_launch()
    return out
