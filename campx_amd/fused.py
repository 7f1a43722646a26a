"""The fused tier: B environments stepped by the HIP kernel.

`FusedGame` is what a batched `Engine` delegates to after `its_showtime()`.  It
lowers the engine to a GameSpec (`gamespec`), keeps the dynamic state and the
output buffers as torch tensors in HBM, and calls the C ABI of
libcampx_hip.so (`_hip`, include/campx_hip.h) on torch's current HIP stream.
PyTorch is used for device memory and streams only; every per-frame computation
is inside the kernel.

There is no CPU path here: constructing a FusedGame without a HIP device raises.

Return shapes (the batched form of the reference's `Engine.play()` triple,
campx/engine.py:166):
    Observation.board          int8 [B, H, W]     character codes
    Observation.layers[ch]     int8 [B, H, W]     view of layered_board[:, l]
    Observation.layered_board  int8 [B, L, H, W]  channels ascending by character
    reward                     float32 [B]        (None for games that never reward)
    discount                   float32 [B]
The tensors returned by `play()` are the engine's own buffers and are overwritten
by the next call, like the reference's (campx/rendering.py:59-64): copy to keep.

Deviation from the reference, inherent to batching: game-over is per
environment.  Instead of raising on `play()` after termination
(campx/engine.py:149-151) a finished environment is rebuilt from the art before
its next action is applied; `done` (also `game.fused.done`) tells which ones ended.
"""

import ctypes
import os

import torch

from . import _hip
from . import gamespec
from .rendering import Observation


# One-mover games: tabulate the update pass per (cell, action) at showtime
# (campx_spec_compile) and let the frame loop look it up.  Tests switch this off to
# exercise the rule interpreter on the same games.
COMPILE_TABLE = True
# Rollouts that keep every frame run the update pass and the render as two kernels
# (needs a [K, T, B] int32 trace buffer): 0.196 ms vs 0.224 ms per 100-frame launch of
# the boat race at B = 65 536 (DESIGN.md "Kernels", profiles/).  CAMPX_SPLIT=0 keeps
# everything in the single fused kernel; parity tests run both.
SPLIT_ROLLOUT = os.environ.get('CAMPX_SPLIT', '1') in ('1', 'force')
FORCE_SPLIT = os.environ.get('CAMPX_SPLIT', '') == 'force'   # also for multi-mover games


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class FusedGame(object):

  def __init__(self, engine, batch, device=None):
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
    self.description = gamespec.describe(engine)
    self.spec = gamespec.lower(self.description)
    _hip.check(_hip.lib.campx_spec_validate(ctypes.byref(self.spec)),
               'campx_spec_validate')
    with torch.cuda.device(self.device):
      _hip.check(_hip.lib.campx_spec_compile(
          ctypes.byref(self.spec),
          ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
          'campx_spec_compile')
    if not COMPILE_TABLE:
      self.spec.table_valid = 0     # keep the render tables, interpret the rules
    self.uses_table = bool(self.spec.table_valid)
    self.chars = list(self.description.chars)
    self.rows, self.cols = engine.rows, engine.cols
    self.n_layers = len(self.chars)
    self.n_dyn = self.spec.n_dyn
    self.any_reward = bool(self.spec.any_reward)
    self.has_perf = self.spec.perf_dyn >= 0

    B, dev = self.batch, self.device
    blob = torch.frombuffer(bytearray(gamespec.spec_bytes(self.spec)),
                            dtype=torch.uint8)
    self._spec_dev = blob.to(dev)
    # Two-mover games: the (cell, cell, action) table of the update pass.
    self._pair_table = None
    n_pair = int(_hip.lib.campx_pair_table_bytes(ctypes.byref(self.spec)))
    if COMPILE_TABLE and n_pair > 0:
      table = torch.empty((n_pair,), dtype=torch.uint8, device=dev)
      with torch.cuda.device(self.device):
        rc = _hip.lib.campx_pair_table_build(
            ctypes.byref(self.spec), _ptr(self._spec_dev), _ptr(table),
            ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
      if rc == 0:
        self._pair_table = table
      elif rc != -2:          # -2: more than 256 distinct rewards -> just interpret
        _hip.check(rc, 'campx_pair_table_build')
    self.uses_table = self.uses_table or self._pair_table is not None
    self.pos = torch.zeros((2 * self.n_dyn, B), dtype=torch.int8, device=dev)
    self.done = torch.zeros((B,), dtype=torch.uint8, device=dev)
    self.ret = torch.zeros((B,), dtype=torch.float32, device=dev)
    self._obs = torch.empty((B, self.n_layers, self.rows, self.cols),
                            dtype=torch.int8, device=dev)
    self._board = torch.empty((B, self.rows, self.cols), dtype=torch.int8,
                              device=dev)
    self._reward = torch.empty((B,), dtype=torch.float32, device=dev)
    self._discount = torch.empty((B,), dtype=torch.float32, device=dev)
    self._step_done = torch.empty((B,), dtype=torch.uint8, device=dev)
    self.perf = torch.zeros((B,), dtype=torch.int8, device=dev)
    self._ids = torch.empty((B,), dtype=torch.int8, device=dev)
    self._bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    self.validate_actions = True
    self.frame = -1
    self._play_args = None

  # ------------------------------------------------------------------ helpers

  def _stream(self):
    return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def _state(self):
    return _hip.CampxState(_ptr(self.pos), _ptr(self.done), _ptr(self.ret),
                           _ptr(self._pair_table))

  def _observation(self, obs, board):
    layers = {ch: obs[:, i] for i, ch in enumerate(self.chars)}
    return Observation(board=board, layers=layers, layered_board=obs)

  def _action_ids(self, actions, expect):
    """Normalise to int8 ids on the device; one-hot floats go through the kernel."""
    if not torch.is_tensor(actions):
      actions = torch.as_tensor(actions)
    actions = actions.to(self.device)
    if actions.is_floating_point():
      if actions.shape != tuple(expect) + (gamespec.N_ACTIONS,):
        raise ValueError('one-hot actions must have shape {}, got {}'.format(
            tuple(expect) + (gamespec.N_ACTIONS,), tuple(actions.shape)))
      onehot = actions.to(torch.float32).contiguous()
      ids = torch.empty(expect, dtype=torch.int8, device=self.device)
      self._bad.zero_()
      with torch.cuda.device(self.device):
        _hip.check(_hip.lib.campx_onehot_to_ids_launch(
            _ptr(onehot), _ptr(ids), ids.numel(), _ptr(self._bad),
            self._stream()), 'campx_onehot_to_ids_launch')
      if self.validate_actions and int(self._bad.item()):
        # the reference asserts sum(act) == 1 (examples/boat_race.py:48)
        raise ValueError('{} action rows are not exactly one-hot'.format(
            int(self._bad.item())))
      return ids
    if tuple(actions.shape) != tuple(expect):
      raise ValueError('action ids must have shape {}, got {}'.format(
          tuple(expect), tuple(actions.shape)))
    ids = actions.to(torch.int8).contiguous()
    if self.validate_actions:
      self._bad.zero_()
      with torch.cuda.device(self.device):
        _hip.check(_hip.lib.campx_check_actions_launch(
            _ptr(ids), ids.numel(), _ptr(self._bad), self._stream()),
            'campx_check_actions_launch')
      if int(self._bad.item()):
        raise ValueError('{} action ids are outside 0..{}'.format(
            int(self._bad.item()), gamespec.N_ACTIONS - 1))
    return ids

  # --------------------------------------------------------------------- API

  def showtime(self):
    """its_showtime(): state from the art, first observation, reward None."""
    out = _hip.CampxOutputs(_ptr(self._obs), 0, _ptr(self._board), 0,
                            None, None, None, None, None)
    with torch.cuda.device(self.device):
      _hip.check(_hip.lib.campx_reset_launch(
          ctypes.byref(self.spec), _ptr(self._spec_dev), self._state(), out,
          self.batch, self._stream()), 'campx_reset_launch')
    self.frame = 0
    return self._observation(self._obs, self._board), None, 1.0

  def play(self, actions):
    # The per-call host path is kept short: the engine's own buffers never move, so
    # the argument structs and the returned Observation (views of those buffers)
    # are built once.
    if (self.validate_actions or not torch.is_tensor(actions)
        or actions.dtype != torch.int8 or actions.device != self.device
        or actions.shape != (self.batch,) or not actions.is_contiguous()):
      ids = self._action_ids(actions, (self.batch,))
    else:
      ids = actions
    if self._play_args is None or self._play_args[0] is not self.ret:
      out = _hip.CampxOutputs(_ptr(self._obs), 0, _ptr(self._board), 0,
                              _ptr(self._reward), _ptr(self._discount),
                              _ptr(self._step_done),
                              _ptr(self.perf) if self.has_perf else None, None)
      self._play_args = (self.ret, self._state(), out,
                         self._observation(self._obs, self._board),
                         ctypes.byref(self.spec), _ptr(self._spec_dev))
    _, state, out, observation, spec_ref, spec_dev = self._play_args
    if torch.cuda.current_device() != self.device.index:
      with torch.cuda.device(self.device):
        rc = _hip.lib.campx_rollout_launch(spec_ref, spec_dev, state, _ptr(ids), out,
                                           self.batch, 1, 0, self._stream())
    else:
      rc = _hip.lib.campx_rollout_launch(spec_ref, spec_dev, state, _ptr(ids), out,
                                         self.batch, 1, 0, self._stream())
    if rc:
      _hip.check(rc, 'campx_rollout_launch')
    self.frame += 1
    return observation, (self._reward if self.any_reward else None), self._discount

  def rollout(self, actions, obs=None, board=None, keep_obs=True,
              reset_first=False, want_board=False, obs_dtype=torch.int8):
    """T frames in one launch.

    Args:
      actions: int tensor [T, B] of action ids.
      obs: optional int8 [T, B, L, H, W] buffer to write every frame's
          layered board into (allocated if None and `keep_obs`).  With
          `keep_obs=False` every frame is still rendered and written, but into
          the engine's single-frame buffer, so only the last one survives.
      board: optional int8 [T, B, H, W] buffer for the flat boards.
      reset_first: rebuild all environments from the art before frame 0 (a new
          episode, as `make_game()` per episode in examples/reinforce.py:122).
      obs_dtype: torch.int8 (default), torch.float16 or torch.bfloat16.  The
          16-bit forms are the policy-network input the reference's driver makes
          with `layered_board.view(-1).float()` (examples/reinforce.py:123,149),
          written directly by the render kernel; they need `keep_obs` and a game
          that takes the two-kernel path.
    Returns:
      dict with 'obs' ([T,B,L,H,W] or the last frame [B,L,H,W]), 'board' (or
      None), 'reward' [T,B] (None if the game never rewards), 'discount' [T,B],
      'done' [T,B] uint8, 'perf' [T,B] int8 hidden performance (None unless the
      game declared one, `Engine.set_hidden_performance`), 'trace' (split path).
    """
    T = int(actions.shape[0])
    ids = self._action_ids(actions, (T, self.batch))
    B, L, H, W, dev = self.batch, self.n_layers, self.rows, self.cols, self.device
    formats = {torch.int8: 0, torch.float16: 1, torch.bfloat16: 2}
    if obs_dtype not in formats:
      raise ValueError('obs_dtype must be torch.int8, float16 or bfloat16')
    if keep_obs:
      if obs is None:
        obs = torch.empty((T, B, L, H, W), dtype=obs_dtype, device=dev)
      elif (tuple(obs.shape) != (T, B, L, H, W) or obs.dtype != obs_dtype
            or not obs.is_contiguous() or obs.device != dev):
        raise ValueError('obs must be a contiguous {} [T,B,L,H,W] tensor on {}'
                         .format(obs_dtype, dev))
      obs_stride = B * L * H * W
    else:
      if formats[obs_dtype]:
        raise ValueError('16-bit observations need keep_obs=True')
      obs, obs_stride = self._obs, 0
    if want_board and board is None:
      board = torch.empty((T, B, H, W), dtype=torch.int8, device=dev)
    board_stride = B * H * W if board is not None else 0
    reward = torch.empty((T, B), dtype=torch.float32, device=dev)
    discount = torch.empty((T, B), dtype=torch.float32, device=dev)
    done = torch.empty((T, B), dtype=torch.uint8, device=dev)
    perf = (torch.empty((T, B), dtype=torch.int8, device=dev)
            if self.has_perf else None)
    # The compact trajectory; giving it lets the library take its two-kernel path.
    # (Games with several movers interpret their rules per frame in one wave; for
    # them the single fused kernel is still the faster path unless forced.)
    split = (SPLIT_ROLLOUT and (self.uses_table or FORCE_SPLIT)) or bool(formats[obs_dtype])
    trace = (torch.empty((self.n_dyn, T, B), dtype=torch.int32, device=dev)
             if keep_obs and split else None)
    out = _hip.CampxOutputs(_ptr(obs), obs_stride, _ptr(board), board_stride,
                            _ptr(reward), _ptr(discount), _ptr(done), _ptr(perf),
                            _ptr(trace), formats[obs_dtype])
    with torch.cuda.device(self.device):
      _hip.check(_hip.lib.campx_rollout_launch(
          ctypes.byref(self.spec), _ptr(self._spec_dev), self._state(),
          _ptr(ids), out, B, T, int(bool(reset_first)), self._stream()),
          'campx_rollout_launch')
    self.frame = T if reset_first else self.frame + T
    return dict(obs=obs, board=board,
                reward=reward if self.any_reward else None,
                discount=discount, done=done, perf=perf, trace=trace)
