"""The wide tier: games run from their STATE table, B environments per launch.

The one-cell tier's kernels index their tables by the things' cells - (rows*cols)^K * 5
entries, 7-bit cells - which stops at 128 cells and grows fast with K.  `tabulate.trace()`
enumerates the states a game can actually reach (by running its own `update()` classes,
rule classes included), and the table over THOSE - one row per state, (state, action) ->
state - has no such limits: PyColab-sized boards (16x16 mazes and up; campx/engine.py:31
sets none), up to eight things that show, and whatever hidden values stand behind them (the
z-order in force, keys picked up, doors opened).  The observation stream is the same render
kernel, fed by a 16-bit trace (csrc/k_wide.hip, include/campx_hip.h CampxWideSpec).
`WideGame` is what a batched `Engine` delegates to for games the one-cell tier cannot take;
same surface as `fused.FusedGame` (showtime / reset / play / rollout / rollout_buffers /
check_actions, `done`, `ret`, `perf`; the dynamic state is `state`, int32 [B] state indices),
through the torch op `campx::wide_rollout`.  No CPU path: constructing one without a HIP
device raises.
"""

import ctypes

import torch

from . import _hip
from . import fused
from . import gamespec
from . import tabulate


class WideGame(fused.FusedGame):

  def __init__(self, engine, batch, device, traced):
    if not torch.cuda.is_available():
      raise RuntimeError(
          'the wide tier needs a HIP device (torch.cuda.is_available() is False) and has '
          'no CPU fallback; use batch=None for the single-environment generic tier')
    self.device = torch.device('cuda' if device is None else device)
    if self.device.type != 'cuda':
      raise ValueError('wide tier: device must be a HIP/cuda device, got {}'.format(self.device))
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    self.batch = int(batch)
    if self.batch < 1:
      raise ValueError('batch must be >= 1')
    self.traced = traced
    self.description = None
    self.spec, self._arrays = tabulate.to_wide_spec(traced)
    self.chars = list(traced.chars)
    _hip.check(_hip.lib.campx_wide_spec_validate(ctypes.byref(self.spec)),
               'campx_wide_spec_validate')
    self.rows, self.cols = engine.rows, engine.cols
    self.n_layers = len(self.chars)
    self.n_dyn = int(self.spec.n_dyn)      # (pieces of the scenery are not among the things)
    # planes of the trace: the things, plus - a scenery of several variants - which one shows,
    # or - a scenery of pieces that come and go - the mask of those that do
    self._n_planes = self.n_dyn + (1 if self.spec.n_variants > 1 or self.spec.n_pieces > 0 else 0)
    self.uses_table = True
    self.any_reward = bool(self.spec.any_reward)
    self.has_perf = bool(self.spec.has_perf)
    B, dev = self.batch, self.device
    row = self.n_layers * self.rows * self.cols
    if B * row >= (1 << 32) - 65536:
      raise ValueError('wide tier: a frame of {} environments x {} bytes does not fit 32-bit '
                       'offsets; use a smaller batch'.format(B, row))
    n = int(_hip.lib.campx_wide_tables_bytes(ctypes.byref(self.spec)))
    self._tables = torch.empty((n,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
      _hip.check(_hip.lib.campx_wide_tables_build(
          ctypes.byref(self.spec), ctypes.c_void_p(self._tables.data_ptr()),
          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
          'campx_wide_tables_build')
    # The launches read the spec's plain fields only; the blob they get carries no pointers
    # to the host arrays (which stay alive in self._arrays anyway).
    for name in ('state_cells', 'next_state', 'reward', 'done', 'perf', 'variant_top_layer', 'state_variant',
                 'state_pieces'):
      setattr(self.spec, name, None)
    self._spec_host = torch.frombuffer(bytearray(gamespec.spec_bytes(self.spec)),
                                       dtype=torch.uint8)
    self.state = torch.zeros((B,), dtype=torch.int32, device=dev)   # index into traced.st_*
    self.pos = None                   # (positions: see the trace)
    self.done = torch.zeros((B,), dtype=torch.uint8, device=dev)
    self.ret = torch.zeros((B,), dtype=torch.float32, device=dev)
    self._obs = torch.empty((B, self.n_layers, self.rows, self.cols), dtype=torch.int8, device=dev)
    self._board = torch.empty((B, self.rows, self.cols), dtype=torch.int8, device=dev)
    self._reward = torch.empty((B,), dtype=torch.float32, device=dev)
    self._discount = torch.empty((B,), dtype=torch.float32, device=dev)
    self._step_done = torch.empty((B,), dtype=torch.uint8, device=dev)
    self._step_trace = self._trace_rows(None)
    self.perf = torch.zeros((B,), dtype=torch.int8, device=dev)
    self._perf_arg = self.perf if self.has_perf else None
    self._bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    self._onehot_bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    self._bad_flag = torch.zeros((1,), dtype=torch.int32).pin_memory()
    self._bad_flag_view = self._bad_flag.numpy()
    self.validate_actions = True
    self.frame = -1
    self._observation_cache = self._observation(self._obs, self._board)
    self._wide = _hip.ops.wide_rollout.default

  def _trace_rows(self, T):
    """int16 [K, B] (one frame) or [K, T, B] trace buffer, rows padded like the other streams."""
    B = self.batch
    pitch = (B + 15) // 16 * 16 if fused.PAD_ROWS else B
    shape = (self._n_planes, pitch) if T is None else (self._n_planes, T, pitch)
    return torch.empty(shape, dtype=torch.int16, device=self.device)[..., :B]

  # --------------------------------------------------------------------- API

  def showtime(self):
    """its_showtime(): state from the art, first observation, reward None."""
    first = self._obs
    if first.dtype != torch.int8:       # set_play_obs_dtype(): the first frame is rendered as int8
      first = torch.empty(self._obs.shape, dtype=torch.int8, device=self.device)
    self._wide(self._spec_host, self._tables, self.state, self.done, self.ret, None, first,
               self._board, None, None, None, None, self._step_trace, None, None, False)
    if first is not self._obs:
      self._obs.copy_(first)
    self.frame = 0
    return self._observation_cache, None, 1.0

  def reset(self):
    return self.showtime()

  def play(self, actions):
    if (torch.is_tensor(actions) and actions.dtype == torch.int8 and actions.device == self.device
        and actions.shape == (self.batch,) and actions.is_contiguous()):
      ids = actions
    else:
      ids = self._action_ids(actions, (self.batch,))
    validate = self.validate_actions
    self._wide(self._spec_host, self._tables, self.state, self.done, self.ret, ids, self._obs,
               self._board, self._reward, self._discount, self._step_done, self._perf_arg,
               self._step_trace, self._bad if validate else None,
               self._bad_flag if validate else None, False)
    self.frame += 1
    if validate:
      self._after_launch()
    return (self._observation_cache, (self._reward if self.any_reward else None), self._discount)

  def rollout_buffers(self, T, keep_obs=True, want_board=False, obs_dtype=torch.int8, share=None):
    out = super(WideGame, self).rollout_buffers(T, keep_obs, want_board, obs_dtype, share)
    out['trace'] = self._trace_rows(T)
    return out

  def rollout_deferred(self, actions, out, reset_first=False, actions_ready=False):
    """`FusedGame.rollout_deferred` for this tier, which has no shared launch: the rollout is
    run whole, at once, and `out` is simply complete a call early; returns the previous call's
    dict (None on the first)."""
    prev = getattr(self, '_deferred', None)
    if prev is not None and prev['obs'].data_ptr() == out['obs'].data_ptr():
      raise ValueError('the state-table tier runs a deferred rollout whole, at once: two dicts over ONE '
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
    """T frames in one call (see `fused.FusedGame.rollout`; `pipelined` is not offered here).
    'trace' in the result is int16 [K, T, B]: cell | covered scenery layer << 10 | shows << 15."""
    if pipelined:
      raise ValueError('wide tier: pipelined rollouts are not offered')
    T = int(actions.shape[0])
    if T < 1:
      raise ValueError('a rollout needs at least one frame: actions [T, B] with T >= 1')
    if (torch.is_tensor(actions) and actions.dtype == torch.int8 and actions.device == self.device
        and actions.shape == (T, self.batch) and actions.is_contiguous()):
      ids = actions
    else:
      ids = self._action_ids(actions, (T, self.batch))
    if out is None:
      B, L, H, W, dev = self.batch, self.n_layers, self.rows, self.cols, self.device
      if obs is not None and keep_obs and (
          tuple(obs.shape) != (T, B, L, H, W) or obs.dtype != obs_dtype
          or not obs.is_contiguous() or obs.device != dev):
        raise ValueError('obs must be a contiguous {} [T,B,L,H,W] tensor on {}'.format(obs_dtype, dev))
      out = self.rollout_buffers(T, keep_obs, want_board or board is not None, obs_dtype)
      if obs is not None and keep_obs:
        out['obs'] = obs
      if board is not None:
        out['board'] = board
    validate = self.validate_actions
    self._wide(self._spec_host, self._tables, self.state, self.done, self.ret, ids, out['obs'],
               out['board'], out['reward'], out['discount'], out['done'], out['perf'],
               out['trace'], self._bad if validate else None,
               self._bad_flag if validate else None, bool(reset_first))
    self.frame = T if reset_first else self.frame + T
    if validate:
      self._after_launch()
    return out
