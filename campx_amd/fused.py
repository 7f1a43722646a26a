"""The fused tier: B environments stepped by the HIP kernels.

`FusedGame` is what a batched `Engine` delegates to after `its_showtime()`.  It
lowers the engine to a GameSpec (`gamespec`), keeps the dynamic state and the
output buffers as torch tensors in HBM, and advances them with the torch custom
ops `campx::reset / step / rollout` (csrc/campx_torch.cpp), which unpack the
tensors into the C ABI of libcampx_hip.so (include/campx_hip.h) on torch's
current HIP stream.  PyTorch is used for device memory, streams and op dispatch
only; every per-frame computation is inside the kernels.

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

Deviations from the reference, inherent to batching:

* game-over is per environment.  Instead of raising on `play()` after termination
  (campx/engine.py:149-151) a finished environment is rebuilt from the art before
  its next action is applied; `done` (also `game.fused.done`) tells which ones ended.
* the reference asserts `sum(act) == 1` inside the agent's update
  (examples/boat_race.py:48).  Here the kernel that reads the action ids counts the
  ones outside 0..4 (they act as "stay") and raises a flag in host-mapped memory;
  `validate_actions=True` (default) makes `play()` / `rollout()` look at that flag -
  a plain host read, no stream synchronisation - so the ValueError surfaces on the
  first call after the GPU has consumed the bad ids; `validate_actions='sync'`
  synchronises and raises in the offending call; `False` never looks.
  `check_actions()` synchronises and raises on demand.
"""

import ctypes
import os

import torch

from . import _hip
from . import gamespec
from .rendering import Observation


# Tabulate the update pass at showtime - per (cell, action) for one mover
# (campx_spec_compile), per (cell, ..., cell, action) for two to four
# (campx_pair_table_build) - and let the frame loop look it up.  Tests switch this off to
# exercise the rule interpreter on the same games.
COMPILE_TABLE = True
# Rollouts that keep every frame run the update pass and the render as two kernels
# (needs a [K, T, B] uint8 trace buffer; NOTES.md "Kernels", profiles/): faster for
# every game, tabulated update pass or interpreted.  False keeps everything in
# the single fused kernel; parity tests run both.
SPLIT_ROLLOUT = True
FORCE_SPLIT = SPLIT_ROLLOUT   # kept for callers that toggled it: same as SPLIT_ROLLOUT now
# Pad the rows of the per-frame streams to a multiple of 16 elements (see rollout_buffers).
# (False keeps them back to back: B = 65 535 then costs 47 instead of 16 us per 100 frames.)
PAD_ROWS = True

_OBS_DTYPES = (torch.int8, torch.float16, torch.bfloat16)


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)




class FusedGame(object):

  def __init__(self, engine, batch, device=None, traced=None):
    """`traced`: a `tabulate.TracedGame` - the update pass of a game whose entities are
    arbitrary Python classes, tabulated on the host by running them on the generic tier;
    without it the engine's entities must be `campx_amd.rules` classes and the tables are
    built on the device by the rule interpreter."""
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
    self.traced = traced
    if traced is not None:
      from . import tabulate
      self.description = None
      self.spec = tabulate.to_spec(traced)
      self.chars = list(traced.chars)
    else:
      self.description = gamespec.describe(engine)
      self.spec = gamespec.lower(self.description)
      self.chars = list(self.description.chars)
    _hip.check(_hip.lib.campx_spec_validate(ctypes.byref(self.spec)),
               'campx_spec_validate')
    with torch.cuda.device(self.device):
      _hip.check(_hip.lib.campx_spec_compile(
          ctypes.byref(self.spec),
          ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
          'campx_spec_compile')
    if not COMPILE_TABLE and traced is None:
      self.spec.table_valid = 0     # keep the render tables, interpret the rules
    self.uses_table = bool(self.spec.table_valid)
    self.rows, self.cols = engine.rows, engine.cols
    self.n_layers = len(self.chars)
    self.n_dyn = self.spec.n_dyn
    self.any_reward = bool(self.spec.any_reward)
    self.has_perf = self.spec.perf_dyn >= 0

    B, dev = self.batch, self.device
    # The GameSpec blob twice: on the host (the library reads it to choose and
    # parameterise kernels) and on the device (the kernels' tables).
    self._spec_host = torch.frombuffer(bytearray(gamespec.spec_bytes(self.spec)),
                                       dtype=torch.uint8)
    self._spec_dev = self._spec_host.to(dev)
    # Games with two to four movers: the (cell, ..., cell, action) table of the update pass.
    self._pair_table = None
    n_pair = int(_hip.lib.campx_pair_table_bytes(ctypes.byref(self.spec)))
    if traced is not None and self.n_dyn >= 2:
      if n_pair <= 0:
        raise ValueError('fused tier: the state table of this game ({} moving things on {} '
                         'cells) is too large'.format(self.n_dyn, self.rows * self.cols))
      import numpy as np
      table = torch.empty((n_pair,), dtype=torch.uint8, device=dev)
      trace = np.ascontiguousarray(traced.trace_bytes())
      reward = np.ascontiguousarray(traced.reward, dtype=np.float32)
      done = np.ascontiguousarray(traced.done_bytes(), dtype=np.uint8)
      perf = np.ascontiguousarray(traced.perf, dtype=np.int8)
      with torch.cuda.device(self.device):
        _hip.check(_hip.lib.campx_pair_table_pack(
            ctypes.byref(self.spec), trace.ctypes.data, reward.ctypes.data, done.ctypes.data,
            perf.ctypes.data if self.has_perf else None, _ptr(table),
            ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
            'campx_pair_table_pack')
      self._pair_table = table
    elif COMPILE_TABLE and n_pair > 0:
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
    self._perf_arg = self.perf if self.has_perf else None
    # Bad-action bookkeeping: a device counter and a flag in pinned (device-mapped)
    # host memory that the kernels set and the host reads without synchronising.
    self._bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    self._onehot_bad = torch.zeros((1,), dtype=torch.int32, device=dev)   # never read
    self._bad_flag = torch.zeros((1,), dtype=torch.int32).pin_memory()
    self._bad_flag_view = self._bad_flag.numpy()
    self.validate_actions = True
    self.frame = -1
    self._observation_cache = self._observation(self._obs, self._board)
    self._step = _hip.ops.step.default
    self._rollout = _hip.ops.rollout.default
    self._update = _hip.ops.update.default
    self._render = _hip.ops.render.default
    self._update_render = _hip.ops.update_render.default
    self._rollout_pipelined = _hip.ops.rollout_pipelined.default
    self._flow_scratch = None
    self._flow_state = None    # the scratch block's CampxFlowState: host memory, ours (int64[6])
    self._flow_shared = {}     # (T, pitch) -> whether the library runs such a rollout as one launch
    # what a launch could not do although its call returned: bits the kernels raise in pinned
    # host memory (include/campx_hip.h CampxOutputs.error_flag), looked at after EVERY launch
    self._err_flag = torch.zeros((1,), dtype=torch.int32).pin_memory()
    self._err_flag_view = self._err_flag.numpy()
    self._deferred = None      # rollout_deferred(): the dict whose observations are still owed
    self._deferred_rendered = False    # ... unless that rollout was run whole (no shared launch)
    self._shared_launch = {}   # T -> whether campx_update_render_launch shares one launch
    # pipelined rollouts (campx::rollout_pipelined keeps the side stream): whether that stream
    # already comes after everything this game issued on the caller's stream
    self._aux_in_sync = False

  # ------------------------------------------------------------------ helpers

  def _observation(self, obs, board):
    layers = {ch: obs[:, i] for i, ch in enumerate(self.chars)}
    return Observation(board=board, layers=layers, layered_board=obs)

  def _raise_bad(self):
    n = int(self._bad.item())          # synchronises: we are about to raise anyway
    self._bad.zero_()
    self._bad_flag_view[0] = 0
    if n:
      raise ValueError('{} action ids are outside 0..{} (or came from rows that are not '
                       'exactly one-hot)'.format(n, gamespec.N_ACTIONS - 1))

  def check_actions(self):
    """Synchronise and raise ValueError if any consumed action id was outside 0..4 (and
    RuntimeError if any launch so far raised its error word: after the synchronise that is
    definitive for every rollout issued)."""
    torch.cuda.synchronize(self.device)
    self.check_ok()
    self._raise_bad()

  def check_ok(self):
    """Raise RuntimeError if a launch reported, through the error word, that it could not do what
    it was asked to.  A plain host read of pinned memory: right after an asynchronous launch it
    can only show what an EARLIER launch raised (the rollout named in the message may be a
    previous one); after a synchronise - `check_actions()`, `flush()` of the last deferred
    rollout followed by one - it is definitive.  Called after every rollout launch, whatever
    `validate_actions` says."""
    bits = int(self._err_flag_view[0])
    if bits:
      self._err_flag_view[0] = 0
      what = []
      if bits & _hip.ERR_FLOW_TIMEOUT:
        what.append('a render wave of a one-launch rollout (pipe_table_kernel<true>) waited for '
                    'its own launch\'s trace entries until it gave up and wrote frames from '
                    'stale ones: the observations of that rollout - this call\'s or an earlier '
                    'one\'s, the flag is read without synchronising - are WRONG (two launches '
                    'sharing one scratch block at the same time?)')
      if bits & ~_hip.ERR_FLOW_TIMEOUT:
        what.append('error bits {:#x}'.format(bits & ~_hip.ERR_FLOW_TIMEOUT))
      raise RuntimeError('campx: ' + '; '.join(what))

  def _after_launch(self):
    mode = self.validate_actions
    if mode == 'sync':
      self._raise_bad()
    elif mode and self._bad_flag_view[0]:
      self._raise_bad()

  def _action_ids(self, actions, expect):
    """Normalise to int8 ids on the device; one-hot floats go through a kernel."""
    if (isinstance(actions, (list, tuple)) and len(actions) == gamespec.N_ACTIONS and len(expect) == 1
        and all(isinstance(x, (int, float)) and x in (0, 1) for x in actions) and sum(actions) == 1):
      # the reference's plain one-hot LIST, `game.play([1, 0, 0, 0, 0])` (Demo 1 cell 6): one
      # action for every environment.  (Five ENVIRONMENTS' ids go in as a tensor.)
      if expect[0] == gamespec.N_ACTIONS and not any(isinstance(x, float) for x in actions):
        raise ValueError(
            'with batch == 5 a list of five 0/1 integers is ambiguous (one one-hot action for '
            'every environment, or five action ids): pass a float list / tensor for a one-hot '
            'action, an integer tensor for ids')
      actions = torch.tensor(actions, dtype=torch.float32)
    if not torch.is_tensor(actions):
      actions = torch.as_tensor(actions)
    actions = actions.to(self.device)
    # ONE action - the reference's `play(one_hot[5])` / `play(int)` call, as a driver or a
    # notebook cell written for a single environment makes it - goes to every environment
    if len(expect) == 1:
      if actions.is_floating_point() and tuple(actions.shape) == (gamespec.N_ACTIONS,):
        actions = actions.expand(expect[0], gamespec.N_ACTIONS)
      elif not actions.is_floating_point() and actions.dim() == 0:
        actions = actions.expand(expect[0])
    if actions.is_floating_point():
      if actions.shape != tuple(expect) + (gamespec.N_ACTIONS,):
        raise ValueError('one-hot actions must have shape {}, got {}'.format(
            tuple(expect) + (gamespec.N_ACTIONS,), tuple(actions.shape)))
      onehot = actions.to(torch.float32).contiguous()
      ids = torch.empty(expect, dtype=torch.int8, device=self.device)
      if self.validate_actions == 'sync':
        count = torch.zeros((1,), dtype=torch.int32, device=self.device)
        _hip.ops.onehot_to_ids(onehot, ids, count)
        if int(count.item()):
          # the reference asserts sum(act) == 1 (examples/boat_race.py:48)
          raise ValueError('{} action rows are not exactly one-hot'.format(
              int(count.item())))
      else:
        # a row that is not exactly one-hot becomes id 5, which the kernel that consumes
        # it reports like any other bad id (lazy validation) - no synchronisation here
        _hip.ops.onehot_to_ids(onehot, ids, self._onehot_bad)
      return ids
    if tuple(actions.shape) != tuple(expect):
      raise ValueError('action ids must have shape {}, got {}'.format(
          tuple(expect), tuple(actions.shape)))
    if actions.dtype != torch.int8:
      # narrow without wrapping: 256 must not turn into 0 ("left")
      actions = actions.clamp(-1, gamespec.N_ACTIONS).to(torch.int8)
    return actions.contiguous()

  # --------------------------------------------------------------------- API

  def showtime(self):
    """its_showtime(): state from the art, first observation, reward None."""
    _hip.ops.reset(self._spec_host, self._spec_dev, self.pos, self.done, self.ret,
                   self._pair_table, self._obs, self._board)
    self.frame = 0
    return self._observation_cache, None, 1.0

  def reset(self):
    """A new episode for every environment: what `make_game()` + `its_showtime()` per
    episode does in the reference's driver (examples/reinforce.py:122).  Returns the
    first `(Observation, None, 1.0)` like `its_showtime()`."""
    if self._obs.dtype != torch.int8:
      first = torch.empty(self._obs.shape, dtype=torch.int8, device=self.device)
      _hip.ops.reset(self._spec_host, self._spec_dev, self.pos, self.done, self.ret,
                     self._pair_table, first, self._board)
      self._obs.copy_(first)
      self.frame = 0
      return self._observation_cache, None, 1.0
    return self.showtime()

  def set_play_obs_dtype(self, dtype):
    """Make `play()` return `layered_board` in `dtype` (torch.int8, float16 or bfloat16).

    The 16-bit forms are what the reference's driver builds per step with
    `board.layered_board.view(-1).float()` (examples/reinforce.py:149) to feed its policy:
    here the step kernel writes them directly (0.0 / 1.0), no conversion pass.  Needs a
    game whose update pass is tabulated; the flat `board` stays int8.
    """
    if dtype not in _OBS_DTYPES:
      raise ValueError('obs dtype must be torch.int8, float16 or bfloat16')
    if dtype != torch.int8 and not self.uses_table:
      raise ValueError('16-bit play() observations need a game with a tabulated update pass')
    if dtype != self._obs.dtype:
      shown = self._obs.to(dtype)        # keep the frame currently shown
      self._obs = shown
      self._observation_cache = self._observation(self._obs, self._board)

  def play(self, actions):
    # The per-call host path is kept short: the engine's own buffers never move, so
    # the returned Observation (views of those buffers) is built once, and ids that
    # already are an int8 [B] device tensor go straight to the op.
    if (torch.is_tensor(actions) and actions.dtype == torch.int8
        and actions.device == self.device and actions.shape == (self.batch,)
        and actions.is_contiguous()):
      ids = actions
    else:
      ids = self._action_ids(actions, (self.batch,))
    validate = self.validate_actions
    self._aux_in_sync = False
    self._step(self._spec_host, self._spec_dev, self.pos, self.done, self.ret,
               self._pair_table, ids, self._obs, self._board, self._reward,
               self._discount, self._step_done, self._perf_arg,
               self._bad if validate else None,
               self._bad_flag if validate else None)
    self.frame += 1
    if validate:
      self._after_launch()
    return (self._observation_cache,
            (self._reward if self.any_reward else None), self._discount)

  def capture_play(self, n_frames, policy=None, record_obs=False):
    """`n_frames` consecutive `play()` calls - and, with `policy(observation, t) -> ids [B]`, the
    policy's forward pass and sampling in front of each - captured once in a HIP graph;
    `.replay()` then runs them with one launch of the host's (campx_amd/play_graph.py)."""
    from .play_graph import PlayGraph
    return PlayGraph(self, n_frames, policy=policy, record_obs=record_obs)

  def _one_launch(self, T, pitch):
    """Whether the library runs a T-frame rollout of this game, rows `pitch` apart, as ONE
    launch (campx_flow_shared: its own bounds and knobs, asked once per shape)."""
    key = (T, pitch)
    got = self._flow_shared.get(key)
    if got is None:
      got = self._flow_shared[key] = bool(_hip.lib.campx_flow_shared(
          ctypes.byref(self.spec), self.batch, T, pitch))
    return got

  def _scratch(self, T, out):
    """(scratch, scratch_state) = CampxOutputs.overlap_ctl / flow_state for this rollout: the
    tagged copy of the trace that lets a one-mover game's rollout run as ONE launch, and the
    block's tag state (host memory, ours).  (None, None) where the library would not use them -
    at B = 65 536, T = 4 000 the block would be half a gigabyte."""
    trace = out.get('trace')
    if trace is None or out['obs'].dim() != 5 or out['obs'].dtype != torch.int8:
      return None, None
    if not self._one_launch(T, trace.stride(1)):
      return None, None
    need = (int(_hip.lib.campx_flow_scratch_bytes(self.batch, T)) + 3) // 4
    if self._flow_scratch is None or self._flow_scratch.numel() < need:
      self._flow_scratch = torch.zeros(need, dtype=torch.int32, device=self.device)
      self._flow_state = torch.zeros(ctypes.sizeof(_hip.CampxFlowState) // 8, dtype=torch.int64)
    return self._flow_scratch, self._flow_state

  def rollout_buffers(self, T, keep_obs=True, want_board=False,
                      obs_dtype=torch.int8, share=None):
    """Allocate the output buffers of a T-frame rollout once, for `rollout(out=...)`.

    `share`: another such dict whose 'obs' / 'board' tensors this one reuses (the
    second buffer set of pipelined rollouts needs its own scalars and trace only).
    """
    B, L, H, W, dev = self.batch, self.n_layers, self.rows, self.cols, self.device
    if obs_dtype not in _OBS_DTYPES:
      raise ValueError('obs_dtype must be torch.int8, float16 or bfloat16')
    sixteen = obs_dtype != torch.int8
    if sixteen and not keep_obs:
      raise ValueError('16-bit observations need keep_obs=True')
    if not keep_obs and self._obs.dtype != torch.int8:
      raise ValueError('keep_obs=False renders into play()\'s frame buffer, which '
                       'set_play_obs_dtype() made {}: set it back to int8 first'
                       .format(self._obs.dtype))
    if share is not None:
      obs, board = share['obs'], share['board']
    else:
      obs = (torch.empty((T, B, L, H, W), dtype=obs_dtype, device=dev)
             if keep_obs else self._obs)
      board = None
      if want_board:
        board = (torch.empty((T, B, H, W), dtype=torch.int8, device=dev)
                 if keep_obs else self._board)
    # The compact trajectory; giving it lets the library take its two-kernel path.
    split = SPLIT_ROLLOUT or sixteen
    # The per-frame streams are [T, B] VIEWS of [T, pitch] arrays, pitch = B rounded up to a
    # multiple of 16 (CampxOutputs.scalar_pitch): with a batch size that is not one, every
    # row still starts 16-byte aligned and the update kernels store whole aligned groups
    # (B = 65 535: 47 -> 16 us per 100 frames).  For B % 16 == 0 they are plain contiguous.
    pitch = (B + 15) // 16 * 16 if PAD_ROWS else B

    def rows(dtype, *lead):
      return torch.empty(lead + (T, pitch), dtype=dtype, device=dev)[..., :B]
    return dict(
        obs=obs, board=board,
        reward=rows(torch.float32) if self.any_reward else None,
        discount=rows(torch.float32),
        done=rows(torch.uint8),
        perf=rows(torch.int8) if self.has_perf else None,
        trace=rows(torch.uint8, self.n_dyn) if split else None)

  def rollout(self, actions, obs=None, board=None, keep_obs=True,
              reset_first=False, want_board=False, obs_dtype=torch.int8,
              out=None, pipelined=False):
    """T frames in one launch.

    Args:
      actions: int tensor [T, B] of action ids.
      obs: optional int8 [T, B, L, H, W] buffer to write every frame's
          layered board into (allocated if None and `keep_obs`).  With
          `keep_obs=False` only the last frame is kept, in the engine's single-frame
          buffer (the two-kernel path renders just that one; the single fused kernel
          writes every frame over the previous one).
      board: optional int8 [T, B, H, W] buffer for the flat boards.
      reset_first: rebuild all environments from the art before frame 0 (a new
          episode, as `make_game()` per episode in examples/reinforce.py:122).
      obs_dtype: torch.int8 (default), torch.float16 or torch.bfloat16.  The
          16-bit forms are the policy-network input the reference's driver makes
          with `layered_board.view(-1).float()` (examples/reinforce.py:123,149),
          written directly by the render kernel; they need `keep_obs` and a game
          that takes the two-kernel path.
      out: a dict from `rollout_buffers()` (or a previous `rollout()` of the same
          T and options) whose tensors are overwritten instead of allocating new
          ones: the whole call is then one op dispatch.  Overrides
          obs/board/keep_obs/want_board/obs_dtype.
      pipelined: run this call's update pass (a short latency-bound kernel) on a side
          stream so that it overlaps the observation stream of the PREVIOUS rollout,
          which is still running on the current stream; the render kernel then waits
          for it by event, and an update pass waits for the render that last read the
          trace buffer it is about to overwrite.  Results are identical; the caller
          promises two things the engine cannot check: (1) `actions` are ready - not
          being produced by work still queued on the current stream - and (2) `out` is
          not the dict of the previous pipelined call (alternate two `rollout_buffers()`;
          they may share `obs`), and its per-frame scalars have been consumed before it
          comes round again.  Needs the two-kernel path (`out['trace']`).  Open-loop action
          streams (random exploration, scripted or replayed episodes) are the use.
    Returns:
      dict with 'obs' ([T,B,L,H,W] or the last frame [B,L,H,W]), 'board' (or
      None), 'reward' [T,B] (None if the game never rewards), 'discount' [T,B],
      'done' [T,B] uint8, 'perf' [T,B] int8 hidden performance (None unless the
      game declared one, `Engine.set_hidden_performance`), 'trace' (split path:
      uint8 [K,T,B], cell | visible << 7 per moving thing).
    """
    T = int(actions.shape[0])
    if T < 1:
      raise ValueError('a rollout needs at least one frame: actions [T, B] with T >= 1')
    if (torch.is_tensor(actions) and actions.dtype == torch.int8
        and actions.device == self.device and actions.shape == (T, self.batch)
        and actions.is_contiguous()):
      ids = actions
    else:
      ids = self._action_ids(actions, (T, self.batch))
    if out is None:
      B, L, H, W, dev = self.batch, self.n_layers, self.rows, self.cols, self.device
      if obs is not None and keep_obs and (
          tuple(obs.shape) != (T, B, L, H, W) or obs.dtype != obs_dtype
          or not obs.is_contiguous() or obs.device != dev):
        raise ValueError('obs must be a contiguous {} [T,B,L,H,W] tensor on {}'
                         .format(obs_dtype, dev))
      out = self.rollout_buffers(T, keep_obs, want_board or board is not None,
                                 obs_dtype)
      if obs is not None and keep_obs:
        out['obs'] = obs
      if board is not None:
        out['board'] = board
    validate = self.validate_actions
    if pipelined:
      if out['trace'] is None:
        raise ValueError('pipelined rollouts need the two-kernel path (a trace buffer: '
                         'keep_obs=True on a game whose update pass is tabulated)')
      # one op does the two-stream choreography (csrc/campx_torch.cpp rollout_pipelined: the
      # update pass on a high-priority side stream, the render on this stream behind it; from
      # Python the same calls cost the host 36-46 us and are gone, as is the side stream confined
      # to a subset of the compute units - measured slower, profiles/r03_cumask_ab.txt).
      # (ids made from the caller's actions by kernels on THIS stream - clamp / to(int8) /
      # onehot_to_ids - are work the side stream has to come after: resync for this call)
      self._rollout_pipelined(self._spec_host, self._spec_dev, self.pos, self.done, self.ret,
                              self._pair_table, ids, out['obs'], out['board'], out['reward'],
                              out['discount'], out['done'], out['perf'], out['trace'],
                              self._bad if validate else None,
                              self._bad_flag if validate else None, bool(reset_first),
                              not self._aux_in_sync or ids is not actions or pipelined == 'resync')
      self._aux_in_sync = True
    else:
      self._aux_in_sync = False
      self._rollout(self._spec_host, self._spec_dev, self.pos, self.done, self.ret,
                    self._pair_table, ids, out['obs'], out['board'], out['reward'],
                    out['discount'], out['done'], out['perf'], out['trace'],
                    self._bad if validate else None,
                    self._bad_flag if validate else None, bool(reset_first),
                    *(self._scratch(T, out) + (self._err_flag,)))
    self.frame = T if reset_first else self.frame + T
    self.check_ok()
    if validate:
      self._after_launch()
    return out

  def rollout_deferred(self, actions, out, reset_first=False, actions_ready=False):
    """Rollouts pipelined across calls: T frames of update pass now, their observations with
    the NEXT call.

    One launch holds this rollout's update pass (Engine.play() x T, campx/engine.py:145-222,
    without rendering) and the render pass (campx/engine.py:286-324) of the rollout handed to
    the previous `rollout_deferred()` call - two pieces of work that do not depend on each
    other, so the latency chain of the one hides under the observation stream of the other
    (`campx_update_render_launch`, include/campx_hip.h).  After the call `out`'s per-frame
    scalars and trace are this rollout's; its 'obs' are written by the next call, or by
    `flush()`.  For callers whose next actions do not wait for those observations (open-loop
    action streams: random exploration, scripted or replayed episodes).

    Where the library has no shared launch for the game or the sizes (a multi-mover game without
    its table or past its bounds, more than 32 768 environments or ~2 GB of observations per
    rollout, batches whose frames are not whole 16-byte chunks: `campx_update_render_shared`),
    the rollout is run whole at once - deferring would only make its render read a trace gone
    cold - and `out` is simply complete a call early; unless `out` shares its observation buffer
    with the previous call's dict, which the caller is about to read: then the update pass runs
    now and the render kernel with the next call, as everywhere else.

    Args:
      actions: int tensor [T, B] of action ids.
      out: a dict from `rollout_buffers(T)` (every frame kept, no flat board); not the dict
          of the previous call - alternate two of them.  They may share 'obs'
          (`rollout_buffers(T, share=first)`): a rollout's observations are complete after the
          next call and, shared, overwritten by the one after that.
      actions_ready: the caller's promise that `actions` and `out` are not being produced or
          read by work still queued on the current stream (the two promises of
          `rollout(pipelined=True)`): only then may a multi-mover game past the shared launch's
          bounds run its update pass on the side stream, AHEAD of what the current stream holds.
          Without it (the default) everything this call issues is in order on the current stream
          (round 5 took the two-stream route silently: actions made by a `torch.randint` on the
          current stream could be read before they were written).
    Returns:
      the previous call's dict, whose 'obs' this launch completes; None on the first call.
    """
    T = int(actions.shape[0])
    if T < 1:
      raise ValueError('a rollout needs at least one frame: actions [T, B] with T >= 1')
    if (torch.is_tensor(actions) and actions.dtype == torch.int8
        and actions.device == self.device and actions.shape == (T, self.batch)
        and actions.is_contiguous()):
      ids = actions
    else:
      ids = self._action_ids(actions, (T, self.batch))
    if out.get('trace') is None or out.get('board') is not None or out['obs'].dim() != 5:
      raise ValueError('deferred rollouts need rollout_buffers(T) of the two-kernel path: every '
                       'frame kept, a trace buffer, no flat board')
    prev = self._deferred
    if prev is not None and (prev is out or prev['trace'].data_ptr() == out['trace'].data_ptr()):
      raise ValueError('`out` is the dict whose observations are still to be rendered: '
                       'alternate two rollout_buffers() (they may share `obs`: '
                       'rollout_buffers(T, share=first))')
    validate = self.validate_actions
    one_launch = self._shared_launch.get(T)
    if one_launch is None:
      one_launch = self._shared_launch[T] = bool(
          (self.n_dyn == 1 or self._pair_table is not None) and
          _hip.lib.campx_update_render_shared(ctypes.byref(self.spec), self.batch, T))
    head = (self._spec_host, self._spec_dev, self.pos, self.done, self.ret, self._pair_table, ids,
            out['reward'], out['discount'], out['done'], out['perf'], out['trace'],
            self._bad if validate else None, self._bad_flag if validate else None,
            bool(reset_first))
    if not one_launch or out['obs'].dtype != torch.int8:
      # Nothing to gain from deferring (a game without its pair / tuple table, a batch or a rollout
      # too big for the shared launch, 16-bit observations).
      if prev is not None and prev['obs'].data_ptr() == out['obs'].data_ptr():
        # ... but the two dicts share their observation buffer, and the caller reads the PREVIOUS
        # rollout's observations from it after this call: this rollout's render has to wait for
        # the next call, as the contract says (update pass now; render kernel then).  (Until round
        # 5 this case rendered at once and handed back a dict whose observations were already the
        # new rollout's - tests/test_random_warehouses.py found it.)
        if not self._deferred_rendered:
          self._render(self._spec_host, self._spec_dev, prev['trace'], prev['obs'], None)
        self._aux_in_sync = False
        self._update(*head)
        self._deferred, self._deferred_rendered = out, False
        self.frame = T if reset_first else self.frame + T
        self.check_ok()
        if validate:
          self._after_launch()
        return prev
      # Separate observation buffers: the whole rollout now, rendered while its trace is still
      # cached.  `out` is complete a call early; what the caller sees is the same.
      self.flush()
      if (actions_ready and out['obs'].dtype == torch.int8 and
          self.batch > 8192 and (self.n_dyn >= 3 or self.batch < 32768)):
        # Games of two to four movers past the shared launch's bounds (16 384 environments with
        # two movers, 8 192 with more): the update pass on the (high-priority) side stream, under
        # the render of the rollout before it - two kernels on two streams instead of two roles of
        # one launch, one op (campx::rollout_pipelined).  Sokoban with two / three boxes, of peak,
        # in order -> two streams: B = 16 384 0.58 -> 0.70 / 0.60 -> 0.75, 32 768 0.72 -> 0.85 / 0.72 ->
        # 0.84, 65 536 0.81 -> 0.77-0.85 / 0.81 -> 0.86; the two-mover game gains nothing from 32 768
        # up (0.73 -> 0.74, 0.83 -> 0.81) and stays in order (profiles/r05_multimover_deferred_ab.txt).
        # Complete a call early, like the rollout below.
        # (ids this call made from the caller's actions, on the caller's stream: the side stream waits)
        self.rollout(ids, out=out, reset_first=reset_first,
                     pipelined='resync' if ids is not actions else True)
        self._deferred, self._deferred_rendered = out, True
        return prev
      self._aux_in_sync = False
      self._rollout(self._spec_host, self._spec_dev, self.pos, self.done, self.ret,
                    self._pair_table, ids, out['obs'], None, out['reward'], out['discount'],
                    out['done'], out['perf'], out['trace'], self._bad if validate else None,
                    self._bad_flag if validate else None, bool(reset_first),
                    *(self._scratch(T, out) + (self._err_flag,)))
      self._deferred, self._deferred_rendered = out, True
      self.frame = T if reset_first else self.frame + T
      self.check_ok()
      if validate:
        self._after_launch()
      return prev
    self._aux_in_sync = False
    # (a rollout of another length: its observations now, by the ordinary render kernel)
    share = (prev is not None and not self._deferred_rendered
             and tuple(prev['trace'].shape) == tuple(out['trace'].shape))
    if prev is not None and not share:
      self.flush()
    if share:
      self._update_render(*(head + (prev['trace'], prev['obs'])))
    else:
      self._update(*head)
    self._deferred, self._deferred_rendered = out, False
    self.frame = T if reset_first else self.frame + T
    self.check_ok()
    if validate:
      self._after_launch()
    return prev

  def flush(self):
    """Render the rollout `rollout_deferred()` still owes its observations; returns its dict
    (None if there is none)."""
    prev, self._deferred = self._deferred, None
    if prev is not None and not self._deferred_rendered:
      self._render(self._spec_host, self._spec_dev, prev['trace'], prev['obs'], None)
    self._deferred_rendered = False
    self.check_ok()
    return prev
