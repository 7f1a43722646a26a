"""`Engine.play()` frames captured once in a HIP graph and replayed: the closed RL loop without
the host in it.

The reference's driver (examples/reinforce.py:136-149) is a Python loop per frame: policy
forward, sample, `game.play(action)`.  Batched, every one of those steps is a handful of kernels
of a few microseconds each, and what the loop costs is the HOST - an op dispatch per kernel:
`play()` alone reads 6.5 us per call at B = 65 536 where its kernel takes 4.5.  All of them only
ENQUEUE work on torch's current stream (the campx:: ops take no lock, allocate nothing, never
synchronise), so n frames of the loop - the policy network, the sampling and `play()` - can be
captured in one HIP graph:

    graph = game.capture_play(32, policy=lambda obs, t: sample(net(obs.layered_board)))
    graph.replay()            # 32 frames: policy -> ids -> campx::step, one launch of the host's
    graph.reward, graph.done  # [32, B], what play() returned frame by frame

and without a policy, for action ids that are already on the device (open loop, or chosen by a
learner outside the graph):

    graph = game.capture_play(32)
    graph.replay(actions)     # int [32, B]

The game's state lives in the engine's own buffers and carries over from replay to replay, and
to and from ordinary `play()` / `rollout()` calls in between.  Bit-exact against frame-by-frame
`play()` (tests/test_play_graph.py).  For every batched tier (`FusedGame`, `WideGame`,
`ShapeGame`): the capture simply calls the tier's own `play()` with the per-frame outputs
pointed at row t of the graph's buffers.
"""

import torch


class PlayGraph(object):
  """n captured frames of `play()`.  Attributes after `replay()`:

    actions   int8 [n, B]    what was played (the policy's choices, or the ids handed in)
    reward    f32  [n, B]    (None if the game never rewards)      campx/plot.py:186-211
    discount  f32  [n, B]
    done      u8   [n, B]    the episode ended on that frame (the environment is rebuilt from
                             the art before its next frame, as everywhere in the batched tiers)
    perf      i8   [n, B]    hidden performance (None unless the game declared one)
    obs       [n, B, L, H, W] in `play()`'s observation dtype - the frame the policy SAW when it
                             chose actions[t] (only with `record_obs=True`); `observation` is
                             the engine's own buffer: the frame after the last action
  """

  def __init__(self, fused, n_frames, policy=None, record_obs=False, warmup=2):
    n = int(n_frames)
    if n < 1:
      raise ValueError('capture_play() needs at least one frame')
    if fused.validate_actions == 'sync':
      raise ValueError('validate_actions="sync" reads a device counter after every frame, which a '
                       'graph cannot hold: use True (the flag is looked at after each replay) or False')
    if fused.frame < 0:
      raise RuntimeError('capture_play() needs an engine that has been through its_showtime()')
    self.fused, self.n, self.policy = fused, n, policy
    f, B, dev = fused, fused.batch, fused.device
    self.actions = torch.zeros((n, B), dtype=torch.int8, device=dev)
    self.reward = torch.zeros((n, B), dtype=torch.float32, device=dev) if f.any_reward else None
    self.discount = torch.zeros((n, B), dtype=torch.float32, device=dev)
    self.done = torch.zeros((n, B), dtype=torch.uint8, device=dev)
    has_perf = getattr(f, '_perf_arg', None) is not None
    self.perf = torch.zeros((n, B), dtype=torch.int8, device=dev) if has_perf else None
    self.obs = (torch.zeros((n,) + tuple(f._obs.shape), dtype=f._obs.dtype, device=dev)
                if record_obs else None)
    self._graph = torch.cuda.CUDAGraph()
    # what the frames change: saved round the warm-up runs, which really play (capture does not)
    state = [t for t in (getattr(f, 'pos', None), getattr(f, 'state', None), f.done, f.ret,
                         getattr(f, 'backdrop', None), f._obs, f._board) if torch.is_tensor(t)]
    saved = [t.clone() for t in state]
    frame0 = f.frame
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
      for _ in range(max(1, warmup)):          # (lazy allocations and library set-up happen here)
        self._frames()
        for t, kept in zip(state, saved):
          t.copy_(kept)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    with torch.cuda.graph(self._graph):
      self._frames()
    f.frame = frame0
    # (the graph holds ADDRESSES: the engine's frame buffers as they were when it was captured)
    self._captured = (f._obs.data_ptr(), f._board.data_ptr())

  def _frames(self):
    """The n frames, issued on the current stream: called to warm up and, once, under capture."""
    f = self.fused
    single = (f._reward, f._discount, f._step_done, getattr(f, '_perf_arg', None))
    try:
      with torch.no_grad():                    # (acting only: a learner recomputes what it needs)
        self._frames_no_grad()
    finally:
      f._reward, f._discount, f._step_done = single[:3]
      if self.perf is not None:
        f._perf_arg = single[3]

  def _frames_no_grad(self):
    f = self.fused
    for t in range(self.n):
      if self.obs is not None:
        self.obs[t].copy_(f._obs)
      if self.policy is not None:
        ids = self.policy(f._observation_cache, t)
        if tuple(ids.shape) != (f.batch,):
          raise ValueError('the policy must return action ids of shape [B] = [{}], got {}'.format(
              f.batch, tuple(ids.shape)))
        self.actions[t].copy_(ids)             # (any integer dtype: narrowed here, on the device)
      # the tier's own play(), its per-frame outputs pointed at row t
      f._reward = self.reward[t] if self.reward is not None else f._reward
      f._discount, f._step_done = self.discount[t], self.done[t]
      if self.perf is not None:
        f._perf_arg = self.perf[t]
      f.play(self.actions[t])

  @property
  def observation(self):
    """The engine's own `Observation` (campx/rendering.py:181-219): after `replay()` the frame
    that follows the last action."""
    return self.fused._observation_cache

  def replay(self, actions=None):
    """Run the n frames.  `actions`: integer ids [n, B] (a graph captured without a policy plays
    these; with a policy they are ignored and `self.actions` holds what the policy chose).
    Nothing is synchronised; returns self."""
    f = self.fused
    if (f._obs.data_ptr(), f._board.data_ptr()) != self._captured:
      raise RuntimeError('the engine\'s frame buffers are not the ones this graph was captured with '
                         '(set_play_obs_dtype() after capture_play()?): capture again')
    if actions is not None:
      if self.policy is not None:
        raise ValueError('this graph was captured with a policy: it chooses the actions')
      actions = torch.as_tensor(actions)
      if tuple(actions.shape) != (self.n, f.batch):
        raise ValueError('actions must have shape {}, got {}'.format((self.n, f.batch), tuple(actions.shape)))
      if actions.dtype != torch.int8:          # narrow without wrapping (256 is not "left")
        actions = actions.to(f.device).clamp(-1, 5)
      self.actions.copy_(actions, non_blocking=True)
    self._graph.replay()
    f.frame += self.n
    if getattr(f, '_aux_in_sync', None) is not None:
      f._aux_in_sync = False
    if f.validate_actions:
      f._after_launch()
    return self
