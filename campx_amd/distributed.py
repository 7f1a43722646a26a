"""Multi-GPU: shard environments across ranks, gather episode returns for logging.

Environments never interact (the reference's engine holds exactly one board,
campx/engine.py:31-40), so the batch is split contiguously across one process
per GPU and the step path needs NO collective.  The only exchange is for
logging: once per episode every rank contributes its per-environment episode
returns (the quantity the reference logs as `R`, examples/reinforce.py:186-194)
to an all-gather - RCCL over xGMI on GPUs (`backend='nccl'`), gloo in CPU tests.

The gather is latency-bound (256 KiB per rank at B = 65 536) and is issued on a
side stream from a snapshot of the returns, so the next episode's kernel never
waits for it.
"""

import time

import torch


def shard_range(global_batch, rank, world):
  """Contiguous [start, stop) of the environments owned by `rank`.

  The first `global_batch % world` ranks own one environment more.
  """
  if not 0 <= rank < world:
    raise ValueError('rank {} outside world of {}'.format(rank, world))
  base, extra = divmod(global_batch, world)
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


class ReturnGatherer(object):
  """All-gather of per-environment episode returns, off the critical path.

  Requires every rank to hold the same number of environments (`batch`), which
  is how bench.py shards (weak scaling).  `gather_async(ret)` may be called once
  per episode; `wait()` returns the most recent gathered tensor
  `[world * batch]` in rank order.

  On GPUs the collective is issued with `async_op=True`: torch's RCCL process
  group runs it on its own stream, ordered after the work already queued on the
  caller's stream (so it sees the episode's returns) and the caller's stream
  never waits for it.  The returns are first snapshotted (the next episode resets
  them); two snapshot/result slots alternate, and a slot is reused only after the
  collective that read it two episodes earlier has been waited for, which by then
  costs nothing.
  """

  def __init__(self, batch, device, dist, group=None):
    self.dist = dist
    self.group = group
    self.device = torch.device(device)
    self.world = dist.get_world_size(group)
    self.on_gpu = self.device.type == 'cuda'
    self._snap = [torch.zeros(batch, dtype=torch.float32, device=self.device)
                  for _ in range(2)]
    self._out = [torch.zeros(self.world * batch, dtype=torch.float32,
                             device=self.device) for _ in range(2)]
    self._work = [None, None]
    self._turn = 0
    self._last = None

  def gather_async(self, ret, snapshot=True):
    """Start gathering `ret` (float32 [batch]).

    With `snapshot=False` the collective reads `ret` itself: the caller must leave
    it untouched until the next-but-one `gather_async` (bench.py alternates two
    return buffers between episodes), which keeps every extra kernel off the
    rollout's stream.
    """
    i = self._turn
    self._turn ^= 1
    work = self._work[i]
    if work is not None and not (self.on_gpu and work.is_completed()):
      work.wait()                            # two episodes old: normally complete
    src = ret
    if snapshot:
      self._snap[i].copy_(ret, non_blocking=True)
      src = self._snap[i]
    self._work[i] = self.dist.all_gather_into_tensor(
        self._out[i], src, group=self.group, async_op=self.on_gpu)
    self._last = i

  def wait(self):
    if self._last is None:
      return None
    work = self._work[self._last]
    if work is not None:
      work.wait()
    return self._out[self._last]


class ReturnLog(object):
  """Per-rank log of episode returns, all-gathered every `episodes` episodes.

  A collective that runs concurrently with the rollout kernel takes compute units
  away from it for its whole duration (measured on MI355X: +30 us per 240 us
  episode for one 256 KiB all-gather per episode, although nothing waits for it).
  Logging does not need per-episode latency, so each episode's returns go to their
  own row of a local `[episodes, batch]` buffer - the kernel writes them there
  directly, `row()` is the `ret` buffer to hand to the rollout - and the whole
  block is gathered once it is full, double-buffered like `ReturnGatherer`.
  """

  def __init__(self, batch, episodes, device, dist, group=None):
    self.episodes = int(episodes)
    self.dist = dist
    self.group = group
    self.device = torch.device(device)
    self.world = dist.get_world_size(group) if dist is not None else 1
    self.on_gpu = self.device.type == 'cuda'
    self._log = [torch.zeros((self.episodes, batch), dtype=torch.float32,
                             device=self.device) for _ in range(2)]
    self._out = [torch.zeros((self.world, self.episodes, batch),
                             dtype=torch.float32, device=self.device)
                 for _ in range(2)]
    # (the rows as tensors of their own, made once: indexing a tensor costs the host ~3 us,
    # which at small batches - one launch per 14-20 us - is what the device would wait for)
    self._rows = [[block[r] for r in range(self.episodes)] for block in self._log]
    self._work = [None, None]
    self._count = 0
    self._last = None
    # (bench.py: `time_gathers()` makes every gather leave a record - see there)
    self.timing = None

  def time_gathers(self):
    """From now on every gather leaves a record in `self.timing`: an event on the caller's
    stream when the block was complete (`ready`), the host's clock before and after the
    collective call (`issued`, `returned`) and the host's clock when the collective was first
    SEEN complete (`seen_done`: `is_completed()` is asked once per episode while a gather is
    pending, and by `poll()` whenever the caller likes - a query, ~1 us).  (Two earlier forms put
    the END on the device clock with an observer stream that waited for the work object:
    `Work.wait()` cost the host 0.5-2 ms per gather - at small batches, a launch per 20 us, it
    tripled the step time - and what the observer's event then showed was the launch stream's
    progress at the time of the call, not the collective's end.)"""
    self.timing = []

  def poll(self):
    """Stamp the gathers that have completed since the last look with the host's clock."""
    if self.timing:
      self._poll()

  def row(self):
    """The float32 [batch] buffer the next episode accumulates its returns in."""
    block, row = divmod(self._count, self.episodes)
    return self._rows[block & 1][row]

  def episode_done(self):
    """Call after launching an episode; gathers the block when it is complete."""
    self._count += 1
    block, row = divmod(self._count, self.episodes)
    if row != 0:
      if self.timing:
        self._poll()
      return False
    i = (block - 1) & 1                      # the block just completed
    nxt = block & 1                          # about to be overwritten
    work = self._work[nxt]
    if work is not None and not (self.on_gpu and work.is_completed()):
      work.wait()
    timed = self.timing is not None and self.on_gpu
    if timed:
      self._poll()
      ready = torch.cuda.Event(enable_timing=True)
      ready.record()                         # the block's last episode has finished
      issued = time.perf_counter()
    if self.dist is not None:
      self._work[i] = self.dist.all_gather_into_tensor(
          self._out[i].view(-1), self._log[i].view(-1), group=self.group,
          async_op=self.on_gpu)
    else:
      self._out[i][0].copy_(self._log[i])
    if timed:
      self.timing.append(dict(count=self._count, ready=ready, issued=issued,
                              returned=time.perf_counter(), seen_done=None, work=self._work[i]))
    self._last = i
    return True

  def _poll(self):
    for rec in self.timing:
      if rec['seen_done'] is None and (rec['work'] is None or rec['work'].is_completed()):
        rec['seen_done'] = time.perf_counter()

  def align(self):
    """Skip to the start of the next block (rows already logged in the current one are
    dropped): what follows - a timed window - then gathers after exactly `episodes`
    episodes, wherever the warm-up left the counter."""
    block, row = divmod(self._count, self.episodes)
    if row:
      block += 1
      if self._last is not None and (block & 1) == self._last:
        block += 1                           # (not onto the rows of the last gathered block)
      work = self._work[block & 1]
      if work is not None:
        work.wait()
      self._count = block * self.episodes

  def last_local_block(self):
    """This rank's own `[episodes, batch]` rows of the most recently gathered block."""
    return None if self._last is None else self._log[self._last]

  def wait(self):
    """Most recent gathered block `[world, episodes, batch]` (None before the first)."""
    if self._last is None:
      return None
    work = self._work[self._last]
    if work is not None:
      work.wait()
    if self.timing:
      self._poll()
    return self._out[self._last]


def episode_stats(gathered):
  """(mean, min, max) of gathered episode returns, as Python floats."""
  return (float(gathered.mean()), float(gathered.min()), float(gathered.max()))
