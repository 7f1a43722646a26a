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
    self._free = [None, None]      # event: collective reading slot i finished
    self._turn = 0
    self._last = None
    self._work = None
    self._stream = torch.cuda.Stream(self.device) if self.on_gpu else None

  def gather_async(self, ret):
    i = self._turn
    self._turn ^= 1
    snap, out = self._snap[i], self._out[i]
    if not self.on_gpu:
      snap.copy_(ret)
      self.dist.all_gather_into_tensor(out, snap, group=self.group)
      self._last = out
      return
    current = torch.cuda.current_stream(self.device)
    if self._free[i] is not None:
      current.wait_event(self._free[i])      # slot reuse, two episodes later
    snap.copy_(ret, non_blocking=True)       # ordered after the episode's kernel
    self._stream.wait_stream(current)
    with torch.cuda.stream(self._stream):
      self._work = self.dist.all_gather_into_tensor(out, snap, group=self.group,
                                                    async_op=True)
      self._work.wait()                      # stream-level wait, not host
      done = torch.cuda.Event()
      done.record(self._stream)
    self._free[i] = done
    self._last = out

  def wait(self):
    if self.on_gpu and self._last is not None:
      torch.cuda.current_stream(self.device).wait_stream(self._stream)
    return self._last


def episode_stats(gathered):
  """(mean, min, max) of gathered episode returns, as Python floats."""
  return (float(gathered.mean()), float(gathered.min()), float(gathered.max()))
