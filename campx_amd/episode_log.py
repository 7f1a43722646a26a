"""Episode log in the reference's CSV format.

The reference's driver writes one row per episode,
`id,step,t(s),ep,L,R,R_av_5,P,P_av` (examples/reinforce.py:270-284 header,
186-190 row): run id, total environment steps so far, wall time of the episode,
episode index, policy loss, episode return, mean return of the last five episodes,
hidden performance of the episode and its mean.  There an episode is ONE
environment; here it is a batch, so `R` and `P` are means over every environment
that played the episode (all ranks' shards after the all-gather of
`distributed.ReturnLog`).  `step` keeps the reference's meaning - frames one
environment has played so far - and `L` is empty unless a learner supplies a loss.

Deviation, deliberate: the reference re-creates its list of performances every
episode (reinforce.py:124), so its `P_av` always equals `P`; here `P_av` is the
running mean over the episodes logged so far, which is what the column says.

Off the step path by construction: it consumes host numbers (or tensors it reduces
with `.mean()` once per episode block) after `ReturnLog.wait()`; rank 0 only.
"""

import csv
import time

FIELDNAMES = ['id', 'step', 't(s)', 'ep', 'L', 'R', 'R_av_5', 'P', 'P_av']


def _mean(x):
  if x is None:
    return None
  if hasattr(x, 'float') and hasattr(x, 'mean'):      # torch tensor of any dtype
    return float(x.float().mean())
  try:
    import numpy as np
    return float(np.mean(x))
  except Exception:
    return float(x)


class EpisodeCsvLog(object):
  """Writes the reference's episode rows; one `episode()` call per episode."""

  def __init__(self, file, run_id=0, frames_per_episode=100, write_header=True,
               clock=time.time):
    self._own = isinstance(file, str)
    self._file = open(file, mode='w', newline='') if self._own else file
    self._writer = csv.writer(self._file, delimiter=',', quotechar='"',
                              quoting=csv.QUOTE_MINIMAL)
    if write_header:
      self._writer.writerow(FIELDNAMES)
    self.run_id = run_id
    self.frames_per_episode = int(frames_per_episode)
    self._clock = clock
    self._last = clock()
    self.episodes = 0
    self.total_steps = 0
    self._returns = []
    self._performances = []

  def episode(self, returns, performance=None, loss=None, frames=None, seconds=None):
    """Log one episode.

    returns:     per-environment episode returns (tensor / array, any shape) or a number
    performance: per-environment hidden performance of the episode (sum of the frames'
                 -1/0/+1, `rollout()['perf'].sum(0)`), or a number, or None
    loss:        the learner's loss for column L, or None (left empty)
    """
    now = self._clock()
    seconds = round(now - self._last, 2) if seconds is None else seconds
    self._last = now
    self.total_steps += self.frames_per_episode if frames is None else int(frames)
    r, p = _mean(returns), _mean(performance)
    self._returns.append(r)
    if p is not None:
      self._performances.append(p)
    last5 = self._returns[-5:]
    row = [self.run_id, self.total_steps, seconds, self.episodes,
           '' if loss is None else round(float(loss), 2), r, sum(last5) / len(last5),
           '' if p is None else p,
           '' if not self._performances
           else sum(self._performances) / len(self._performances)]
    self._writer.writerow(row)
    self.episodes += 1
    return row

  def block(self, gathered, performance=None, seconds=None):
    """Log a gathered `ReturnLog` block `[world, episodes, batch]`, one row per episode.

    `performance`, if given, is `[world, episodes, batch]` (or `[episodes, batch]`)
    per-environment episode performance laid out the same way.
    """
    rows = []
    n = gathered.shape[1]
    for e in range(n):
      perf = None
      if performance is not None:
        perf = performance[:, e] if len(performance.shape) == 3 else performance[e]
      rows.append(self.episode(gathered[:, e], perf,
                               seconds=None if seconds is None else seconds / n))
    return rows

  def close(self):
    self._file.flush()
    if self._own:
      self._file.close()
