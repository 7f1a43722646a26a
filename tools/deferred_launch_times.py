#!/usr/bin/env python3
"""Per-launch durations of deferred rollouts (HIP events around every launch), by buffer set:
    python tools/deferred_launch_times.py [batch] [launches]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from campx_amd.games import boat_race  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
T = 100
game = boat_race.build(batch=B, device='cuda')
game.its_showtime()
fused = game.fused
fused.validate_actions = False
acts = [torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda') for _ in range(2)]
first = fused.rollout_buffers(T)
sets = [first, fused.rollout_buffers(T, share=first)]
for i in range(300):
  fused.rollout_deferred(acts[i & 1], sets[i & 1], reset_first=True)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
for i in range(N):
  ev[i][0].record()
  fused.rollout_deferred(acts[i & 1], sets[i & 1], reset_first=True)
  ev[i][1].record()
torch.cuda.synchronize()
us = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
print('B=%d: all  median %.1f  p10 %.1f  p90 %.1f us' % (B, np.median(us), np.percentile(us, 10), np.percentile(us, 90)))
for k in (0, 1):
  part = us[k::2]
  print('  set %d median %.1f  p10 %.1f  p90 %.1f' % (k, np.median(part), np.percentile(part, 10), np.percentile(part, 90)))
print('  first 40:', ' '.join('%.0f' % x for x in us[:40]))
# the same actions for both sets: is it the data?
for i in range(N):
  ev[i][0].record()
  fused.rollout_deferred(acts[0], sets[i & 1], reset_first=True)
  ev[i][1].record()
torch.cuda.synchronize()
us = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
print('  same actions every launch: median %.1f  p10 %.1f  p90 %.1f;  first 20: %s' % (
    np.median(us), np.percentile(us, 10), np.percentile(us, 90), ' '.join('%.0f' % x for x in us[:20])))
fused.flush()
