#!/usr/bin/env python3
"""Rollout rate of game k of tests/random_pickups.py (drapes of several cells, changing backdrops):
    python tools/bench_pickups.py [k] [batch] [road]  (k = 3: seven coins; road: own | variants | things -
    how the pieces reach the kernels, tests/test_random_pickups.py::_road)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch  # noqa: E402

import random_pickups  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
road = sys.argv[3] if len(sys.argv) > 3 else 'own'
from campx_amd import gamespec  # noqa: E402
if road == 'mask':          # (also where three tracked things would stay on the one-cell tier)
  gamespec.PIECES_AS_THINGS_MAX = 0
if road in ('variants', 'things'):
  gamespec.WIDE_MAX_PIECES = 0
if road == 'things':
  gamespec.WIDE_MAX_VARIANTS = 1
d = random_pickups.definitions()[k]
game = random_pickups.builder(d)(batch=B, device='cuda')
game.its_showtime()
f = game.fused
f.validate_actions = False
T, n = 100, 30
acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
out = f.rollout_buffers(T)
for _ in range(10):
  f.rollout(acts, out=out, reset_first=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
  f.rollout(acts, out=out, reset_first=True)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / n
row = f.n_layers * f.rows * f.cols
spec = getattr(f, 'spec', None)
print('pickup %d (%s, %s, %d things, %d pieces in a mask, %d variants) B=%d  %.4f ms per 100-frame rollout  %.2f TB/s of observations' % (
    k, d['kind'], type(f).__name__, f.n_dyn, getattr(spec, 'n_pieces', 0), max(1, getattr(spec, 'n_variants', 1)), B, ms,
    row * B * T / (ms / 1e3) / 1e12))
