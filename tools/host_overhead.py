#!/usr/bin/env python3
"""How long does the host spend in one FusedGame.rollout() call (no sync)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race
B, T = 65536, 100
game = boat_race.build(batch=B, device='cuda'); game.its_showtime()
f = game.fused; f.validate_actions = False
acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
obs = torch.empty((T, B, f.n_layers, 5, 5), dtype=torch.int8, device='cuda')
for _ in range(5): f.rollout(acts, obs=obs, reset_first=True)
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter(); host = []
for _ in range(N):
  h0 = time.perf_counter(); f.rollout(acts, obs=obs, reset_first=True); host.append(time.perf_counter() - h0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
host.sort()
print('split=%s: host per call median %.1f us (min %.1f max %.1f); enqueue loop %.1f us/call; with drain %.1f us/call' % (
    os.environ.get('CAMPX_SPLIT', '0'), host[N // 2] * 1e6, host[0] * 1e6, host[-1] * 1e6, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
