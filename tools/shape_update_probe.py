#!/usr/bin/env python3
"""Shape tier, frame-major path: kernel times of the update pass and the render pass by rollout length
(torch profiler), to separate per-frame from per-launch cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from campx_amd.games import hello_world

for B in (4096, 32768):
  for T in (4, 20, 100):
    game, _, _, _ = hello_world.make_game(batch=B, device='cuda')
    game.fused.validate_actions = False
    acts = torch.randint(0, 4, (T, B), dtype=torch.int8, device='cuda')
    bufs = game.fused.rollout_buffers(T)
    for _ in range(3):
      game.rollout(acts, out=bufs, reset_first=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
      for _ in range(10):
        game.rollout(acts, out=bufs, reset_first=True)
      torch.cuda.synchronize()
    for e in prof.key_averages():
      if 'campx_impl' in e.key:
        print('B=%6d T=%4d  %-40s %8.1f us' % (B, T, e.key.split('::')[-1].split('(')[0][:40], e.device_time_total / e.count))
