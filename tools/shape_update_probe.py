#!/usr/bin/env python3
"""Shape tier, frame-major path: kernel times of the update pass and the render pass by rollout length
(torch profiler), to separate per-frame from per-launch cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from campx_amd.games import hello_world
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))

for B in [int(x) for x in (sys.argv[1:] or ['4096', '32768'])]:
  for T in [int(x) for x in os.environ.get('PROBE_T', '20 100').split()]:
    if os.environ.get('PROBE_GAME'):
      import shape_zoo
      game = shape_zoo.library_builders()[os.environ['PROBE_GAME']](batch=B, device='cuda')
      game.its_showtime()
    else:
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
