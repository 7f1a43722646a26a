#!/usr/bin/env python3
"""Host cost of one rollout() call (where the device waits for the host): in order, and pipelined
over two streams (update pass on the side stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race, sokoban
for name, build, B in (('boat_race', boat_race.build, 4096), ('sokoban', sokoban.build, 4096),
                       ('sokoban', sokoban.build, 16384), ('sokoban', sokoban.build, 32768)):
  for mode in ('in order', 'pipelined'):
    game = build(batch=B, device='cuda'); game.its_showtime()
    f = game.fused; f.validate_actions = False
    T = 100
    acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
    first = f.rollout_buffers(T)
    bufs = [first, f.rollout_buffers(T, share=first)]
    for i in range(20): f.rollout(acts, out=bufs[i & 1], reset_first=True, pipelined=(mode == 'pipelined'))
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): f.rollout(acts, out=bufs[i & 1], reset_first=True, pipelined=(mode == 'pipelined'))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-10s B=%5d %-10s host %.1f us per call, with device %.1f us per call' % (name, B, mode, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
