#!/usr/bin/env python3
"""Host cost of one rollout() call at a small batch (where the device waits for the host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race, sokoban
for name, build in (('boat_race', boat_race.build), ('sokoban', sokoban.build)):
  for B in (1000, 4096):
    game = build(batch=B, device='cuda'); game.its_showtime()
    f = game.fused; f.validate_actions = False
    T = 100
    acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
    out = f.rollout_buffers(T)
    for _ in range(20): f.rollout(acts, out=out, reset_first=True)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): f.rollout(acts, out=out, reset_first=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-10s B=%5d  host %.1f us per call, with device %.1f us per call' % (name, B, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    # the pieces
    t0 = time.perf_counter()
    for _ in range(n): f._scratch(T, out)
    print('   _scratch %.2f us' % ((time.perf_counter() - t0) / n * 1e6))
    t0 = time.perf_counter()
    for _ in range(n): f.check_ok()
    print('   check_ok %.2f us' % ((time.perf_counter() - t0) / n * 1e6))
