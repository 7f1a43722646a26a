#!/usr/bin/env python3
"""Workload for tracing the device-side state enumeration: its_showtime() of the 16x16 sokoban
with two boxes (campx_amd/enumerate_states.py: 100 breadth-first levels, 4.4 M states), twice
(the second one has every code object paged in), and one 100-frame rollout."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import sokoban

for attempt in range(2):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  game = sokoban.build(batch=65536, device='cuda', level=3)
  game.its_showtime()
  torch.cuda.synchronize()
  tr = game.fused.traced
  print('its_showtime %d: %.2f s, %d states, %d levels' % (attempt, time.perf_counter() - t0, tr.n_states, tr.n_levels))
actions = torch.randint(0, 5, (100, 65536), dtype=torch.int8, device='cuda')
game.rollout(actions)
torch.cuda.synchronize()
