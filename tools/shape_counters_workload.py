#!/usr/bin/env python3
"""Workload for the shape tier's counter passes: 12 rollouts (T = 100) of Hello World at
B = 32 768 (shape_rollout_kernel: 10.7 GB per launch) and, as the yardstick that DOES reach the
write ceiling, 12 of the boat race at B = 65 536 (render_kernel: 1.15 GB per launch)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import hello_world, boat_race

T = 100
for name, make, B, n_act in (('hello_world', hello_world.make_game, 32768, 4),
                             ('boat_race', boat_race.make_game, 65536, 5)):
  game, _, _, _ = make(batch=B, device='cuda')
  game.fused.validate_actions = False
  acts = torch.randint(0, n_act, (T, B), dtype=torch.int8, device='cuda')
  bufs = game.fused.rollout_buffers(T)
  for _ in range(12):
    game.rollout(acts, out=bufs, reset_first=True)
  torch.cuda.synchronize()
  del game, bufs
  torch.cuda.empty_cache()
print('done')
