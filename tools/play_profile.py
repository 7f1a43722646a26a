#!/usr/bin/env python3
"""Workload for profiling Engine.play(): 300 frames of the boat race at B = 65 536 (the
one-frame kernel), then, as yardsticks for a one-shot launch of that size, torch `fill_`
launches of the same 13.4 MB (observation + flat board: what a frame writes) and of 64 bytes
(the launch floor).   CAMPX_NO_ROWS_STEP=1 profiles the 64-environments-per-wave kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
game, obs, r, d = boat_race.make_game(batch=B, device='cuda')
game.fused.validate_actions = False
acts = torch.randint(0, 5, (300, B), dtype=torch.int8, device='cuda')
for t in range(300):
  game.play(acts[t])
torch.cuda.synchronize()
frame = torch.empty(B * (175 + 25), dtype=torch.int8, device='cuda')
tiny = torch.empty(64, dtype=torch.int8, device='cuda')
for _ in range(300):
  frame.fill_(1)
torch.cuda.synchronize()
for _ in range(300):
  tiny.fill_(1)
torch.cuda.synchronize()
print('played 300 frames at B =', B)
