#!/usr/bin/env python3
"""Workload for tracing Engine.play() per game: 300 frames of every library game at the batch
given (default 65 536), each followed by `fill_` launches of the bytes one of its frames writes
(observation + flat board) - the yardstick for a one-shot launch of that size.  Kernel names
in the trace tell the games apart by their grid sizes; the fills are printed with their sizes.
    python tools/play_trace_games.py [B] [game ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race, sokoban, wall_world

GAMES = {'boat_race': boat_race.build, 'sokoban': sokoban.build,
         'sokoban_l1': lambda **k: sokoban.build(level=1, **k),
         'sokoban_l2': lambda **k: sokoban.build(level=2, **k), 'wall_world': wall_world.build}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
names = sys.argv[2:] or list(GAMES)
for name in names:
  game = GAMES[name](batch=B, device='cuda')
  game.its_showtime()
  f = game.fused
  f.validate_actions = False
  acts = torch.randint(0, 5, (8, B), dtype=torch.int8, device='cuda')
  for t in range(100):
    game.play(acts[t & 7])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(300):
    game.play(acts[t & 7])
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 300
  nbytes = B * (f.n_layers + 1) * f.rows * f.cols
  frame = torch.empty(nbytes, dtype=torch.int8, device='cuda')
  for _ in range(100):
    frame.fill_(1)
  torch.cuda.synchronize()
  print('%-11s K=%d B=%d: %.2f us per play(), frame %.1f MB (fill_ grid follows it in the trace)'
        % (name, f.n_dyn, B, dt * 1e6, nbytes / 1e6))
  del game, f, frame
  torch.cuda.empty_cache()
