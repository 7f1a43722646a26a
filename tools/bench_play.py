#!/usr/bin/env python3
"""Engine.play()-per-frame mode: one launch per frame, state round-trips HBM."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race

for B in (1, 1024, 65536):
  game, obs, r, d = boat_race.make_game(batch=B, device='cuda')
  acts = torch.randint(0, 5, (200, B), dtype=torch.int8, device='cuda')
  for validate in (True, False):
    game.fused.validate_actions = validate
    for t in range(20):
      game.play(acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(200):
      game.play(acts[t])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print('play() B=%6d validate=%d: %.1f us/frame  %.3e env-steps/s' % (B, validate, dt * 1e6, B / dt))
  onehot = torch.nn.functional.one_hot(acts.long(), 5).float()
  game.fused.validate_actions = True
  for t in range(20):
    game.play(onehot[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(200):
    game.play(onehot[t])
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 200
  print('play(one-hot float [B, 5]) B=%6d lazy validation: %.1f us/frame' % (B, dt * 1e6))
