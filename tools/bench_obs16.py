"""Rollouts with 16-bit observations (the consumer hand-off): ms per launch and
TB/s of the observation stream for int8 / float16 / bfloat16, boat race B = 65 536, T = 100."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race

B, T = 65536, 100
game, _, _, _ = boat_race.make_game(batch=B, device='cuda')
game.fused.validate_actions = False
acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
for dtype in (torch.int8, torch.float16, torch.bfloat16):
  bufs = game.fused.rollout_buffers(T, obs_dtype=dtype)
  for _ in range(50):
    game.rollout(acts, out=bufs, reset_first=True)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  n = 100
  for _ in range(n):
    game.rollout(acts, out=bufs, reset_first=True)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / n
  gb = bufs['obs'].numel() * bufs['obs'].element_size() / 1e9
  print('obs %-14s %.4f ms per launch, %.2f TB/s of observations, %.3e env-steps/s'
        % (str(dtype), ms, gb / ms, B * T / ms * 1e3))
  del bufs
  torch.cuda.empty_cache()
