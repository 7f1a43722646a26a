"""Rollout API modes, boat race B = 65 536, T = 100: ms per launch and TB/s of what is written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import boat_race

B, T = 65536, 100
game, _, _, _ = boat_race.make_game(batch=B, device='cuda')
game.fused.validate_actions = False
acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')


def timed(fn, n=100, warm=50):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


for label, kw in (('obs int8', dict()),
                  ('obs int8 + board', dict(want_board=True)),
                  ('last frame only (keep_obs=False)', dict(keep_obs=False)),
                  ('obs bf16', dict(obs_dtype=torch.bfloat16))):
  bufs = game.fused.rollout_buffers(T, **kw)
  ms = timed(lambda: game.rollout(acts, out=bufs, reset_first=True))
  nbytes = sum(v.numel() * v.element_size() for v in bufs.values() if torch.is_tensor(v))
  print('%-34s %.4f ms per launch, %7.1f MB written, %.2f TB/s' % (label, ms, nbytes / 1e6,
                                                                 nbytes / 1e9 / ms))
  del bufs
  torch.cuda.empty_cache()
for dtype in (torch.int8, torch.bfloat16):
  game.fused.set_play_obs_dtype(dtype)
  us = timed(lambda: game.play(acts[0]), n=500) * 1e3
  print('play() obs %-10s %.2f us per call' % (str(dtype).replace('torch.', ''), us))
