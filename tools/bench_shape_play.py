"""Shape tier, short launches at B = 32 768: play() and T = 1 / 2 / 4 / 8 rollouts (kernel time
against T shows the fixed per-launch cost).  Run under rocprofv3 --kernel-trace for durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import hello_world

B = 32768
game, _, _, _ = hello_world.make_game(batch=B, device='cuda')
game.fused.validate_actions = False
acts = torch.randint(0, 4, (64, B), dtype=torch.int8, device='cuda')
for T in (1, 2, 4, 8):
  bufs = game.fused.rollout_buffers(T)
  for _ in range(5):
    game.rollout(acts[:T], out=bufs)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(100):
    game.rollout(acts[:T], out=bufs)
  e1.record(); torch.cuda.synchronize()
  print('rollout T=%d (no board): %.1f us per launch' % (T, e0.elapsed_time(e1) * 10))
for _ in range(20):
  game.play(acts[0])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(200):
  game.play(acts[t % 64])
e1.record(); torch.cuda.synchronize()
print('play() (board too): %.1f us per call' % (e0.elapsed_time(e1) * 5))
