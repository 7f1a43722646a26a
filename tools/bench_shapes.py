#!/usr/bin/env python3
"""Shape tier (Hello World) throughput: rollout and play()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from campx_amd.games import hello_world
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import shape_zoo

from campx_amd import shapes

# hello_world: trails; shape_zoo4: the same art without trails; each on the frame-major path (the
# default since round 5: update pass + row-word render, csrc/k_shape.hip) and on the serial
# one-wave-per-environment kernel (`serial`: shapes.FRAME_MAJOR = False when the game is built)
for which, B in (('hello_world', 4096), ('hello_world', 32768), ('hello_world serial', 4096),
                 ('hello_world serial', 32768), ('shape_zoo4', 4096), ('shape_zoo4', 32768),
                 ('shape_zoo4 serial', 4096), ('shape_zoo4 serial', 32768)):
  shapes.FRAME_MAJOR = not which.endswith('serial')
  T = 100
  if which.startswith('hello_world'):
    game, _, _, _ = hello_world.make_game(batch=B, device='cuda')
  else:
    game = shape_zoo.library_builders()['shape_zoo4'](batch=B, device='cuda')
    game.its_showtime()
  game.fused.validate_actions = False
  acts = torch.randint(0, 4, (T, B), dtype=torch.int8, device='cuda')
  bufs = game.fused.rollout_buffers(T)
  for _ in range(3):
    game.rollout(acts, out=bufs, reset_first=True)
  torch.cuda.synchronize()
  n = 20
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
  ev[0].record()
  for i in range(n):
    game.rollout(acts, out=bufs, reset_first=True)
    ev[i + 1].record()
  torch.cuda.synchronize()
  per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
  ms = ev[0].elapsed_time(ev[n]) / n
  print('  per launch: min %.3f median %.3f max %.3f ms' % (per[0], per[n // 2], per[-1]))
  L, H, W = game.fused.n_layers, game.fused.rows, game.fused.cols
  gb = B * T * (L * H * W + 9) / 1e9
  print('%s B=%d T=%d: %.3f ms per launch, %.2e env-steps/s, %.2f TB/s of observations'
        % (which, B, T, ms, B * T / ms * 1e3, gb / ms))
  t0 = time.perf_counter()
  for t in range(200):
    game.play(acts[t % T])
  torch.cuda.synchronize()
  print('  play(): %.1f us per call' % ((time.perf_counter() - t0) / 200 * 1e6))
