#!/usr/bin/env python3
"""Engine.play() per frame on the wide tier (16x16 / 32x32 maze): us per call, validation off.

    python tools/bench_wide_play.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from campx_amd.games import maze  # noqa: E402


def main():
  for rows, B in ((16, 65536), (16, 4096), (32, 16384), (32, 1024)):
    game = maze.build(rows, rows, batch=B, device='cuda')
    game.its_showtime()
    game.fused.validate_actions = False
    ids = [torch.randint(0, 5, (B,), dtype=torch.int8, device='cuda') for _ in range(8)]
    for i in range(200):
      game.play(ids[i & 7])
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for i in range(n):
      game.play(ids[i & 7])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    frame = B * game.fused.n_layers * rows * rows
    print('maze {0}x{0} B={1}: {2:.2f} us per play() = {3:.3e} env-steps/s, {4:.2f} TB/s of '
          'observations ({5:.1f} MB per frame)'.format(rows, B, dt * 1e6, B / dt, frame / dt / 1e12,
                                                       frame / 1e6))


if __name__ == '__main__':
  main()
