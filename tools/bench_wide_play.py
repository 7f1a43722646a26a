#!/usr/bin/env python3
"""Engine.play() per frame on the wide tier (16x16 / 32x32 maze; the coin field of
examples/coins_batched.py, rows of 300 bytes and three pieces in a mask): us per call, validation off.

    python tools/bench_wide_play.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from campx_amd.games import maze  # noqa: E402


def main():
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
  import coins_batched
  for rows, B in ((16, 65536), (16, 4096), (32, 16384), (32, 1024), (0, 65536), (0, 16384), (0, 4096), (0, 1000),
                  (-1, 32768), (-1, 4096)):
    if rows == -1:       # ... with its switch: a floor that turns, 13 variants of the scenery, rows of 420 bytes
      game = coins_batched.make_game(batch=B, device='cuda')
    elif rows == 0:
      game = coins_batched.make_game(floor=False, batch=B, device='cuda')
    else:
      game = maze.build(rows, rows, batch=B, device='cuda')
    game.its_showtime()
    game.fused.validate_actions = False
    ids = [torch.randint(0, 5, (B,), dtype=torch.int8, device='cuda') for _ in range(8)]
    from campx_amd import _hip
    times = []
    for step in (1, 0):           # (0: the update + render pair, what every such call was before wide_step_kernel)
      with _hip.config(wide_step=step):
        for i in range(200):
          game.play(ids[i & 7])
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for i in range(n):
          game.play(ids[i & 7])
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / n)
    dt = times[0]
    frame = B * game.fused.n_layers * game.fused.rows * game.fused.cols
    print('{0} B={1}: {2:.2f} us per play() = {3:.3e} env-steps/s, {4:.2f} TB/s of '
          'observations ({5:.1f} MB per frame)'.format(
              'maze {0}x{0}'.format(rows) if rows > 0 else ('coin field 6x10' if rows == 0 else 'coins + floor in 13 variants'), B, dt * 1e6, B / dt,
              frame / dt / 1e12, frame / 1e6) + '; as update + render pair {:.2f} us'.format(times[1] * 1e6))


if __name__ == '__main__':
  main()
