#!/usr/bin/env python3
"""Workload of profiles/r03_wide_rocprofv3.txt: the wide tier beyond the maze - a four-thing
user game (examples/own_game_batched.py) in rollouts, and play() per frame on both."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'examples'))
import own_game_batched as ex  # noqa: E402
from campx_amd.games import maze  # noqa: E402

ex.run(batch=65536, frames=100, launches=20)
for game, B in ((ex.make_game(batch=65536, device='cuda'), 65536),
                (maze.build(16, 16, batch=4096, device='cuda'), 4096)):
  game.its_showtime()
  game.fused.validate_actions = False
  ids = torch.randint(0, 5, (B,), dtype=torch.int8, device='cuda')
  for _ in range(300):
    game.play(ids)
torch.cuda.synchronize()
