"""its_showtime() cost per game, including the state-table builds (run on the GPU box)."""
import time, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from campx_amd.games import sokoban, boat_race
torch.zeros(1, device='cuda')
for name, mk in (('boat_race', lambda: boat_race.build(batch=1024, device='cuda')),
                 ('sokoban', lambda: sokoban.build(batch=1024, device='cuda')),
                 ('sokoban_l1', lambda: sokoban.build(level=1, batch=1024, device='cuda')),
                 ('sokoban_l2', lambda: sokoban.build(level=2, batch=1024, device='cuda'))):
  for rep in range(2):
    t0 = time.perf_counter(); g = mk(); g.its_showtime(); torch.cuda.synchronize()
    print('SHOWTIME %s %.3f s' % (name, time.perf_counter() - t0))
