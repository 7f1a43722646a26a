"""Build, run and drop games in a loop (incl. the 212 MB four-mover table): device memory free, torch allocation and host RSS must not drift."""
import os, sys, gc
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from campx_amd.games import sokoban, boat_race, hello_world, maze
def rss():
  with open('/proc/self/status') as f:
    for l in f:
      if l.startswith('VmRSS'): return int(l.split()[1]) // 1024
free0 = None
for i in range(12):
  g = sokoban.build(level=2, batch=4096, device='cuda'); g.its_showtime()
  a = torch.randint(0, 5, (20, 4096), dtype=torch.int8, device='cuda')
  g.rollout(a); g.play(a[0])
  h = hello_world.build(batch=512, device='cuda'); h.its_showtime(); h.play(a[0, :512].clamp(0, 3))
  w = maze.build(16, 16, batch=4096, device='cuda'); w.its_showtime()     # wide tier (tabulation cached)
  w.rollout(a); w.play(a[0])
  del g, h, w, a
  gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
  free, total = torch.cuda.mem_get_info()
  if free0 is None: free0 = free
  print('LEAK iter %2d  device free %.1f MB (delta %+.1f MB)  torch allocated %.1f MB  host RSS %d MB'
        % (i, free / 1e6, (free - free0) / 1e6, torch.cuda.memory_allocated() / 1e6, rss()))
