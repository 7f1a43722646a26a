import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import torch
import traced_games
from campx_amd import tabulate
def run(name, build, B=65536, T=100):
  game = build(batch=B, device='cuda'); game.its_showtime()
  f = game.fused; f.validate_actions = False
  acts = torch.randint(0, 5, (T, B), dtype=torch.int8, device='cuda')
  bufs = f.rollout_buffers(T)
  for _ in range(20): f.rollout(acts, out=bufs, reset_first=True)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(50): f.rollout(acts, out=bufs, reset_first=True)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
  ids = acts[0].contiguous()
  for _ in range(200): f.play(ids)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(2000): f.play(ids)
  torch.cuda.synchronize(); dp = (time.perf_counter() - t0) / 2000
  row = f.n_layers * f.rows * f.cols
  print('%-10s %-9s K=%d states=%d: rollout %.3f ms (%.2f TB/s of observations), play() %.2f us' % (name, type(f).__name__, f.n_dyn, f.traced.n_states, dt * 1e3, B * T * row / dt / 1e12, dp * 1e6))
for name in ('vault', 'trio', 'mirror', 'burrow', 'ice_rink'):
  build = traced_games.GAMES[name]
  tabulate._CACHE.clear(); tabulate.DENSE_MAX_ENTRIES = 8 << 20
  run(name, build)
  tabulate._CACHE.clear(); tabulate.DENSE_MAX_ENTRIES = 0
  run(name, build)
