import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from campx_amd.games import boat_race, sokoban, wall_world
for name, build in (('boat_race', boat_race.build), ('sokoban', sokoban.build), ('sokoban_l1', lambda **k: sokoban.build(level=1, **k)), ('sokoban_l2', lambda **k: sokoban.build(level=2, **k)), ('wall_world', wall_world.build)):
  for B in (1024, 65536, 131072):
    game = build(batch=B, device='cuda'); game.its_showtime()
    f = game.fused; f.validate_actions = False
    acts = torch.randint(0, 5, (8, B), dtype=torch.int8, device='cuda')
    for t in range(300): game.play(acts[t & 7])
    torch.cuda.synchronize()
    n = 3000
    t0 = time.perf_counter()
    for t in range(n): game.play(acts[t & 7])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    frame = B * (f.n_layers + 1) * f.rows * f.cols
    print('%-11s K=%d B=%6d: %.2f us per play(), %.1f MB per frame, %.2f TB/s' % (name, f.n_dyn, B, dt * 1e6, frame / 1e6, frame / dt / 1e12))
