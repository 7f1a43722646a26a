import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from campx_amd.games import boat_race
B = 64
game = boat_race.build(batch=B, device='cuda'); game.its_showtime()
f = game.fused; f.validate_actions = False
ids = torch.randint(0, 5, (B,), dtype=torch.int8, device='cuda')
args = (f._spec_host, f._spec_dev, f.pos, f.done, f.ret, f._pair_table, ids, f._obs, f._board, f._reward, f._discount, f._step_done, f._perf_arg, None, None)
def bench(label, fn, n=20000):
  for _ in range(500): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
  print('%-44s host %.2f us per call (drained after %.2f us per call)' % (label, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
bench('game.play(ids)', lambda: game.play(ids))
bench('f.play(ids)', lambda: f.play(ids))
bench('torch.ops.campx.step.default(*args)', lambda: f._step(*args))
with torch.no_grad():
  bench('... under no_grad', lambda: f._step(*args))
with torch.inference_mode():
  bench('... under inference_mode', lambda: f._step(*args))
x = torch.zeros(64, device='cuda')
bench('x.add_(1) (a torch in-place op, for scale)', lambda: x.add_(1))
bench('torch.ops.campx.check_actions (2 args)', lambda: torch.ops.campx.check_actions.default(ids, f._bad))
