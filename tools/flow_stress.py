#!/usr/bin/env python3
"""Stress of the one-launch rollout: N launches compared, every one, with the two-kernel path of a
twin engine (campx::update and campx::render as separate ops on the same stream).
    python tools/flow_stress.py [batch] [launches] [frames] [1: with another stream busy] [game]
game: boat_race (default), sokoban, sokoban_l1, sokoban_l2 (two to four movers: pipe_multi_kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from campx_amd.games import boat_race, sokoban  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
GAME = sys.argv[5] if len(sys.argv) > 5 else 'boat_race'
build = {'boat_race': boat_race.build, 'sokoban': sokoban.build,
         'sokoban_l1': lambda **kw: sokoban.build(level=1, **kw),
         'sokoban_l2': lambda **kw: sokoban.build(level=2, **kw)}[GAME]
a, b = (build(batch=B, device='cuda') for _ in range(2))
for g in (a, b):
  g.its_showtime()
  g.fused.validate_actions = False
gen = torch.Generator(device='cuda').manual_seed(B)
pool = torch.randint(0, 5, (64, T, B), generator=gen, dtype=torch.int8, device='cuda')
out_a = a.fused.rollout_buffers(T)
ob = b.fused.rollout_buffers(T)
bad = torch.zeros((), dtype=torch.int64, device='cuda')
fb = b.fused
# optional noise: another stream streaming and chasing latency chains at the same time (argv[4] = 1)
noisy = len(sys.argv) > 4 and sys.argv[4] == '1'
if noisy:
  side = torch.cuda.Stream()
  junk = torch.empty(64 << 20, dtype=torch.int8, device='cuda')
  c = boat_race.build(batch=2048, device='cuda')
  c.its_showtime()
  c.fused.validate_actions = False
  c_out = c.fused.rollout_buffers(T)
  c_acts = torch.randint(0, 5, (T, 2048), dtype=torch.int8, device='cuda')
t0 = time.time()
for i in range(N):
  if noisy:
    with torch.cuda.stream(side):
      junk.fill_(i & 1)
      c.rollout(c_acts, out=c_out)          # (a third game: its own one-launch rollouts)
  acts = pool[i & 63]
  fresh = (i % 7 == 0)
  a.rollout(acts, out=out_a, reset_first=fresh)
  # the twin: the two kernels as ops of their own, on the same stream
  fb._update(fb._spec_host, fb._spec_dev, fb.pos, fb.done, fb.ret, fb._pair_table, acts, ob['reward'],
             ob['discount'], ob['done'], ob['perf'], ob['trace'], None, None, fresh)
  fb._render(fb._spec_host, fb._spec_dev, ob['trace'], ob['obs'], None)
  bad += (out_a['obs'] != ob['obs']).any().to(torch.int64)
  bad += (out_a['reward'] != ob['reward']).any().to(torch.int64)
  bad += (out_a['trace'] != ob['trace']).any().to(torch.int64)
  if i % 5000 == 4999:
    print('B=%d: %d launches, %d mismatching, %.1f s' % (B, i + 1, int(bad), time.time() - t0), flush=True)
torch.cuda.synchronize()
assert int(bad) == 0, int(bad)
assert torch.equal(a.fused.pos, b.fused.pos) and torch.equal(a.fused.ret, b.fused.ret)
print('ok B=%d launches=%d' % (B, N))   # (tests look for this line)
