#!/usr/bin/env python3
"""Host-side cost of ReturnGatherer.gather_async (1 rank, RCCL)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29542')
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from campx_amd.distributed import ReturnGatherer
B = 65536
ret = torch.zeros(B, device=dev)
g = ReturnGatherer(B, dev, dist)
for _ in range(10): g.gather_async(ret)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): g.gather_async(ret)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('gather_async host %.1f us/call, drained %.1f us/call' % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
out = torch.empty(B, device=dev)
t0 = time.perf_counter()
for _ in range(200): dist.all_gather_into_tensor(out, ret)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('bare all_gather_into_tensor host %.1f us/call, drained %.1f us/call' % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
dist.destroy_process_group()
