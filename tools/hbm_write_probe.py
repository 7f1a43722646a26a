#!/usr/bin/env python3
"""Measure the box's achievable HBM streaming-WRITE bandwidth with stock torch kernels.

Context for bench.py's roofline fraction: the rollout kernel is a pure write
stream (1.2 GB of observations per launch).  This prints GB/s of fill_/zero_/copy_
over the same footprint so the kernel can be compared with what the memory system
sustains, next to the 8 TB/s spec peak.
"""
import torch

def timed(fn, n=20, warm=3):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(n): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / n / 1e3

nbytes = 100 * 65536 * 175
for dtype, name in ((torch.int8, 'int8'), (torch.float32, 'f32')):
  x = torch.empty(nbytes // torch.empty((), dtype=dtype).element_size(), dtype=dtype, device='cuda')
  y = torch.empty_like(x)
  t = timed(lambda: x.fill_(1))
  print('fill_  %-5s %.3f ms  %.0f GB/s (write only)' % (name, t * 1e3, nbytes / t / 1e9))
  t = timed(lambda: x.zero_())
  print('zero_  %-5s %.3f ms  %.0f GB/s (write only)' % (name, t * 1e3, nbytes / t / 1e9))
  t = timed(lambda: y.copy_(x))
  print('copy_  %-5s %.3f ms  %.0f GB/s (read+write %.0f)' % (name, t * 1e3, nbytes / t / 1e9, 2 * nbytes / t / 1e9))
