"""The N > 1 path on CPU: shard environments across 2 gloo ranks, gather returns.

On GPUs the per-rank work is the HIP rollout and the collective is RCCL
(bench.py --gpus N); here the oracle stands in for the rollout so that the
sharding arithmetic and the gather (campx_amd.distributed) are exercised with the
same data flow: rank r owns a contiguous slice, no step-path communication, one
all-gather of per-environment episode returns per episode.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from campx_amd import gamespec
from campx_amd.distributed import shard_range, ReturnGatherer, ReturnLog, episode_stats
from campx_amd.games import boat_race


def test_shard_range_partitions_exactly():
  for total, world in [(65536 * 8, 8), (10, 3), (7, 8), (1, 1), (100, 7)]:
    cuts = [shard_range(total, r, world) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    sizes = [b - a for a, b in cuts]
    assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    shard_range(8, 2, 2)


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, global_batch, frames, out_dir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from oracle import cpu as oracle_cpu
    oracle_cpu.set_threads(1)
    start, stop = shard_range(global_batch, rank, world)
    actions = np.load(os.path.join(out_dir, 'actions.npy'))[:, start:stop]
    og = oracle_cpu.OracleGame.from_description(gamespec.describe(boat_race.build()))
    gatherer = ReturnGatherer(stop - start, 'cpu', dist)
    for episode in range(2):            # two episodes: exercises the double buffering
      out = og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
      returns = torch.from_numpy(out['reward'].sum(0))
      gatherer.gather_async(returns)
    gathered = gatherer.wait()
    assert gathered.shape == (global_batch,)
    # the episode-return log: 5 episodes, gathered in blocks of 2
    log = ReturnLog(stop - start, 2, 'cpu', dist)
    for episode in range(5):
      row = log.row()
      row.copy_(returns + episode)            # stands in for the kernel's accumulation
      full = log.episode_done()
      assert full == (episode % 2 == 1)
    block = log.wait()                        # episodes 2 and 3
    assert block.shape == (world, 2, stop - start)
    assert torch.equal(block[rank, 0], returns + 2) and torch.equal(block[rank, 1], returns + 3)
    if rank == 0:
      np.save(os.path.join(out_dir, 'gathered.npy'), gathered.numpy())
      mean, lo, hi = episode_stats(gathered)
      assert lo <= mean <= hi
    dist.barrier()
  finally:
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
  world, global_batch, frames = 2, 64, 30
  actions = np.random.RandomState(3).randint(0, 5, size=(frames, global_batch)).astype(np.int8)
  np.save(tmp_path / 'actions.npy', actions)
  mp.spawn(_worker, args=(world, _free_port(), global_batch, frames, str(tmp_path)),
           nprocs=world, join=True)
  from oracle import cpu as oracle_cpu
  og = oracle_cpu.OracleGame.from_description(gamespec.describe(boat_race.build()))
  whole = og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
  gathered = np.load(tmp_path / 'gathered.npy')
  assert np.array_equal(gathered, whole['reward'].sum(0))
