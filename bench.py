#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec, boat race 5x5, 65 536 environments per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" of this bench = one pass of the hot path over one batch of synthetic
input = ONE rollout launch: an episode of `--frames` (default 100, the
reference's episode length, examples/reinforce.py:36) consecutive Engine.play()
frames for every environment of the rank's shard, rebuilt from the art at the
start (make_game() per episode, reinforce.py:122), on a committed-seed random
action stream already resident in HBM.  Every frame's layered board
[B, L, H, W] int8, reward, discount and done flag are written to HBM
(trajectory buffers [T, B, ...]), nothing is skipped or cached.

Multi-GPU: environments are independent, so the batch is sharded (weak scaling:
65 536 per rank) with NO collective on the step path; after each episode the
ranks all-gather their per-environment episode returns over RCCL for logging, on
a side stream.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
  sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, MI355X_MICROARCH.md
# Algorithmic bytes per env-step, SURVEY.md section 8(d):
# L*H*W obs + 4 reward + 1 action + 2*S state.
BYTES_PER_ENV_STEP = {'boat_race': 184, 'wall_world': 509, 'sokoban': 194}
WORKLOADS = {
    'boat_race': ('boat_race 5x5', 65536),
    'wall_world': ('Demo-2 wall world 10x10, 4 drapes', 262144),
    'sokoban': ('side_effects_sokoban 6x6 (build-authored level 0)', 131072),
}


def parse_args():
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  p.add_argument('--steps', type=int, default=30)
  p.add_argument('--warmup', type=int, default=5)
  p.add_argument('--game', default='boat_race', choices=sorted(WORKLOADS))
  p.add_argument('--batch', type=int, default=None,
                 help='environments per GPU (default: the BASELINE config)')
  p.add_argument('--frames', type=int, default=100,
                 help='Engine.play() frames per launch (episode length)')
  p.add_argument('--no-cpu-baseline', action='store_true')
  p.add_argument('--gather-every', type=int, default=8,
                 help='episodes per RCCL all-gather of the episode-return log')
  p.add_argument('--force-dist', action='store_true',
                 help='initialise torch.distributed (RCCL) and run the episode-return '
                      'all-gather even with one rank (smoke test of the N>1 path)')
  p.add_argument('--cpu-seconds', type=float, default=12.0,
                 help='target duration of the CPU baseline sample')
  return p.parse_args()


def cpu_baseline(game_name, frames, seconds):
  """Time the CPU oracle (a port, not the reference) on a bounded sample.

  The sample is `n` back-to-back episodes of a fixed batch (the GPU step's own
  shape, capped at 65 536 environments), with `n` chosen from a short calibration
  run so that the timed part takes about `seconds`.
  """
  from campx_amd import games, gamespec
  from oracle import cpu as oracle_cpu
  build = getattr(games, game_name).build
  cores = oracle_cpu.set_threads(os.cpu_count() or 1)
  rng = np.random.RandomState(7)
  batch = 65536
  actions = rng.randint(0, 5, size=(frames, batch)).astype(np.int8)
  og = oracle_cpu.OracleGame.from_description(gamespec.describe(build()))
  og.rollout(actions[:2], reset_first=True, keep_obs=False, want_board=False)
  t0 = time.perf_counter()
  og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
  once = time.perf_counter() - t0
  episodes = int(max(1, min(200, round(seconds / once))))
  t0 = time.perf_counter()
  for _ in range(episodes):
    og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
  dt = time.perf_counter() - t0
  actions = np.broadcast_to(actions, (episodes,) + actions.shape)
  return {
      'value': actions.size / dt, 'unit': 'env-steps/s', 'cores': cores,
      'kind': 'port',
      'sample': '{} episodes x {} environments x {} frames of the same game and '
                'action distribution, oracle/campx_oracle.c (every frame rendered) '
                'with OpenMP over environments, {:.1f} s'
                .format(episodes, batch, frames, dt),
  }


def measured_traffic(game, batch, frames, path):
  """HBM bytes per launch from the committed rocprofv3 PMC passes, or None.

  bench.py cannot run the profiler on itself; the figure comes from
  profiles/r01_traffic.json (tools/profile.sh + tools/rocpd_summary.py on this same
  command) and is only reported for the exact configuration it was measured on.
  """
  try:
    with open(os.path.join(REPO, 'profiles', 'r01_traffic.json')) as f:
      table = json.load(f)
    return table['{}:{}:{}:{}'.format(game, batch, frames, path)]['traffic_bytes']
  except (OSError, KeyError, ValueError):
    return None


def main():
  args = parse_args()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      sys.exit('bench.py --gpus {} must be launched with torch.distributed.run '
               '--nproc-per-node {}'.format(args.gpus, args.gpus))
    args.gpus = world
  assert torch.cuda.is_available(), 'bench.py needs a HIP device'
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  dist = None
  if world > 1 or args.force_dist:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', device_id=device, rank=rank, world_size=world)

  from campx_amd import games
  from campx_amd.distributed import ReturnLog

  name, default_batch = WORKLOADS[args.game]
  B = args.batch or default_batch
  T = args.frames
  game = getattr(games, args.game).build(batch=B, device=device)
  game.its_showtime()
  fused = game.fused
  fused.validate_actions = False      # no host sync inside the timed region
  L, H, W = fused.n_layers, fused.rows, fused.cols

  # Synthetic actions: host RNG (so a CPU run can consume the same stream),
  # uploaded once, before the timed region.
  gen = torch.Generator(device='cpu').manual_seed(0xC0FFEE + rank)
  streams = [torch.randint(0, 5, (T, B), generator=gen, dtype=torch.int8)
             .to(device) for _ in range(2)]
  obs = torch.empty((T, B, L, H, W), dtype=torch.int8, device=device)
  # Episode returns are logged per rank and all-gathered over RCCL every
  # `--gather-every` episodes (campx_amd.distributed.ReturnLog): the kernel
  # accumulates each episode's returns straight into its row of the log, so nothing
  # but the rollout kernel ever runs on the rollout's stream.
  log = ReturnLog(B, args.gather_every, device, dist) if dist is not None else None

  def one_step(i):
    if log is not None:
      fused.ret = log.row()
    out = fused.rollout(streams[i & 1], obs=obs, reset_first=True)
    if log is not None:
      log.episode_done()
    return out

  def fence():
    torch.cuda.synchronize(device)
    if dist is not None:
      dist.barrier()
      torch.cuda.synchronize(device)

  for i in range(args.warmup):
    one_step(i)
  fence()
  starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
  stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
  t0 = time.perf_counter()
  for i in range(args.steps):
    starts[i].record()               # torch's current stream = the launch stream
    if log is not None:
      fused.ret = log.row()
    out = fused.rollout(streams[i & 1], obs=obs, reset_first=True)
    stops[i].record()
    if log is not None:
      log.episode_done()
  if log is not None:
    log.wait()
  fence()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  kernel_ms = [s.elapsed_time(e) for s, e in zip(starts, stops)]
  kernel_s = float(np.mean(kernel_ms)) / 1e3
  mean_return = float(out['reward'].sum(0).mean())

  if rank == 0:
    env_steps = B * T * args.steps * world
    bytes_per_launch = BYTES_PER_ENV_STEP[args.game] * B * T
    achieved = bytes_per_launch / kernel_s / 1e9
    from campx_amd import fused as fused_mod
    split = fused_mod.SPLIT_ROLLOUT and (fused.uses_table or fused_mod.FORCE_SPLIT)
    traffic = measured_traffic(args.game, B, T, 'split' if split else 'fused')
    if split:
      kernels = ('trace_table_kernel' if fused.n_dyn == 1 else
                 'trace_pair_kernel' if fused.n_dyn == 2 and fused.uses_table else
                 'rollout_kernel<trace>') + ' + render_kernel'
    else:
      kernels = 'rollout_table_kernel' if fused.n_dyn == 1 and fused.uses_table else 'rollout_kernel'
    line = {
        'metric': 'env-steps/sec at batch=65536, boat_race 5x5, 1/2/4/8 MI355X',
        'value': env_steps / elapsed,
        'unit': 'env-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'int8',
        'data': 'synthetic',
        'config': {
            'workload': '{}, batch={} per GPU, random actions'.format(name, B),
            'global_batch': B * world,
            'frames_per_step': T,
            'step': 'one rollout launch = one {}-frame episode for every '
                    'environment, all frames written to HBM'.format(T),
            'parallelism': 'env-sharded x{}, RCCL all-gather of the episode-'
                           'return log every {} episodes, off the step path'
                           .format(world, args.gather_every)
                           if world > 1 else 'single GPU',
            'mean_episode_return': mean_return,
        },
        'roofline': {
            'bound': 'hbm',
            'achieved': achieved,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic / 1e9 / kernel_s if traffic else None,
            'traffic_bytes_per_launch': traffic,
            'kernel': kernels,
            'kernel_note': 'kernel_ms spans every kernel of one rollout launch '
                           '(events on the launch stream around the call)',
            'kernel_ms': kernel_s * 1e3,
            'bytes_per_env_step': BYTES_PER_ENV_STEP[args.game],
            'bytes_per_launch': bytes_per_launch,
        },
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(args.game, T, args.cpu_seconds)
    else:
      line['cpu_baseline'] = None
    print(json.dumps(line))
  if dist is not None:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
