#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec, boat race 5x5, 65 536 environments per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 (and no WORLD_SIZE in the environment) this process starts N child
ranks itself - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py ...` - BEFORE anything touches the
GPU, relays their output and exits with their return code.  Started under
torch.distributed.run directly it is one of those ranks (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment).

One "step" of this bench = one pass of the hot path over one batch of synthetic
input = ONE rollout launch: an episode of `--frames` (default 100, the
reference's episode length, examples/reinforce.py:36) consecutive Engine.play()
frames for every environment of the rank's shard, rebuilt from the art at the
start (make_game() per episode, reinforce.py:122), on a committed-seed random
action stream already resident in HBM.  Every frame's layered board
[B, L, H, W] int8, reward, discount, done flag and hidden performance are written
to HBM (trajectory buffers [T, B, ...]), nothing is skipped or cached.

Multi-GPU: environments are independent, so the batch is sharded (weak scaling:
65 536 per rank) with NO collective on the step path; the ranks all-gather their
per-environment episode returns over RCCL for logging, off the step path.

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" explains every field).
"""

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

# (the pool's host driver only supports dmabuf IPC: without this RCCL between processes fails with
# hipIpcGetMemHandle: invalid argument; the image exports it, a rank launched from elsewhere might not)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
  sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, MI355X_MICROARCH.md
# Algorithmic bytes per env-step, SURVEY.md section 8(d):
# L*H*W obs + 4 reward + 1 action + 2*S state.
BYTES_PER_ENV_STEP = {'boat_race': 184, 'wall_world': 509, 'sokoban': 194,
                      # 6x8 boards, 6 / 7 characters, 3 / 4 moving things (S = 6 / 8)
                      'sokoban_l1': 288 + 4 + 1 + 12 + 1, 'sokoban_l2': 336 + 4 + 1 + 16 + 1,
                      # wide tier: 6 characters, one mover (S = 2) + the done byte
                      'maze16': 6 * 256 + 4 + 1 + 4 + 1, 'maze32': 6 * 1024 + 4 + 1 + 4 + 1,
                      # 16x16 sokoban, two boxes: 6 characters, three movers (S = 6)
                      'sokoban16': 6 * 256 + 4 + 1 + 12 + 1,
                      # Hello World 13x36 (shape tier): 7 characters, five things' offsets (S = 10)
                      'hello_world': 7 * 468 + 4 + 1 + 20 + 1,
                      # a user-written coin field 6x10 (state-table tier): 5 characters, the walker's
                      # plane of the trace and the mask of the coins that show (S = 2)
                      'coin_field': 5 * 60 + 4 + 1 + 4 + 1}
WORKLOADS = {
    'boat_race': ('boat_race 5x5', 65536),
    'wall_world': ('Demo-2 wall world 10x10, 4 drapes', 262144),
    'sokoban': ('side_effects_sokoban 6x6 (build-authored level 0)', 131072),
    # not BASELINE configs: games with 3 and 4 moving things (rule interpreter path)
    'sokoban_l1': ('sokoban 6x8 with two boxes (build-authored level 1)', 131072),
    'sokoban_l2': ('sokoban 6x8 with three boxes (build-authored level 2)', 131072),
    # not BASELINE configs: boards above 128 cells (the wide tier, campx_amd/games/maze.py)
    'maze16': ('maze 16x16, 6 characters (build-authored, wide tier)', 65536),
    'maze32': ('maze 32x32, 6 characters (build-authored, wide tier)', 16384),
    # not a BASELINE config: a multi-mover rule game above 128 cells - 4.4 million states
    # enumerated on the device (campx_amd/enumerate_states.py), run by the wide tier
    'sokoban16': ('sokoban 16x16 with two boxes (build-authored level 3, wide tier)', 65536),
    # not a BASELINE config: the reference's Hello World notebook (rigid multi-cell things, trails)
    'hello_world': ('Hello World 13x36, 7 characters, trails (shape tier)', 32768),
    # not a BASELINE config: plain Python update() classes (examples/coins_batched.py without its
    # switch), tabulated on the host during its_showtime(); a drape of three coins that are taken
    # one by one = pieces of the scenery in a 16-bit mask per state (round 6)
    'coin_field': ('coin field 6x10, user-written plain Python classes, three coins in one drape '
                   '(state-table tier, pieces in a mask)', 262144),
}
NO_C_ORACLE = ('coin_field',)      # plain Python classes: no rule description for oracle/campx_oracle.c


def build_game(game_name, **where):
  """The library game behind a --game name (set up, not started)."""
  from campx_amd import games
  if game_name == 'coin_field':
    examples = os.path.join(REPO, 'examples')
    if examples not in sys.path:
      sys.path.insert(0, examples)
    import coins_batched
    return coins_batched.make_game(floor=False, **where)
  if game_name.startswith('sokoban_l'):
    return games.sokoban.build(level=int(game_name[-1]), **where)
  if game_name == 'sokoban16':
    return games.sokoban.build(level=3, **where)
  if game_name.startswith('maze'):
    from campx_amd.games import maze
    n = int(game_name[4:])
    return maze.build(n, n, **where)
  return getattr(games, game_name).build(**where)
HEADLINE_METRIC = 'env-steps/sec at batch=65536, boat_race 5x5, 1/2/4/8 MI355X'
# the latest round's counter passes (tools/gpu_profile_all.sh -> tools/make_traffic.py)
TRAFFIC_FILE = max([os.path.join(REPO, 'profiles', f) for f in os.listdir(os.path.join(REPO, 'profiles'))
                    if f.endswith('_traffic.json')] or [os.path.join(REPO, 'profiles', 'none')])


def parse_args(argv=None):
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  # defaults: the timed region pays a fixed ~0.2 ms after the synchronise that opens it (the
  # chip idles during the fence and ramps back): 30 timed launches read 0.186-0.193 ms per
  # step, 100 read 0.181-0.183, whatever the warm-up (tools/gpu_warm_ab.sh, NOTES.md section 5)
  p.add_argument('--steps', type=int, default=100)
  p.add_argument('--warmup', type=int, default=50)
  p.add_argument('--game', default='boat_race', choices=sorted(WORKLOADS))
  p.add_argument('--batch', type=int, default=None,
                 help='environments per GPU (default: the BASELINE config)')
  p.add_argument('--frames', type=int, default=100,
                 help='Engine.play() frames per launch (episode length)')
  p.add_argument('--no-cpu-baseline', action='store_true')
  p.add_argument('--no-group', action='store_true',
                 help='N = 1 only, for A/B measurements: no process group (the episode-return '
                      'log is copied locally instead of all-gathered)')
  p.add_argument('--no-extras', action='store_true',
                 help='skip play()-mode and the wall_world / sokoban side measurements')
  p.add_argument('--deferred', action='store_true',
                 help='A/B: rollouts pipelined across calls (FusedGame.rollout_deferred): one '
                      'launch holds the update pass of rollout i+1 and the render pass of rollout '
                      'i; every timed step still does one update pass and one render pass')
  p.add_argument('--gather-every', type=int, default=32,
                 help='episodes per RCCL all-gather of the episode-return log')
  p.add_argument('--gather-at', type=float, default=2.0 / 3.0,
                 help='a timed window shorter than --gather-every episodes gathers once, this '
                      'fraction of the way into it')
  p.add_argument('--pg-priority', choices=['high', 'normal'], default='high',
                 help='priority of the stream RCCL runs its collectives on (ProcessGroupNCCL.Options.'
                      'is_high_priority_stream): high = the episode-return gather is dispatched '
                      'ahead of the rollout\'s queued workgroups instead of competing with them')
  p.add_argument('--no-pin', action='store_true',
                 help='do not pin the rank\'s host threads to the NUMA node of its GPU')
  p.add_argument('--episode-csv', default=None,
                 help='rank 0: after the timed region, write the last all-gathered block of '
                      'episode returns in the reference\'s CSV format '
                      '(campx_amd/episode_log.py; examples/reinforce.py:270-284)')
  p.add_argument('--force-dist', action='store_true',
                 help='go through the multi-rank launcher and the RCCL episode-return '
                      'all-gather even with one rank (smoke test of the N>1 path)')
  p.add_argument('--cpu-seconds', type=float, default=12.0,
                 help='target duration of the CPU baseline sample')
  p.add_argument('--standin', default=None,
                 help='TESTS ONLY: "module:function" building a CPU stand-in for the '
                      'batched engine; ranks then use gloo and no GPU, and the line is '
                      'marked as a dry run (tests/test_bench_launcher.py)')
  return p.parse_args(argv)


# ------------------------------------------------------------------- launcher

def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def launch_ranks(n, argv):
  """Start `n` ranks of this script under torch.distributed.run; return their rc.

  Called before this process has made any GPU call (nothing may exec or fork a
  GPU-initialised process on this pool); the children are ordinary subprocesses
  and their stdout (rank 0's JSON line) is relayed line by line.
  """
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  env.setdefault('OMP_NUM_THREADS', '1')
  proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
  for line in proc.stdout:
    sys.stdout.write(line)
    sys.stdout.flush()
  return proc.wait()


def pin_to_gpu_numa_node(local_rank):
  """Pin this process (and the threads it will start) to the CPUs of the NUMA node its GPU hangs
  off, BEFORE anything touches the GPU: on a two-socket node a rank whose launch loop runs on
  the far socket pays a cross-socket hop per doorbell.  Best effort from sysfs - amdgpu cards in
  PCI order, `*_VISIBLE_DEVICES` honoured when it is a plain list of ordinals - and reported in
  the line (`config.numa_node`, `config.cpus_pinned`): None when the node could not be told.
  Returns (node, number of CPUs, the affinity mask to restore before all-core CPU work)."""
  import glob
  import re
  before = None
  try:
    before = os.sched_getaffinity(0)
    ordinal = local_rank
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
      listed = os.environ.get(var)
      if listed and all(x.strip().isdigit() for x in listed.split(',')):
        ids = [int(x) for x in listed.split(',')]
        if local_rank < len(ids):
          ordinal = ids[local_rank]
        break
    cards = []
    for d in glob.glob('/sys/class/drm/card*'):
      if not re.fullmatch(r'card\d+', os.path.basename(d)):
        continue
      dev = os.path.join(d, 'device')
      try:
        with open(os.path.join(dev, 'vendor')) as f:
          if f.read().strip() != '0x1002':
            continue
        with open(os.path.join(dev, 'numa_node')) as f:
          node = int(f.read().strip())
      except (OSError, ValueError):
        continue
      cards.append((os.path.basename(os.path.realpath(dev)), node))
    cards.sort()
    if ordinal >= len(cards) or cards[ordinal][1] < 0:
      return None, None, before
    node = cards[ordinal][1]
    with open('/sys/devices/system/node/node{}/cpulist'.format(node)) as f:
      cpus = set()
      for part in f.read().strip().split(','):
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    cpus &= before
    if not cpus:
      return node, None, before
    os.sched_setaffinity(0, cpus)
    return node, len(cpus), before
  except (OSError, ValueError, AttributeError):
    return None, None, before


def measured_write_ceiling(buf, device, launches=20):
  """What this chip sustains for a pure stream of stores over `buf` (the observation buffer of
  the rollout just timed: exactly the bytes a launch writes), GB/s - the denominator SURVEY
  section 8(d) asks for beside the 8 TB/s vendor peak.  Two probes, HIP events on the current
  stream round `launches` back-to-back launches each: the library's own streaming-store kernel
  (campx_write_probe_launch: the render kernel's store form and block order, nothing to compute)
  and torch's `fill_`; the ceiling is the better of the two."""
  import ctypes
  import torch
  from campx_amd import _hip
  flat = buf.view(-1).view(torch.uint8)
  n = flat.numel() // 16 * 16
  stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

  def probe():
    _hip.check(_hip.lib.campx_write_probe_launch(ctypes.c_void_p(flat.data_ptr()), n, 0x01010101, stream),
               'campx_write_probe_launch')

  def fill():
    flat.fill_(1)

  rates = {}
  for name, fn in (('probe', probe), ('fill', fill)):
    for _ in range(5):
      fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
      fn()
    e1.record()
    e1.synchronize()
    rates[name] = n / (e0.elapsed_time(e1) / launches / 1e3) / 1e9
  return rates


def short_row(value, ms_per_step, roof, cpu):
  """One configuration as a string short enough to survive the driver's record whole (its parser
  keeps scalar keys only and cuts strings at 120 characters): everything a reader needs to
  recompute the roofline fraction - value, ms per step, kernel ms, bytes per env-step (x batch x
  frames of the key's name = bytes per launch), fraction of peak / of the measured ceiling,
  counter traffic over algorithmic bytes, CPU baseline / cores."""
  traffic = roof.get('traffic')
  return 'v={:.4g} ms={:.4f} kms={:.4f} Bps={} frac={:.4f} ofmeas={} traf={} cpu={}'.format(
      value, ms_per_step, roof['kernel_ms'], roof['bytes_per_env_step'], roof['frac'],
      'n/a' if roof.get('frac_of_measured') is None else '{:.3f}'.format(roof['frac_of_measured']),
      'n/a' if not traffic else '{:.3f}'.format(traffic / roof['bytes_per_launch']),
      'n/a' if not cpu else '{:.3g}/{}c'.format(cpu['value'], cpu['cores']))


# ------------------------------------------------------------ CPU baselines

def cpu_baseline(game_name, frames, seconds, batch=65536):
  """Time the CPU oracle (a port, not the reference) on a bounded sample.

  The sample is `n` back-to-back episodes of `batch` environments (the GPU step's own
  shape: BASELINE.md section 3 item 2), with `n` chosen from a short calibration run so
  that the timed part takes about `seconds`.
  """
  import numpy as np
  from campx_amd import gamespec
  from oracle import cpu as oracle_cpu
  cores = oracle_cpu.set_threads(os.cpu_count() or 1)
  rng = np.random.RandomState(7)
  actions = rng.randint(0, 5, size=(frames, batch)).astype(np.int8)
  og = oracle_cpu.OracleGame.from_description(gamespec.describe(build_game(game_name)))
  og.rollout(actions[:2], reset_first=True, keep_obs=False, want_board=False)
  t0 = time.perf_counter()
  og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
  once = time.perf_counter() - t0
  episodes = int(max(1, min(200, round(seconds / once))))
  t0 = time.perf_counter()
  for _ in range(episodes):
    og.rollout(actions, reset_first=True, keep_obs=False, want_board=False)
  dt = time.perf_counter() - t0
  return {
      'value': episodes * actions.size / dt, 'unit': 'env-steps/s', 'cores': cores,
      'kind': 'port',
      'sample': '{} episodes x {} environments x {} frames of the same game and '
                'action distribution, oracle/campx_oracle.c (every frame rendered) '
                'with OpenMP over environments, {:.1f} s'
                .format(episodes, batch, frames, dt),
  }


def generic_b1(seconds=3.0):
  """This repo's generic tier (Python update() bodies, torch CPU ops, B = 1).

  The execution model of the reference itself (campx/engine.py:114-324), on this
  host: the number to hold against the reference's own 955 env-steps/s
  (measured in the build container).
  """
  import torch
  from campx_amd.games import boat_race
  torch.set_num_threads(1)
  game, _, _, _ = boat_race.make_game()
  acts = [torch.nn.functional.one_hot(torch.tensor(a), 5).float()
          for a in (1, 1, 3, 3, 0, 0, 2, 2)]
  for i in range(20):
    game.play(acts[i % 8])
  n, t0 = 0, time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    for i in range(50):
      game.play(acts[i % 8])
    n += 50
  dt = time.perf_counter() - t0
  return {'value': n / dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
          'sample': '{} Engine.play() calls of the boat race on the generic tier '
                    '(batch=None), one-hot actions, {:.1f} s'.format(n, dt)}


# -------------------------------------------------------------- measurements

def measured_traffic(game, batch, frames, path):
  """HBM bytes per launch from the committed rocprofv3 PMC passes, or None.

  bench.py cannot run the profiler on itself; the figure comes from
  profiles/rNN_traffic.json (tools/profile.sh + tools/rocpd_summary.py on this same
  command; the latest round's file) and is only reported for the exact configuration it was measured on.
  """
  try:
    with open(TRAFFIC_FILE) as f:
      table = json.load(f)
    return table['{}:{}:{}:{}'.format(game, batch, frames, path)]['traffic_bytes']
  except (OSError, KeyError, ValueError):
    return None


def kernel_names(fused, split, B=None, T=None):
  if type(fused).__name__ == 'WideGame':
    return 'wide_update_kernel + render_kernel'
  if type(fused).__name__ == 'ShapeGame':
    return ('shape_update_split_kernel + shape_render_split_kernel' if fused._tables is not None
            else 'shape_rollout_kernel')
  if split:
    # (one-mover games at small batches: update pass and render share ONE launch - asked of the
    # library itself, campx_flow_shared: its bounds and knobs, not a copy of them)
    one = getattr(fused, '_one_launch', None)
    if one is not None and B is not None and one(T, (B + 15) // 16 * 16):
      return ('pipe_table_kernel<true>' if fused.n_dyn == 1 else 'pipe_multi_kernel<K, ., true>') + \
          ' (update pass + render in one launch)'
    first = ('update_table_kernel' if fused.n_dyn == 1 else
             'update_pair_kernel' if fused.n_dyn == 2 and fused.uses_table else
             'update_tuple_kernel' if fused.uses_table else 'rollout_kernel<trace>')
    return first + ' + render_kernel'
  return ('rollout_table_kernel' if fused.n_dyn == 1 and fused.uses_table
          else 'rollout_kernel')


def measure_rollout(game_name, B, T, steps, warmup, device, rank, dist, gather_every,
                    standin=None, deferred=False, with_log=False, gather_at=2.0 / 3.0):
  """Warm up, then time exactly `steps` rollout launches.  Returns a dict.

  Wall clock: perf_counter around the timed region, bracketed by synchronize +
  barrier on both sides.  Kernel time: ONE pair of HIP events on the launch stream
  around the same `steps` launches (mean per launch, inter-launch gaps included), so
  the timed region carries no per-launch event packets.  A second, untimed pass of
  the same launches with an event pair per launch gives the median / min / max.
  """
  import numpy as np
  import torch
  on_gpu = standin is None
  if on_gpu:
    game = build_game(game_name, batch=B, device=device)
  else:
    game = standin(game_name, B)
  game.its_showtime()
  fused = game.fused
  fused.validate_actions = False      # no host read-back inside the timed region

  # Synthetic actions: host RNG (so a CPU run can consume the same stream),
  # uploaded once, before the timed region.
  gen = torch.Generator(device='cpu').manual_seed(0xC0FFEE + rank)
  streams = [torch.randint(0, 5, (T, B), generator=gen, dtype=torch.int8)
             .to(device) for _ in range(2)]
  # Output buffers are allocated once; a step is then op dispatches only.  Deferred
  # rollouts (the update pass of launch i+1 shares a launch with the observation stream of
  # launch i) alternate two sets of scalars / trace over one observation buffer.
  bufs = [fused.rollout_buffers(T)]
  deferred = bool(deferred) and bufs[0].get('trace') is not None and hasattr(fused, 'rollout_deferred')
  bufs.append(fused.rollout_buffers(T, share=bufs[0]) if deferred else bufs[0])
  log = None
  if dist is not None or with_log:
    # Episode returns are logged per rank and all-gathered every `gather_every`
    # episodes (campx_amd.distributed.ReturnLog): the kernel accumulates each
    # episode's returns straight into its row of the log, so nothing but the
    # rollout kernels ever runs on the rollout's stream.  (Without a process group -
    # RCCL could not be initialised on a one-GPU run - the "gather" is a local copy.)
    from campx_amd.distributed import ReturnLog
    # (never rarer than once per timed region, so that a short run still exercises the gather -
    # and then two thirds into it, with launches still to come as in a long run, not at its
    # very end, where the whole all-gather would sit exposed in front of the closing fence)
    gather_every = max(1, min(gather_every, int(gather_at * steps + 0.67)))
    log = ReturnLog(B, gather_every, device, dist)

  n_calls = [0]

  def one_step(i):
    if log is not None:
      fused.ret = log.row()
    if deferred:
      # update pass of this rollout + render pass of the one before it, one launch (the two
      # buffer sets strictly alternate, whatever `i` the caller's loops restart from)
      k = n_calls[0] & 1
      n_calls[0] += 1
      # (the action streams are resident and complete: the promise that lets multi-mover games
      # past the shared launch's bounds run their update pass ahead, on the side stream)
      fused.rollout_deferred(streams[k], bufs[k], reset_first=True, actions_ready=True)
      out = bufs[k]
    else:
      out = fused.rollout(streams[i & 1], out=bufs[i & 1], reset_first=True)
    if log is not None:
      log.episode_done()
    return out

  # (A host-side barrier - a gloo group beside the RCCL one - was tried for the fence: 196-242 us
  # at the end of a timed window on one rank against 70-96 us for the RCCL barrier; a tight loop
  # of either reads ~20 us, tools/barrier_cost.py, but the window's closing barrier is a cold one.)
  def fence():
    if on_gpu:
      torch.cuda.synchronize(device)
    if dist is not None:
      dist.barrier()
      if on_gpu:
        torch.cuda.synchronize(device)

  for i in range(warmup):
    one_step(i)
  # Settle: whatever --warmup was, the timed window opens only after at least 50 ms of GPU
  # time in this process and on these buffers (a side measurement that starts right after
  # empty_cache() and a new game otherwise reads several per cent low: round 2's sokoban
  # read 0.790 of peak in the driver's run against 0.859 in its profile).  Untimed; the
  # count is reported as config.settle_launches; `steps` / `warmup` stay as given.
  settle = 0
  # (the collector runs BEFORE the settle launches, and stays off until the window has closed:
  # round 6 found the driver's 20-launch windows of the side measurements 7-9 % slower than the
  # same launches a moment later - sokoban 0.401 against 0.368 ms, Hello World 1.79 against 1.68 -
  # because gc.collect() sat between the last settle launch and the fence: by then a process that
  # has built six games takes longer to collect than the queued launches take to run, the chip
  # idled for tens of milliseconds, and the window opened on lowered clocks)
  gc.collect()
  gc.disable()
  if on_gpu:
    probe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    probe[0].record()
    for i in range(4):
      one_step(warmup + i)
    probe[1].record()
    probe[1].synchronize()
    per = max(probe[0].elapsed_time(probe[1]) / 4, 1e-3)
    settle = 4 + int(min(5000, max(0, 50.0 / per - 4)) + 0.999)
    for i in range(4, settle):
      one_step(warmup + i)
  if log is not None:
    log.wait()
    log.align()                       # the timed window starts on a block boundary of the log
  if log is not None and on_gpu and os.environ.get('CAMPX_BENCH_NO_GATHER_TIMING') != '1':
    log.time_gathers()                # the gathers of the timed window leave their start / end events
  fence()                             # (nothing between the last settle launch and the fence but the log's bookkeeping)
  if on_gpu:
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()                      # torch's current stream = the launch stream
  t0 = time.perf_counter()
  for i in range(steps):
    out = one_step(i)
  if on_gpu:
    ev1.record()
  t_loop = time.perf_counter()
  t_launches = t_open = t_loop
  if on_gpu:
    # The last launch has finished: was every gather done by then?  The host SPINS on the two
    # events (hipEventQuery) instead of sleeping in a synchronise: one driver-style run in twenty of
    # round 6 read 13.0 ms here for launches the device's clock put at 3.68 ms (`ms_per_step` 3.7 x
    # `kernel_ms`; round 4's driver line, 1.219 x, was of this kind) - either the sleeping thread
    # was woken late, which a spin does not suffer, or the device STARTED late, which the time the
    # opening event is seen to have passed (`window_open_seen_us`, normally within the loop) tells.
    while not ev0.query():
      pass
    t_open = time.perf_counter()
    while not ev1.query():
      pass
    t_launches = time.perf_counter()
  if log is not None:
    log.poll()
    log.wait()
  t_wait = time.perf_counter()
  if on_gpu:
    torch.cuda.synchronize(device)
  t_sync = time.perf_counter()
  # The clock stops here: every launch of this rank and its share of the return log have
  # completed.  The window OPENED barrier-aligned, and what is reported is the MAX over ranks
  # of these per-rank times (all-reduced below) - the time at which the slowest rank was done,
  # which is all a closing barrier inside the clock could add, plus the barrier's own cold
  # 70-100 us (round 3's line carried it: ms_per_step 3.5 % above kernel_ms).  The barrier
  # itself still closes the region, outside the clock.
  elapsed = t_sync - t0
  fence()
  if os.environ.get('CAMPX_BENCH_DEBUG'):
    sys.stderr.write('TIMED WINDOW: loop (host) %.1f us, log.wait %.1f, synchronize %.1f, total %.1f; closing fence (off the clock) %.1f\n' % (
        (t_loop - t0) * 1e6, (t_wait - t_loop) * 1e6, (t_sync - t_wait) * 1e6,
        elapsed * 1e6, (time.perf_counter() - t_sync) * 1e6))
  own_elapsed = elapsed               # this rank's; `elapsed` becomes the max over ranks
  gc.enable()
  # where the window's time went, on the host's clock, and when its gathers ran, on the device's
  # (relative to the event that opens the window on the launch stream)
  if log is not None:
    log.poll()                        # (whatever was not seen complete before: stamped now, at the end)
  window_us = {'loop': (t_loop - t0) * 1e6, 'open_seen': (t_open - t0) * 1e6, 'launches': (t_launches - t_loop) * 1e6,
               'log_wait': (t_wait - t_launches) * 1e6, 'synchronize': (t_sync - t_wait) * 1e6,
               'total': elapsed * 1e6}
  gathers = None
  if log is not None and log.timing is not None:
    # per gather of the window: when the block was complete (device clock, from the window's
    # opening event), when the collective was issued and how long the call took the host, and
    # when it was first SEEN complete (host clock from the window's start, asked once per
    # launch: a gather still running after the last launch shows here)
    gathers = [{'after_episode': rec['count'] - log.timing[0]['count'] + gather_every,
                'ready_us': ev0.elapsed_time(rec['ready']) * 1e3,
                'issued_us': (rec['issued'] - t0) * 1e6,
                'call_us': (rec['returned'] - rec['issued']) * 1e6,
                'seen_done_us': None if rec['seen_done'] is None else (rec['seen_done'] - t0) * 1e6,
                # (asked right after the last launch finished: a gather that still ran then is
                # what stretches the window)
                'done_when_launches_ended': rec['seen_done'] is not None and rec['seen_done'] <= t_launches + 2e-5}
               for rec in log.timing]
    window_us['launches_done'] = ev0.elapsed_time(ev1) * 1e3
    log.timing = None
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  result = dict(fused=fused, elapsed=elapsed, own_elapsed=own_elapsed, log=log, out=out,
                pipelined='deferred' if deferred else False, settle=settle, window_us=window_us, gathers=gathers,
                gather_every=gather_every if log is not None else None,
                mean_return=float(out['reward'].sum(0).mean())
                if out['reward'] is not None else None)
  if on_gpu:
    result['kernel_ms'] = ev0.elapsed_time(ev1) / steps
    # untimed second pass: one event pair per launch
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for i in range(steps):
      starts[i].record()
      one_step(i)
      stops[i].record()
    if log is not None:
      log.wait()
    fence()
    if deferred:
      fused.flush()                   # the last rollout's observations
      fence()
    per = [s.elapsed_time(e) for s, e in zip(starts, stops)]
    result['per_launch_ms'] = {'median': float(np.median(per)), 'min': float(min(per)),
                               'max': float(max(per)), 'mean': float(np.mean(per))}
    # ... and, with the chip as warm as it was for the timed launches, what it sustains for a
    # pure stream of stores over the same observation buffer
    obs = out['obs']
    result['ceiling'] = measured_write_ceiling(obs, device) if obs.dim() == 5 else None
  return result


def roofline(game_name, B, T, fused, kernel_ms, per_launch, ceiling=None):
  from campx_amd import fused as fused_mod
  bytes_per_launch = BYTES_PER_ENV_STEP[game_name] * B * T
  achieved = bytes_per_launch / (kernel_ms / 1e3) / 1e9
  split = fused_mod.SPLIT_ROLLOUT
  traffic = measured_traffic(game_name, B, T, 'split' if split else 'fused')
  return {
      'bound': 'hbm',
      'achieved': achieved,
      'peak': HBM_PEAK_GBS,
      'unit': 'GB/s',
      'frac': achieved / HBM_PEAK_GBS,
      # SURVEY 8(d) "Bound": the box's measured streaming-write bandwidth over the launch's own
      # observation buffer, timed in this run right after the launches (measured_write_ceiling)
      'measured_write_ceiling_gbs': None if not ceiling else max(ceiling.values()),
      'ceiling_probe_gbs': None if not ceiling else ceiling['probe'],
      'ceiling_fill_gbs': None if not ceiling else ceiling['fill'],
      'frac_of_measured': None if not ceiling else achieved / max(ceiling.values()),
      'traffic': traffic,
      'traffic_over_algorithmic': None if not traffic else traffic / bytes_per_launch,
      'traffic_note': 'HBM bytes per launch (WRITE_SIZE + FETCH_SIZE, separate '
                      'rocprofv3 --pmc passes, profiles/' + os.path.basename(TRAFFIC_FILE) + ')',
      'kernel': kernel_names(fused, split, B, T),
      'kernel_note': 'kernel_ms = HIP-event time on the launch stream around the timed '
                     'launches / steps: every kernel of a rollout launch plus the gaps '
                     'between launches',
      'kernel_ms': kernel_ms,
      'per_launch_ms': per_launch,
      'bytes_per_env_step': BYTES_PER_ENV_STEP[game_name],
      'bytes_per_launch': bytes_per_launch,
  }


def play_mode(device, B=65536, calls=2000):
  """Engine.play() per frame (one launch per frame, state round-trips HBM)."""
  import torch
  from campx_amd.games import boat_race
  game, _, _, _ = boat_race.make_game(batch=B, device=device)
  acts = torch.randint(0, 5, (64, B), dtype=torch.int8, device=device)
  rows = [acts[i] for i in range(64)]
  modes = {}
  for name, validate in (('validate_lazy', True), ('validate_off', False),
                         ('validate_sync', 'sync')):
    game.fused.validate_actions = validate
    for i in range(50):
      game.play(rows[i & 63])
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(calls):
      game.play(rows[i & 63])
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / calls
    modes[name] = {'us_per_call': dt * 1e6, 'env_steps_per_s': B / dt}
  # the same frames captured once in a HIP graph (Engine.capture_play, campx_amd/play_graph.py:
  # campx::step only enqueues kernels on the current stream): one launch of the host's per 32
  # frames instead of 32 op dispatches
  game.fused.validate_actions = False
  graph = game.capture_play(32)
  graph.actions.copy_(acts[:32])
  for _ in range(5):
    graph.replay()
  torch.cuda.synchronize(device)
  t0 = time.perf_counter()
  for _ in range(calls // 32):
    graph.replay()
  torch.cuda.synchronize(device)
  dt = (time.perf_counter() - t0) / (calls // 32 * 32)
  modes['hip_graph_32_frames'] = {'us_per_call': dt * 1e6, 'env_steps_per_s': B / dt}
  # ... and the closed loop of examples/reinforce.py:136-149 - policy forward (its one-hidden-layer
  # network, reinforce.py:53-67, in bf16 straight from the engine's bf16 observation), sampling,
  # play() - op by op from Python, and as one graph of 32 frames
  game.fused.set_play_obs_dtype(torch.bfloat16)
  n_in = game.fused.n_layers * game.fused.rows * game.fused.cols
  net = torch.nn.Sequential(torch.nn.Linear(n_in, 32), torch.nn.ReLU(), torch.nn.Linear(32, 5)).to(
      device=device, dtype=torch.bfloat16)

  def policy(obs, t):
    logits = net(obs.layered_board.view(B, n_in))
    return torch.multinomial(torch.softmax(logits.float(), dim=-1), 1).squeeze(1)

  frames = max(32, calls // 8 // 32 * 32)
  with torch.no_grad():
    for i in range(20):
      game.play(policy(game.fused._observation_cache, i).to(torch.int8))
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(frames):
      game.play(policy(game.fused._observation_cache, i).to(torch.int8))
    torch.cuda.synchronize(device)
  dt = (time.perf_counter() - t0) / frames
  modes['policy_eager'] = {'us_per_frame': dt * 1e6, 'env_steps_per_s': B / dt}
  closed = game.capture_play(32, policy=policy)
  for _ in range(3):
    closed.replay()
  torch.cuda.synchronize(device)
  t0 = time.perf_counter()
  for _ in range(frames // 32):
    closed.replay()
  torch.cuda.synchronize(device)
  dt = (time.perf_counter() - t0) / frames
  modes['policy_in_graph'] = {'us_per_frame': dt * 1e6, 'env_steps_per_s': B / dt}
  return {'workload': 'boat_race 5x5, batch={}, Engine.play() per frame through '
                      'campx::step, {} calls'.format(B, calls), **modes}


def run_rank(args):
  # stdout carries ONE line, rank 0's JSON.  Libraries print there too (RCCL writes a version
  # banner through C stdio, flushed whenever it likes - it came out AFTER the JSON line in a
  # first run), so file descriptor 1 is pointed at stderr for the life of the rank and the
  # line goes to a private duplicate of the real stdout at the very end.
  sys.stdout.flush()
  real_stdout = os.fdopen(os.dup(1), 'w')
  os.dup2(2, 1)
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  # (before torch is imported and long before the first GPU call: the launch loop's thread and
  # every thread started from here on stay on the socket the rank's GPU hangs off)
  numa_node, cpus_pinned, affinity_before = (None, None, None) if args.no_pin else \
      pin_to_gpu_numa_node(local_rank)
  import torch
  standin = None
  if args.standin:
    import importlib
    mod, fn = args.standin.split(':')
    standin = getattr(importlib.import_module(mod), fn)
    device = torch.device('cpu')
  else:
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    if local_rank >= torch.cuda.device_count():
      raise SystemExit('bench.py: rank {} wants GPU {} but this node shows {} HIP device(s)'
                       .format(rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
  # Every run - one rank too - goes through the process group and the episode-return log,
  # so that the N = 1 line times the same protocol as the ranks of an N > 1 run (the driver
  # computes scaling efficiency from those lines).
  dist, group_note = None, None
  if (world > 1 or args.force_dist or standin is None) and not (args.no_group and world == 1):
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(_free_port()))
    try:
      if standin is not None:
        dist.init_process_group('gloo', rank=rank, world_size=world)
      else:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=args.pg_priority == 'high')
        dist.init_process_group('nccl', device_id=device, rank=rank, world_size=world,
                                pg_options=opts)
      world = dist.get_world_size()      # as the process group reports it
    except Exception as e:               # noqa: BLE001 - one rank can do without
      if world > 1 or args.force_dist:
        raise
      group_note = 'RCCL process group could not be initialised ({}: {}); the episode-' \
                   'return log is copied locally instead of all-gathered'.format(
                       type(e).__name__, str(e)[:120])
      dist = None

  name, default_batch = WORKLOADS[args.game]
  B = args.batch or default_batch
  T = args.frames
  m = measure_rollout(args.game, B, T, args.steps, args.warmup, device, rank, dist,
                      args.gather_every, standin, args.deferred,
                      with_log=True, gather_at=args.gather_at)
  fused, elapsed = m['fused'], m['elapsed']

  gathered_ok = None
  if m['log'] is not None:
    block = m['log'].wait()            # [world, episodes, B] of the last full block
    if block is not None:
      mine = m['log'].last_local_block()
      gathered_ok = bool(block.shape[0] == world and torch.equal(block[rank], mine))
  # Per-rank figures for whoever reads the line - the first N > 1 run on hardware will be read
  # blind: a straggler, a rank whose launches were slow (its kernel_ms) as against one whose
  # WINDOW was slow (ms_per_step well above kernel_ms: host, gather, fence), a gather that was
  # still running when the launches ended (exposed_us), a rank whose part of the gathered log is
  # wrong.  All of it as scalars in `config` (the driver's record keeps scalars).
  own_kernel_ms = m.get('kernel_ms', m['own_elapsed'] / args.steps * 1e3)
  launches_end_us = m['window_us']['loop'] + m['window_us']['launches']
  call_us = max([g['call_us'] for g in m['gathers']] or [0.0]) if m['gathers'] else 0.0
  exposed_us = 0.0
  for g in m['gathers'] or []:
    if not g['done_when_launches_ended']:
      seen = g['seen_done_us'] if g['seen_done_us'] is not None else m['window_us']['total']
      exposed_us = max(exposed_us, seen - launches_end_us)
  per_rank = [[m['own_elapsed'] / args.steps * 1e3, own_kernel_ms, call_us, exposed_us,
               -1.0 if gathered_ok is None else float(gathered_ok),
               -1.0 if numa_node is None else float(numa_node)]]
  if dist is not None:
    mine = torch.tensor(per_rank[0], dtype=torch.float64, device=device)
    everyone = torch.zeros((world, mine.numel()), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(everyone.view(-1), mine)
    per_rank = [[float(x) for x in row] for row in everyone.cpu()]
  per_rank_ms = [row[0] for row in per_rank]
  per_rank_kernel_ms = [row[1] for row in per_rank]
  per_rank_ok = [None if row[4] < 0 else bool(row[4]) for row in per_rank]
  gathered_ok = None if any(x is None for x in per_rank_ok) else all(per_rank_ok)
  worst = max(range(len(per_rank)), key=lambda r: per_rank_ms[r])

  if rank == 0 and args.episode_csv and m['log'] is not None:
    block = m['log'].wait()
    if block is not None:
      from campx_amd.episode_log import EpisodeCsvLog
      csv_log = EpisodeCsvLog(args.episode_csv, run_id=0, frames_per_episode=T)
      csv_log.block(block.cpu(), seconds=elapsed / args.steps * block.shape[1])
      csv_log.close()

  if rank == 0:
    env_steps = B * T * args.steps * world
    headline = args.game == 'boat_race' and B == 65536
    line = {
        'metric': HEADLINE_METRIC if headline else
                  'env-steps/sec at batch={}, {}, 1/2/4/8 MI355X'.format(B, name),
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
        'data': 'synthetic' if standin is None else
                'synthetic; CPU STAND-IN DRY RUN of the launcher, not a measurement',
        'config': {
            'workload': '{}, batch={} per GPU, random actions'.format(name, B),
            'global_batch': B * world,
            'frames_per_step': T,
            'step': 'one rollout launch = one {}-frame episode for every '
                    'environment, all frames written to HBM'.format(T),
            'pipelined': m['pipelined'],
            'parallelism': 'env-sharded x{}, {} all-gather of the episode-'
                           'return log every {} episodes, off the step path'
                           .format(world, 'gloo' if standin else 'RCCL', m['gather_every'])
                           if dist is not None else
                           'single GPU, no process group: ' + str(group_note),
            'world': world,
            'rccl_world': world if (dist is not None and standin is None) else None,
            'gather_every': m['gather_every'],
            'settle_launches': m['settle'],
            # every launch issued before the timed window opened (`warmup` is the argument;
            # the settle launches make sure the chip and these buffers have seen 50 ms of work)
            'untimed_launches_before_window': args.warmup + m['settle'],
            # rank 0's timed window on the host's clock: the loop that issues the launches, the
            # wait for the last launch to finish, for the log's last gather, the synchronise that
            # closes the window; `window_launches_done_us`: the launches on the DEVICE's clock
            'window_loop_us': m['window_us']['loop'],
            # (host time at which the event that OPENS the window on the launch stream was seen to
            # have passed, looked for once the loop is done: about `window_loop_us` unless the
            # device started the window's first launch late)
            'window_open_seen_us': m['window_us']['open_seen'],
            'window_launches_us': m['window_us']['launches'],
            'window_log_wait_us': m['window_us']['log_wait'],
            'window_synchronize_us': m['window_us']['synchronize'],
            'window_total_us': m['window_us']['total'],
            'window_launches_done_us': m['window_us'].get('launches_done'),
            # the all-gathers of the timed window, over all ranks: how many (rank 0), the dearest
            # call on the host, the longest any of them was still running after a rank's last
            # launch had finished (0: every gather was hidden under launches)
            'gather_count': len(m['gathers'] or []),
            'gather_call_us_max': max(row[2] for row in per_rank),
            'gather_exposed_us_max': max(row[3] for row in per_rank),
            # per rank (lists for a human; the scalars below are what the driver's record keeps)
            'per_rank_ms_per_step': per_rank_ms,
            'per_rank_kernel_ms': per_rank_kernel_ms,
            'per_rank_numa_node': [None if row[5] < 0 else int(row[5]) for row in per_rank],
            'worst_rank': worst,
            'worst_rank_ms_per_step': per_rank_ms[worst],
            'worst_rank_kernel_ms': per_rank_kernel_ms[worst],
            'best_rank_kernel_ms': min(per_rank_kernel_ms),
            # how much of the slowest rank's window the fastest rank's kernels account for: what
            # scaling efficiency would be if only the kernels counted (1.0 = nothing but kernels)
            'efficiency_vs_own_kernel': min(per_rank_kernel_ms) / max(per_rank_ms),
            'numa_node': numa_node,
            'cpus_pinned': cpus_pinned,
            'gathered_log_matches_local': gathered_ok,
            'per_rank_gathered_log_matches_local': per_rank_ok,
            'mean_episode_return': m['mean_return'],
        },
    }
    details = {'window_us': m['window_us'], 'gathers': m['gathers']}
    if standin is None:
      from campx_amd import _hip
      line['config']['library_settings'] = _hip.config_string()
    if standin is None:
      line['roofline'] = roofline(args.game, B, T, fused, m['kernel_ms'],
                                  m['per_launch_ms'], m.get('ceiling'))
      line['config']['launch_ms_median'] = m['per_launch_ms']['median']
    solo = world == 1 and standin is None and not args.force_dist
    if affinity_before is not None:
      os.sched_setaffinity(0, affinity_before)      # the CPU baselines below use every core
    if solo and not args.no_cpu_baseline and args.game not in NO_C_ORACLE:
      line['cpu_baseline'] = cpu_baseline(args.game, T, args.cpu_seconds,
                                          batch=4096 if args.game.startswith(('maze', 'sokoban16', 'hello')) else 65536)
      line['cpu_baseline']['generic_b1'] = generic_b1()
      line['cpu_baseline']['reference_b1_build_container'] = {
          'value': 954.8, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'reference',
          'sample': 'the reference itself, boat race B=1, measured in the build '
                    'container (BASELINE.md section 2); it cannot run on the GPU box'}
    else:
      line['cpu_baseline'] = None
    if solo and not args.no_extras and headline:
      del m
      torch.cuda.empty_cache()
      also = []
      # (maze16, sokoban16: not BASELINE configs - the wide tier, boards above 128 cells; the
      # second one's 4.4 M-state table is enumerated on the device during its_showtime())
      for other in ('wall_world', 'sokoban', 'maze16', 'sokoban16', 'hello_world', 'coin_field'):
        oname, ob = WORKLOADS[other]
        steps = args.steps
        om = measure_rollout(other, ob, T, steps, args.warmup, device, 0, None, 0,
                             deferred=args.deferred)
        row = {
            'workload': '{}, batch={}, random actions, {} frames per launch'.format(
                oname, ob, T),
            'value': ob * T * steps / om['elapsed'], 'unit': 'env-steps/s',
            'steps': steps, 'ms_per_step': om['elapsed'] / steps * 1e3,
            'settle_launches': om['settle'],
            'roofline': roofline(other, ob, T, om['fused'], om['kernel_ms'],
                                 om['per_launch_ms'], om.get('ceiling')),
            'cpu_baseline': None if args.no_cpu_baseline or other in NO_C_ORACLE else
                            cpu_baseline(other, T, args.cpu_seconds / 2,
                                         batch=4096 if other.startswith(('maze', 'sokoban16', 'hello')) else ob)}
        for note in ('traffic_note', 'kernel_note'):      # (said once, in the headline's roofline)
          row['roofline'].pop(note, None)
        if row['cpu_baseline']:
          details.setdefault('cpu_samples', {})[other] = row['cpu_baseline'].pop('sample')
        # the same row as ONE short string among the headline's scalars: the driver's record
        # keeps those and drops this list (BENCH_r05: sokoban's number was not in it at all)
        line['config']['{}_{}'.format(other, ob)] = short_row(
            row['value'], row['ms_per_step'], row['roofline'], row['cpu_baseline'])
        also.append(row)
        del om
        torch.cuda.empty_cache()
      line['also'] = also
      line['play_mode'] = play_mode(device)
      line['config']['play_us_per_call'] = line['play_mode']['validate_off']['us_per_call']
      line['config']['play_graph32_us_per_call'] = line['play_mode']['hip_graph_32_frames']['us_per_call']
      line['config']['policy_eager_us_per_frame'] = line['play_mode']['policy_eager']['us_per_frame']
      line['config']['policy_in_graph_us_per_frame'] = line['play_mode']['policy_in_graph']['us_per_frame']
      if not args.deferred:
        # Rollouts pipelined across calls (FusedGame.rollout_deferred: ONE launch = the update
        # pass of rollout i+1 + the render pass of rollout i), beside the headline's two
        # launches per rollout, at the batches where the library shares the launch.  Reported, not
        # the headline: it changes what a caller gets back when (observations one call late).
        rows = []
        for db in (32768, 16384, 4096):
          dsteps = args.steps * max(1, B // db)      # (the same env-steps per timed window)
          dm = measure_rollout(args.game, db, T, dsteps, args.warmup, device, 0, None, 0,
                               deferred=True)
          rows.append({'batch': db, 'steps': dsteps, 'value': db * T * dsteps / dm['elapsed'],
                       'unit': 'env-steps/s', 'ms_per_step': dm['elapsed'] / dsteps * 1e3,
                       'kernel_ms': dm['kernel_ms'],
                       'frac': BYTES_PER_ENV_STEP[args.game] * db * T / (dm['kernel_ms'] / 1e3) / 1e9
                       / HBM_PEAK_GBS})
          del dm
          torch.cuda.empty_cache()
        # (rollout_deferred(): one launch per step = update pass of rollout i+1 + render pass of
        # rollout i, pipe_table_kernel; same work per step as the headline, observations delivered
        # one call late; these rows run without the episode-return log of the headline)
        line['deferred_rollouts'] = {'rows': rows}
        for r in rows:
          line['config']['deferred_{}_frac'.format(r['batch'])] = r['frac']
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
  if rank == 0:
    # what does not fit one line a parser keeps whole: on stderr, one JSON line of its own
    sys.stderr.write('BENCH_DETAILS ' + json.dumps(details) + '\n')
    sys.stderr.flush()
    real_stdout.write(json.dumps(line) + '\n')
    real_stdout.flush()
  real_stdout.close()


def main(argv=None):
  argv = sys.argv[1:] if argv is None else argv
  args = parse_args(argv)
  if args.gpus > 1 and not args.standin:
    # (counted in a child: should torch ever fall back from amdsmi to hipGetDeviceCount, the
    # HIP runtime comes up in that child and not in this launcher process)
    child = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = child.stdout.strip().splitlines()
    have = int(lines[-1]) if child.returncode == 0 and lines and lines[-1].strip().isdigit() else 0
    if child.returncode != 0 or not lines:
      sys.stderr.write('bench.py: could not count HIP devices (rc {}): {}\n'.format(
          child.returncode, child.stderr.strip()[-400:]))
    if have < args.gpus:
      sys.stderr.write('bench.py: --gpus {} asked for, but this node shows {} HIP device(s); '
                       'nothing was launched\n'.format(args.gpus, have))
      return 2
  if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.force_dist):
    # Not under a launcher yet: become one.  No GPU call has happened in this process.
    return launch_ranks(args.gpus, argv)
  run_rank(args)
  return 0


if __name__ == '__main__':
  sys.exit(main())
