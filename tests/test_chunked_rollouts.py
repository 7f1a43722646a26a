"""Long rollouts run as chunks of frames (update pass and render alternating) so that the
trace the render kernel reads stays cached; the library cuts from a 28 MB trace up, 16 MB per
chunk.  Here the same code at test sizes: the settings trace_whole_mb=0 / trace_chunk_mb=0 make
every rollout longer than 16 frames run in 16-frame chunks.  A whole test FILE is run again under
them, so they are given the way an embedding process would give them for its lifetime: the
library's one environment variable, CAMPX_CONFIG (include/campx_hip.h)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rollouts_in_16_frame_chunks_match_the_goldens_and_the_oracle():
  env = dict(os.environ, CAMPX_CONFIG='trace_whole_mb=0,trace_chunk_mb=0')
  run = subprocess.run(
      [sys.executable, '-m', 'pytest', os.path.join(HERE, 'test_fused_parity.py'), '-m', 'gpu', '-q',
       '-x', '-k', 'golden_traj or random_streams or sixteen_bit_obs or reset_first or keep_obs',
       '-p', 'no:cacheprovider'],
      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200, cwd=HERE)
  assert run.returncode == 0, run.stdout[-3000:]
  assert ' passed' in run.stdout and 'failed' not in run.stdout


def test_state_table_rollouts_in_16_frame_chunks_match_too():
  """The state-table tier cuts long launches the same way (campx_wide_rollout_launch: the trace of
  all planes at most 2 x trace_chunk_mb per chunk): its parity file and the pickups family's GPU leg
  (pieces in a mask, variants of the scenery, odd batches, 16-bit observations, 60- and 120-frame
  rollouts) in 16-frame chunks."""
  env = dict(os.environ, CAMPX_CONFIG='trace_whole_mb=0,trace_chunk_mb=0')
  for target, pick in ((os.path.join(HERE, 'test_wide_parity.py'), None),
                       (os.path.join(HERE, 'test_random_pickups.py'),
                        'hip_path and (pickup3 or pickup5 or pickup10 or pickup13 or pickup15)'),
                       (os.path.join(HERE, 'test_api_sequences.py'), 'pickup or maze or wide')):
    cmd = [sys.executable, '-m', 'pytest', target, '-m', 'gpu', '-q', '-x', '-p', 'no:cacheprovider']
    if pick:
      cmd += ['-k', pick]
    run = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=1500, cwd=HERE)
    assert run.returncode == 0, run.stdout[-3000:]
    assert ' passed' in run.stdout and 'failed' not in run.stdout, run.stdout[-1000:]
