"""Compile the HIP library in-tree: campx_amd/csrc/libcampx_hip.so (gfx950 only).

    python -m campx_amd.build

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
working tree to the GPU box.
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc', 'campx_hip.hip')
OUT = os.path.join(HERE, 'csrc', 'libcampx_hip.so')
INCLUDE = os.path.join(REPO, 'include')

HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
               '-ffp-contract=off', '-Wall', '-Wno-unused-function',
               '-Wno-pass-failed']


def find_hipcc():
  for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found (looked on PATH and in /opt/rocm/bin)')


def needs_build():
  if not os.path.exists(OUT):
    return True
  newest = max(os.path.getmtime(SRC),
               os.path.getmtime(os.path.join(INCLUDE, 'campx_hip.h')))
  return os.path.getmtime(OUT) < newest


def build_hip(force=False, verbose=False):
  """Build libcampx_hip.so if it is missing or older than its sources."""
  if not force and not needs_build():
    return OUT
  cmd = [find_hipcc()] + HIPCC_FLAGS + ['-I', INCLUDE, SRC, '-o', OUT + '.tmp']
  if verbose:
    print(' '.join(cmd))
  subprocess.run(cmd, check=True)
  os.replace(OUT + '.tmp', OUT)
  return OUT


if __name__ == '__main__':
  print(build_hip(force='--force' in sys.argv, verbose=True))
