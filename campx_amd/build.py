"""Compile the native code in-tree (gfx950 only):

    campx_amd/csrc/libcampx_hip.so    kernels + the C ABI (include/campx_hip.h)
    campx_amd/csrc/libcampx_torch.so  the torch custom ops campx::step / rollout / ...
                                      (csrc/campx_torch.cpp), linked against the former

    python -m campx_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so files are git-ignored but travel with
the working tree to the GPU box.
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


TORCH_SRC = os.path.join(HERE, 'csrc', 'campx_torch.cpp')
TORCH_OUT = os.path.join(HERE, 'csrc', 'libcampx_torch.so')


def build_torch_ops(force=False, verbose=False):
  """Build libcampx_torch.so (TORCH_LIBRARY registration of the campx:: ops)."""
  build_hip(force=False, verbose=verbose)
  if (not force and os.path.exists(TORCH_OUT) and os.path.getmtime(TORCH_OUT) >= max(
      os.path.getmtime(TORCH_SRC), os.path.getmtime(OUT),
      os.path.getmtime(os.path.join(INCLUDE, 'campx_hip.h')))):
    return TORCH_OUT
  import torch
  tdir = os.path.dirname(os.path.abspath(torch.__file__))
  cmd = [find_hipcc(), '-O2', '-std=c++17', '-shared', '-fPIC', '-DUSE_ROCM',
         '-Wno-unused-result',
         '-D_GLIBCXX_USE_CXX11_ABI={}'.format(int(torch._C._GLIBCXX_USE_CXX11_ABI)),
         '-I', INCLUDE, '-I', os.path.join(tdir, 'include'),
         '-I', os.path.join(tdir, 'include', 'torch', 'csrc', 'api', 'include'),
         TORCH_SRC, '-L', os.path.join(tdir, 'lib'), '-lc10', '-ltorch_cpu', '-ltorch',
         '-lc10_hip', '-ltorch_hip', '-L', os.path.join(HERE, 'csrc'), '-lcampx_hip',
         '-Wl,-rpath,$ORIGIN', '-o', TORCH_OUT + '.tmp']
  if verbose:
    print(' '.join(cmd))
  subprocess.run(cmd, check=True)
  os.replace(TORCH_OUT + '.tmp', TORCH_OUT)
  return TORCH_OUT


def build_all(force=False, verbose=False):
  return build_hip(force, verbose), build_torch_ops(force, verbose)


if __name__ == '__main__':
  print(build_all(force='--force' in sys.argv, verbose=True))
