"""Compile the native code in-tree (gfx950 only):

    campx_amd/csrc/libcampx_hip.so    kernels + the C ABI (include/campx_hip.h)
    campx_amd/csrc/libcampx_torch.so  the torch custom ops campx::step / rollout / ...
                                      (csrc/campx_torch.cpp), linked against the former

    python -m campx_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so files are git-ignored but travel with
the working tree to the GPU box.
"""

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
# One translation unit per kernel family + the C ABI / dispatch; an A/B variant of one
# kernel rebuilds one object (tools/build_variants.py).
UNITS = ('campx_api', 'k_interp', 'k_rollout_table', 'k_step', 'k_update', 'k_render',
         'k_shape', 'k_wide', 'k_misc')
SRCS = [os.path.join(CSRC, u + '.hip') for u in UNITS]
HEADERS = [os.path.join(CSRC, 'campx_common.hip.h'),
           os.path.join(REPO, 'include', 'campx_hip.h')]
OBJ_DIR = os.path.join(REPO, 'build', 'obj')
OUT = os.path.join(CSRC, 'libcampx_hip.so')
INCLUDE = os.path.join(REPO, 'include')

HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
               '-ffp-contract=off', '-Wall', '-Wno-unused-function',
               '-Wno-pass-failed']


def find_hipcc():
  for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found (looked on PATH and in /opt/rocm/bin)')


def _stale(target, sources):
  return (not os.path.exists(target) or
          os.path.getmtime(target) < max(os.path.getmtime(p) for p in sources))


def needs_build():
  return _stale(OUT, SRCS + HEADERS)


def compile_units(obj_dir=OBJ_DIR, defines=(), force=False, verbose=False, jobs=4):
  """hipcc -c every stale translation unit (in parallel); returns the object paths."""
  os.makedirs(obj_dir, exist_ok=True)
  hipcc = find_hipcc()
  objs, todo = [], []
  for unit, src in zip(UNITS, SRCS):
    obj = os.path.join(obj_dir, unit + '.o')
    objs.append(obj)
    if force or _stale(obj, [src] + HEADERS):
      todo.append([hipcc] + HIPCC_FLAGS + list(defines) +
                  ['-I', INCLUDE, '-I', CSRC, '-c', src, '-o', obj])

  def run(cmd):
    if verbose:
      print(' '.join(cmd))
    subprocess.run(cmd, check=True)

  with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
    list(pool.map(run, todo))
  return objs


def link(objs, out, verbose=False):
  cmd = [find_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out + '.tmp']
  if verbose:
    print(' '.join(cmd))
  subprocess.run(cmd, check=True)
  os.replace(out + '.tmp', out)
  return out


def build_hip(force=False, verbose=False):
  """Build libcampx_hip.so if it is missing or older than its sources."""
  if not force and not needs_build():
    return OUT
  return link(compile_units(force=force, verbose=verbose), OUT, verbose)


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


# ---- the host side under sanitizers (tests/test_c_abi_sanitized.py; CPU only)
SAN_DIR = os.path.join(REPO, 'build', 'sanitize')
SAN_OUT = os.path.join(SAN_DIR, 'libcampx_hip_san.so')
SAN_FLAGS = ['--offload-host-only', '-O1', '-g', '-std=c++17', '-fPIC', '-ffp-contract=off',
             '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
             '-fno-omit-frame-pointer', '-shared-libsan', '-Wno-unused-function', '-Wno-pass-failed']


def sanitizer_runtime():
  """Path of clang's shared ASan runtime (to LD_PRELOAD into a python that loads SAN_OUT)."""
  hipcc = find_hipcc()
  out = subprocess.run([hipcc, '-print-file-name=libclang_rt.asan-x86_64.so'],
                       stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
  if not os.path.isabs(out) or not os.path.exists(out):
    raise RuntimeError('clang\'s libclang_rt.asan-x86_64.so not found (hipcc says {!r})'.format(out))
  return out


def build_sanitized(force=False, verbose=False):
  """libcampx_hip.so's HOST side only (`--offload-host-only`: validators, table builders, launch
  arithmetic; the kernels are not compiled, a launch would fail) with AddressSanitizer and
  UndefinedBehaviorSanitizer - the product's own sources, no stand-ins.  Not shipped, not loaded
  by the package: `build/sanitize/libcampx_hip_san.so`, for the fuzz test of the C ABI's
  host-only entry points.  (GPU sanitizers are not available on this pool.)"""
  if not force and not _stale(SAN_OUT, SRCS + HEADERS):
    return SAN_OUT
  os.makedirs(SAN_DIR, exist_ok=True)
  hipcc = find_hipcc()
  objs, todo = [], []
  for unit, src in zip(UNITS, SRCS):
    obj = os.path.join(SAN_DIR, unit + '.o')
    objs.append(obj)
    todo.append([hipcc] + SAN_FLAGS + ['-I', INCLUDE, '-I', CSRC, '-c', src, '-o', obj])

  def run(cmd):
    if verbose:
      print(' '.join(cmd))
    subprocess.run(cmd, check=True)

  with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
    list(pool.map(run, todo))
  # Every host object registers "its" device code at load time and names it by a symbol the
  # device half of the compilation would have defined (__hip_fatbin_<hash>).  There is no device
  # half here: each of those symbols becomes an EMPTY code-object bundle (the bundle's magic, zero
  # entries) - the kernels do not exist in this library, and nothing in its test launches one.
  wanted = set()
  for obj in objs:
    names = subprocess.run(['nm', '-u', obj], stdout=subprocess.PIPE, text=True, check=True).stdout
    wanted.update(line.split()[-1] for line in names.splitlines() if '__hip_fatbin_' in line)
  stub = os.path.join(SAN_DIR, 'no_device_code.c')
  with open(stub, 'w') as f:
    f.write('/* generated by campx_amd/build.py build_sanitized(): empty code-object bundles */\n')
    for name in sorted(wanted):
      f.write('__attribute__((section(".hip_fatbin"), aligned(4096))) const unsigned char {}[32] = '
              '"__CLANG_OFFLOAD_BUNDLE__";\n'.format(name))
  stub_obj = os.path.join(SAN_DIR, 'no_device_code.o')
  run(['gcc', '-c', '-fPIC', stub, '-o', stub_obj])
  cmd = [hipcc, '--offload-host-only', '-shared', '-fPIC', '-fsanitize=address,undefined',
         '-shared-libsan'] + objs + [stub_obj, '-o', SAN_OUT + '.tmp']
  if verbose:
    print(' '.join(cmd))
  subprocess.run(cmd, check=True)
  os.replace(SAN_OUT + '.tmp', SAN_OUT)
  return SAN_OUT


def build_all(force=False, verbose=False):
  return build_hip(force, verbose), build_torch_ops(force, verbose)


if __name__ == '__main__':
  if '--sanitize' in sys.argv:
    print(build_sanitized(force='--force' in sys.argv, verbose=True))
  else:
    print(build_all(force='--force' in sys.argv, verbose=True))
