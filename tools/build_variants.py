#!/usr/bin/env python3
"""Build A/B variants of libcampx_hip.so (different -D knobs) under build/variants/<name>/.

    python tools/build_variants.py name:-DCAMPX_UPD_PROD=4,-DCAMPX_UPD_CONS=4 ...

Each directory also gets a copy of libcampx_torch.so (its rpath is $ORIGIN, so it binds
to the variant next to it).  Run one with  CAMPX_LIB=build/variants/<name>/libcampx_hip.so.
"""
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from campx_amd import build  # noqa: E402


def main(specs):
  build.build_all()
  procs = []
  for spec in specs:
    name, _, defs = spec.partition(':')
    d = os.path.join(REPO, 'build', 'variants', name)
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, 'libcampx_hip.so')
    cmd = [build.find_hipcc()] + build.HIPCC_FLAGS + [x for x in defs.split(',') if x] + [
        '-I', build.INCLUDE, build.SRC, '-o', out]
    procs.append((name, subprocess.Popen(cmd)))
    shutil.copy(build.TORCH_OUT, os.path.join(d, 'libcampx_torch.so'))
    if len(procs) % 4 == 0:
      for _, p in procs[-4:]:
        p.wait()
  for name, p in procs:
    assert p.wait() == 0, name
    print('built', name)


if __name__ == '__main__':
  main(sys.argv[1:])
