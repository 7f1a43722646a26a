#!/usr/bin/env python3
"""Build A/B variants of libcampx_hip.so (different -D knobs) under build/variants/<name>/.

    python tools/build_variants.py name:-DCAMPX_UPD_PROD=4,-DCAMPX_UPD_CONS=4 ...

Each variant has its own object directory (build/variants/<name>/obj): the library is one
translation unit per kernel family (campx_amd/build.py UNITS), compiled in parallel.

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
  for spec in specs:
    name, _, defs = spec.partition(':')
    d = os.path.join(REPO, 'build', 'variants', name)
    os.makedirs(d, exist_ok=True)
    objs = build.compile_units(obj_dir=os.path.join(d, 'obj'),
                               defines=[x for x in defs.split(",") if x], jobs=8, force=True)
    build.link(objs, os.path.join(d, 'libcampx_hip.so'))
    shutil.copy(build.TORCH_OUT, os.path.join(d, 'libcampx_torch.so'))
    print('built', name)


if __name__ == '__main__':
  main(sys.argv[1:])
