"""The C ABI's host-only entry points under AddressSanitizer + UndefinedBehaviorSanitizer, fuzzed.

VERDICT r5 items 8-9: `campx_spec_validate` looked at `CampxSpec.table` only for `table_only`
games (a C caller's hand-filled table on a rule game went through to the kernels, which index
the board with it), and the host-only native code - the validators, `campx_shape_tables_build`,
the launch arithmetic behind `campx_flow_shared` / `campx_update_render_shared` /
`campx_pair_table_bytes` - only ever ran inside the hipcc-built library, never under a sanitizer.

Here: `campx_amd.build.build_sanitized()` compiles the product's OWN sources host-side only
(`hipcc --offload-host-only -fsanitize=address,undefined`: no stand-in headers, no stubs; the
kernels are simply not compiled) into build/sanitize/libcampx_hip_san.so, and a child python with
clang's ASan runtime preloaded (tests/abi_fuzz_worker.py: numpy + ctypes only) drives 100 000
byte-wise mutants of valid spec blobs - every game of the one-cell, shape and state-table tiers -
through those entry points.  Every mutant is rejected, or accepted with every index field in
range (re-checked in Python from the header's words, not from the C code) and its tables built
under the sanitizers' eyes.  No GPU, no compute call ("returns 0 or a negative
error code (no exceptions across the ABI)").
"""
import ctypes
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from campx_amd import _hip, build, gamespec, tabulate
from campx_amd.games import boat_race, wall_world
from conftest import REPO
from games_under_test import FUSED_GAMES, SHAPE_GAMES, WIDE_GAMES

HERE = os.path.dirname(os.path.abspath(__file__))
MUTANTS = 100000


def _blob(spec):
  return np.frombuffer(gamespec.spec_bytes(spec), np.uint8).copy()


def _cases():
  specs = []
  for name in sorted(FUSED_GAMES):
    specs.append(_blob(gamespec.lower(gamespec.describe(FUSED_GAMES[name]()))))
  # host-tabulated games: `table_only`, the (cell, action) table filled
  traced = {name: tabulate.trace(build_()) for name, build_ in (('boat_race', boat_race.build),
                                                                 ('wall_world', wall_world.build))}
  for game in traced.values():
    specs.append(_blob(tabulate.to_spec(game)))
  # ... and what round 5's validator let through unchecked: a RULE game whose caller says
  # `table_valid = 1` and brings the table (here a right one - the mutants make it wrong)
  rule = gamespec.lower(gamespec.describe(boat_race.build()))
  filled = tabulate.to_spec(traced['boat_race'])
  ctypes.memmove(ctypes.addressof(rule) + gamespec.CampxSpec.table.offset,
                 ctypes.addressof(filled) + gamespec.CampxSpec.table.offset, gamespec.CampxSpec.table.size)
  rule.table_valid = 1
  specs.append(_blob(rule))
  shapes = [_blob(gamespec.lower_shapes(gamespec.describe(SHAPE_GAMES[name]()))) for name in sorted(SHAPE_GAMES)]
  wides = []
  import random_pickups
  seasons = random_pickups.builder(random_pickups.definitions()[13])  # a scenery of three variants
  coins = random_pickups.builder(random_pickups.definitions()[3])     # ... and one of seven pieces
  for build_ in [WIDE_GAMES[name] for name in sorted(WIDE_GAMES)] + [seasons, coins]:
    spec, arrays = tabulate.to_wide_spec(tabulate.trace(build_()))
    wides.append((_blob(spec), {k: (None if k == 'perf' and not spec.has_perf else np.array(v))
                                for k, v in arrays.items()}))
  # host-tabulated games of two and three movers: the arrays campx_pair_table_pack() takes
  packs = []
  for level in (0, 1):
    game = tabulate.trace(FUSED_GAMES['sokoban' if level == 0 else 'sokoban_l1']())
    packs.append(dict(spec=_blob(tabulate.to_spec(game)), cells=game.rows * game.cols,
                      trace=np.ascontiguousarray(game.trace_bytes()),
                      reward=np.ascontiguousarray(game.reward, dtype=np.float32),
                      done=np.ascontiguousarray(game.done_bytes(), dtype=np.uint8),
                      perf=np.ascontiguousarray(game.perf, dtype=np.int8) if game.has_perf else None))
  return dict(packs=packs, spec_dtype=np.dtype(gamespec.CampxSpec), specs=specs,
              shape_dtype=np.dtype(gamespec.CampxShapeSpec), shapes=shapes,
              wide_dtype=np.dtype(gamespec.CampxWideSpec), wides=wides)


def test_a_hand_filled_table_on_a_rule_game_is_validated():
  """(the shipped library, no sanitizer: the one-line version of what the fuzz below finds)"""
  rule = gamespec.lower(gamespec.describe(boat_race.build()))
  filled = tabulate.to_spec(tabulate.trace(boat_race.build()))
  ctypes.memmove(ctypes.addressof(rule) + gamespec.CampxSpec.table.offset,
                 ctypes.addressof(filled) + gamespec.CampxSpec.table.offset, gamespec.CampxSpec.table.size)
  rule.table_valid = 1
  assert _hip.lib.campx_spec_validate(ctypes.byref(rule)) == 0
  rule.table[7].next_cell = 25                    # a 5 x 5 board has cells 0..24
  assert _hip.lib.campx_spec_validate(ctypes.byref(rule)) == -2
  rule.table[7].next_cell = 24
  rule.table[7].paint = 7                         # seven characters: layers 0..6
  assert _hip.lib.campx_spec_validate(ctypes.byref(rule)) == -2
  rule.table[7].paint = 0x80 | 6                  # (bit 7: "the scenery hides it" - layer 6 is fine)
  assert _hip.lib.campx_spec_validate(ctypes.byref(rule)) == 0
  rule.table_valid = 0                            # a table nobody vouches for is not looked at
  rule.table[7].next_cell = 200
  assert _hip.lib.campx_spec_validate(ctypes.byref(rule)) == 0


def test_host_only_entry_points_survive_100000_mutants_under_asan_and_ubsan(tmp_path):
  lib = build.build_sanitized()
  runtime = build.sanitizer_runtime()
  cases = str(tmp_path / 'cases.pkl')
  with open(cases, 'wb') as f:
    pickle.dump(_cases(), f)
  env = dict(os.environ, LD_PRELOAD=runtime,
             ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=86:allocator_may_return_null=1',
             UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
  env.pop('CAMPX_CONFIG', None)
  run = subprocess.run([sys.executable, os.path.join(HERE, 'abi_fuzz_worker.py'), lib, cases, str(MUTANTS), '6'],
                       env=env, capture_output=True, text=True, timeout=1500)
  assert 'AddressSanitizer' not in run.stderr and 'runtime error' not in run.stderr, run.stderr[-6000:]
  assert run.returncode == 0, (run.returncode, run.stdout[-2000:], run.stderr[-6000:])
  got = json.loads(run.stdout.strip().splitlines()[-1])
  assert got['spec_mutants'] + got['shape_mutants'] + got['wide_mutants'] == MUTANTS
  # the mutants are a real mix: many die in the validators, many live (a flipped scenery byte is
  # a valid game) - a fuzz in which everything is refused, or nothing, tests little
  for family in ('spec', 'shape', 'wide'):
    n, ok = got[family + '_mutants'], got[family + '_accepted']
    assert 0.05 * n < ok < 0.95 * n, got
  assert got['shape_tables_built'] > 1000, got
  assert 0 < got['pack_refused'] < got['pack_mutants'], got       # (cells off the board: refused)


def test_the_sanitized_library_is_not_the_one_that_ships():
  """It lives under build/ (git-ignored, gpurun-ignored as a build by-product), exports the same C
  ABI, and nothing in the package loads it."""
  lib = build.build_sanitized()
  assert os.path.dirname(lib) == os.path.join(REPO, 'build', 'sanitize')
  assert os.path.realpath(_hip._LIB_PATH) != os.path.realpath(lib)
  for root, _, files in os.walk(os.path.join(REPO, 'campx_amd')):
    for name in files:
      if name.endswith('.py') and name != 'build.py':
        assert 'libcampx_hip_san' not in open(os.path.join(root, name)).read(), name
