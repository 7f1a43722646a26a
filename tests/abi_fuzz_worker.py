"""Run by tests/test_c_abi_sanitized.py in a child python with clang's AddressSanitizer runtime
preloaded: byte-wise mutants of valid spec blobs through the HOST-ONLY entry points of the C ABI
(include/campx_hip.h), built from the product's own sources with -fsanitize=address,undefined
(campx_amd/build.py build_sanitized).  numpy and ctypes only - no torch, nothing of the package -
so that the preloaded runtime has nothing else to trip over.

    python abi_fuzz_worker.py <libcampx_hip_san.so> <cases.pkl> <mutants> <seed>

Every mutant is either REJECTED by its validator (a negative code) or ACCEPTED - and then every
field a kernel or a table builder uses as an index is re-checked here, in Python, from the
mutant's bytes (the restatement below is independent of the C code: written from the header), and
the host-side table builders are run on it (the sanitizers watch them index).  A sanitizer report
aborts the process (-fno-sanitize-recover, ASan's default): the parent sees a non-zero exit code.
Prints one JSON line: counts per family.
"""
import ctypes
import json
import pickle
import sys

import numpy as np

N_ACTIONS = 5
MAX_LIST = None       # set from the dtype


def _mutate(rng, blob, hot, frozen=()):
  """In place: one to three bytes of `blob` (a uint8 array) - half the time inside `hot` (the
  byte ranges that hold counts and indices) - set to a random value, a boundary value, or the
  old value +-1.  `frozen`: byte ranges left alone (host pointers; counts that size the caller's
  own arrays)."""
  n = int(rng.randint(1, 4))
  for _ in range(n):
    for attempt in range(20):
      if hot and rng.rand() < 0.6:
        lo, hi = hot[int(rng.randint(len(hot)))]
        at = int(rng.randint(lo, hi))
      else:
        at = int(rng.randint(blob.size))
      if not any(lo <= at < hi for lo, hi in frozen):
        break
    else:
      continue
    how = rng.randint(4)
    old = int(blob[at])
    blob[at] = (int(rng.randint(256)) if how == 0 else
                int(rng.choice([0, 1, 0x7f, 0x80, 0xff, 0xfe])) if how == 1 else
                (old + 1) & 0xff if how == 2 else (old - 1) & 0xff)


def _span(dtype, name):
  off = dtype.fields[name][1]
  return off, off + dtype.fields[name][0].itemsize


# ------------------------------------------------------------------ CampxSpec (one-cell tier)

def spec_invariants(s):
  """What an ACCEPTED CampxSpec must satisfy for every index the kernels form from it to be in
  range (include/campx_hip.h CampxSpec: 'layer l', 'row * cols + col', 'bit s = static drape s')."""
  rows, cols, L, K = int(s['rows']), int(s['cols']), int(s['n_layers']), int(s['n_dyn'])
  assert 1 <= rows <= 127 and 1 <= cols <= 127 and rows * cols <= 128
  HW = rows * cols
  assert 1 <= L <= 16 and 1 <= K <= 4 and 0 <= int(s['n_static']) <= 16 and 0 <= int(s['n_rules']) <= 16
  for d in range(K):
    assert 0 <= int(s['dyn_layer'][d]) < L
    assert 0 <= int(s['dyn_row0'][d]) < rows and 0 <= int(s['dyn_col0'][d]) < cols
  assert (s['static_top_layer'][:HW] < L).all()
  if int(s['n_static']) < 16:
    assert not (s['static_cover'][:HW] >> int(s['n_static'])).any()
  tmpl = s['obs_template'][:L * HW]
  assert ((tmpl == 0) | (tmpl == 1)).all()
  for i in range(int(s['n_rules'])):
    r = s['rules'][i]
    assert 0 <= int(r['dyn']) < K and int(r['op']) in (1, 2, 3, 4)
    aux = int(r['aux'])
    assert {1: True, 2: 0 <= aux < L, 3: 0 <= aux < K, 4: 0 <= aux < int(s['n_static'])}[int(r['op'])]
  assert -1 <= int(s['perf_dyn']) < K
  if int(s['table_valid']) and K == 1:
    t = s['table'][:HW * N_ACTIONS]
    assert (t['next_cell'] < HW).all() and ((t['paint'] & 0x7f) < L).all() and not (t['done'] & 0x0e).any()
  if int(s['table_only']):
    assert int(s['table_only']) == 1 and int(s['n_rules']) == 0
    assert K > 1 or int(s['table_valid'])
  if int(s['perf_dyn']) >= 0:
    if int(s['perf_mode']) == 0:
      assert 2 <= int(s['perf_n']) <= 255 and (s['cell_class'][:HW] <= int(s['perf_n'])).all()
    else:
      assert int(s['perf_mode']) == 1 and 0 < int(s['perf_mask']) < (1 << K)
      things = bin(int(s['perf_mask'])).count('1')
      assert int(s['cell_class'][:HW].max()) * things <= 7
  d = s['discount_list'][1:]
  assert ((d >= 0) & (d <= 1)).all()


def fuzz_specs(lib, dtype, blobs, n, rng):
  vp = ctypes.c_void_p
  lib.campx_spec_validate.restype = ctypes.c_int32
  lib.campx_spec_validate.argtypes = [vp]
  lib.campx_pair_table_bytes.restype = ctypes.c_int64
  lib.campx_pair_table_bytes.argtypes = [vp]
  lib.campx_flow_shared.restype = ctypes.c_int32
  lib.campx_flow_shared.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64]
  lib.campx_update_render_shared.restype = ctypes.c_int32
  lib.campx_update_render_shared.argtypes = [vp, ctypes.c_int64, ctypes.c_int32]
  lib.campx_flow_scratch_bytes.restype = ctypes.c_int64
  lib.campx_flow_scratch_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
  # the head of the struct (counts, the movers' layers and starts, the rules), the scenery's
  # per-cell tables, the transition table, the hidden-performance words at the end
  head = (0, _span(dtype, 'rules')[1])
  hot = [head, head, _span(dtype, 'static_top_layer'), _span(dtype, 'static_cover'), _span(dtype, 'table'),
         _span(dtype, 'cell_class'), (_span(dtype, 'perf_mode')[0], dtype.itemsize)]
  accepted = 0
  for i in range(n):
    blob = np.array(blobs[i % len(blobs)], dtype=np.uint8, copy=True)
    _mutate(rng, blob, hot)
    ptr = blob.ctypes.data
    rc = lib.campx_spec_validate(ptr)
    pair = lib.campx_pair_table_bytes(ptr)
    shared = lib.campx_flow_shared(ptr, 4096, 100, 4096), lib.campx_update_render_shared(ptr, 4096, 100)
    assert rc in (0, -2), rc
    if rc != 0:
      assert pair == 0 and shared == (0, 0), (rc, pair, shared)       # nothing is planned from a refused spec
      continue
    accepted += 1
    spec_invariants(blob.view(dtype)[0])
    assert pair >= 0
  assert lib.campx_flow_scratch_bytes(4096, 100) == 16 + 2 * 4 * 100 * 4096
  return accepted


# ------------------------------------------------------------------ CampxShapeSpec (shape tier)

def shape_invariants(s, max_list):
  rows, cols, L, N = int(s['rows']), int(s['cols']), int(s['n_layers']), int(s['n_things'])
  assert 1 <= rows <= 127 and 1 <= cols <= 127 and rows * cols <= s['backdrop'].size
  assert 1 <= L <= 16 and 1 <= N <= s['things'].size and 0 <= int(s['first_drape']) < N
  assert sorted(int(x) for x in s['update_order'][:N]) == list(range(N))
  for k in range(N):
    t = s['things'][k]
    assert 0 <= int(t['layer']) < L
    n, begin = int(t['n_cells']), int(t['cell_begin'])
    assert n >= 0 and begin >= 0 and begin + n <= max_list
    cells = s['cells'][begin:begin + n]
    assert ((cells >> 8) < rows).all() and ((cells & 0xff) < cols).all()
    assert ((t['drow'][:N_ACTIONS] >= 0) & (t['drow'][:N_ACTIONS] < rows)).all()
    assert ((t['dcol'][:N_ACTIONS] >= 0) & (t['dcol'][:N_ACTIONS] < cols)).all()
    assert not ((int(t['has_reward_mask']) | int(t['terminate_mask'])) >> N_ACTIONS)
  assert (s['backdrop'][:rows * cols] < L).all()


def fuzz_shapes(lib, dtype, blobs, n, rng):
  vp = ctypes.c_void_p
  lib.campx_shape_spec_validate.restype = ctypes.c_int32
  lib.campx_shape_spec_validate.argtypes = [vp]
  lib.campx_shape_tables_bytes.restype = ctypes.c_int64
  lib.campx_shape_tables_bytes.argtypes = [vp]
  lib.campx_shape_tables_build.restype = ctypes.c_int32
  lib.campx_shape_tables_build.argtypes = [vp, vp, ctypes.c_int64]
  lib.campx_shape_scratch_bytes.restype = ctypes.c_int64
  lib.campx_shape_scratch_bytes.argtypes = [vp, ctypes.c_int64, ctypes.c_int32]
  max_list = dtype.fields['cells'][0].shape[0]
  hot = [(0, _span(dtype, 'things')[1])] * 3 + [_span(dtype, 'backdrop'), _span(dtype, 'cells')]
  accepted = built = 0
  for i in range(n):
    blob = np.array(blobs[i % len(blobs)], dtype=np.uint8, copy=True)
    _mutate(rng, blob, hot)
    ptr = blob.ctypes.data
    rc = lib.campx_shape_spec_validate(ptr)
    need = lib.campx_shape_tables_bytes(ptr)
    scratch = lib.campx_shape_scratch_bytes(ptr, 4096, 100)
    assert rc in (0, -2), rc
    if rc != 0:
      assert need == 0 and scratch == 0
      continue
    accepted += 1
    shape_invariants(blob.view(dtype)[0], max_list)
    assert need >= 0 and scratch >= 0
    if need > 0:
      # (a game the frame-major kernels take: its row-word tables, written into EXACTLY the bytes
      # asked for - one byte more and the sanitizer reports it - and refused in one byte less)
      tables = np.empty(need, np.uint8)
      assert lib.campx_shape_tables_build(ptr, tables.ctypes.data, need) == 0
      assert lib.campx_shape_tables_build(ptr, tables.ctypes.data, need - 1) != 0
      built += 1
  return accepted, built


# ------------------------------------------------------------------ CampxWideSpec (state tables)

def wide_invariants(s, arrays):
  rows, cols, L, K, S = (int(s[k]) for k in ('rows', 'cols', 'n_layers', 'n_dyn', 'n_states'))
  HW = rows * cols
  assert 1 <= rows <= 127 and 1 <= cols <= 127 and 16 <= HW <= s['static_top_layer'].size
  assert 1 <= L <= 16 and 1 <= K <= s['dyn_layer'].size and S >= 1
  V = int(s['n_variants'])
  assert 0 <= V <= 256 and (V <= 1 or K <= s["dyn_layer"].size - 1)
  if V > 1 and 'variant_top_layer' in arrays:
    assert (arrays['variant_top_layer'] < L).all() and (arrays['state_variant'] < V).all()
    assert (arrays['variant_top_layer'][0][:HW] == s['static_top_layer'][:HW]).all()
  P = int(s['n_pieces'])
  assert 0 <= P <= s['piece_cell'].size and (P == 0 or (V <= 1 and K <= s['dyn_layer'].size - 1))
  assert (s['piece_cell'][:P] < HW).all() and (s['piece_layer'][:P] < L).all()
  assert (s['piece_layer'][:P] != s['static_top_layer'][s['piece_cell'][:P]]).all()
  if P and 'state_pieces' in arrays:
    assert not (arrays['state_pieces'] >> P).any()
  assert ((s['dyn_layer'][:K] >= 0) & (s['dyn_layer'][:K] < L)).all()
  assert (s['static_top_layer'][:HW] < L).all()
  cells = arrays['state_cells']
  assert ((cells & 0x3ff) < HW).all() and not (cells & 0x7c00).any()
  nxt = arrays['next_state']
  assert ((nxt >= 0) & (nxt < S)).all()
  assert not (arrays['done'] & 0x0e).any()
  if not int(s['any_dcode']):
    assert not (arrays['done'] >> 4).any()


def fuzz_wide(lib, dtype, cases, n, rng):
  vp = ctypes.c_void_p
  lib.campx_wide_spec_validate.restype = ctypes.c_int32
  lib.campx_wide_spec_validate.argtypes = [vp]
  lib.campx_wide_tables_bytes.restype = ctypes.c_int64
  lib.campx_wide_tables_bytes.argtypes = [vp]
  lib.campx_wide_tables_build.restype = ctypes.c_int32
  lib.campx_wide_tables_build.argtypes = [vp, vp, vp]
  pointers = ('state_cells', 'next_state', 'reward', 'done', 'perf', 'variant_top_layer', 'state_variant',
              'state_pieces')
  # (the host pointers are the harness's; n_states and n_dyn size the caller's own arrays - a
  # caller that lies about them is beyond what a validator can see)
  frozen = [_span(dtype, name) for name in pointers + ('n_states', 'n_dyn')]
  hot = [(0, _span(dtype, 'static_top_layer')[0])]
  accepted = 0
  for i in range(n):
    blob0, arrays0 = cases[i % len(cases)]
    blob = np.array(blob0, dtype=np.uint8, copy=True)
    arrays = {k: np.array(v, copy=True) for k, v in arrays0.items() if v is not None}
    if rng.rand() < 0.5:
      # (a game whose scenery has variants: their number and the board's size also size the
      # caller's [V][rows * cols] array)
      sizes = [_span(dtype, name) for name in ('n_variants', 'rows', 'cols')] if 'variant_top_layer' in arrays else []
      _mutate(rng, blob, hot, frozen + sizes)
    else:                                        # ... or the tables the pointers point at
      names = ['state_cells', 'next_state', 'done'] + [n_ for n_ in ('variant_top_layer', 'state_variant', 'state_pieces')
                                                        if n_ in arrays]
      name = names[int(rng.randint(len(names)))]
      _mutate(rng, arrays[name].view(np.uint8).reshape(-1), [])
    s = blob.view(dtype)[0]
    for name in pointers:
      s[name] = arrays[name].ctypes.data if name in arrays else 0
    ptr = blob.ctypes.data
    rc = lib.campx_wide_spec_validate(ptr)
    need = lib.campx_wide_tables_bytes(ptr)
    assert rc in (0, -2), rc
    if rc != 0:
      continue
    accepted += 1
    wide_invariants(s, arrays)
    assert need > 0
    if i % 8 == 0:
      # the table builder, under the sanitizers: it packs the whole blob on the host and only then
      # copies it to the "device" - here a host buffer and no GPU, so the copy is what fails
      # (CAMPX_ELAUNCH), after every index has been used
      blob_out = np.empty(need, np.uint8)
      rc = lib.campx_wide_tables_build(ptr, blob_out.ctypes.data, None)
      missing = (int(s['n_variants']) > 1 and 'variant_top_layer' not in arrays) or \
          (int(s['has_perf']) and 'perf' not in arrays) or \
          (int(s['n_pieces']) > 0 and 'state_pieces' not in arrays)
      # (a scenery of several variants or of pieces / a hidden performance, and no array given: refused)
      assert (rc == -1) if missing else (rc in (-3, 0)), (rc, missing)
  return accepted


# ------------------------------------------------------------------ campx_pair_table_pack

def fuzz_pack(lib, cases, n, rng):
  """The host arrays of a host-tabulated game of two to four movers (next cell | shows bit per
  mover, reward, done | discount code, perf per (cell, ..., cell, action)) packed into the pair /
  tuple table: a cell outside the board is refused (CAMPX_EINVAL); otherwise the packing runs to
  its end and the copy to a device that is not there fails (CAMPX_ELAUNCH)."""
  vp = ctypes.c_void_p
  lib.campx_pair_table_bytes.restype = ctypes.c_int64
  lib.campx_pair_table_bytes.argtypes = [vp]
  lib.campx_pair_table_pack.restype = ctypes.c_int32
  lib.campx_pair_table_pack.argtypes = [vp] * 7
  refused = 0
  for i in range(n):
    case = cases[i % len(cases)]
    spec = np.array(case['spec'], np.uint8)
    arrays = {k: np.array(case[k], copy=True) for k in ('trace', 'reward', 'done', 'perf') if case[k] is not None}
    name = ('trace', 'trace', 'done', 'reward')[int(rng.randint(4))]
    _mutate(rng, arrays[name].view(np.uint8).reshape(-1), [])
    need = lib.campx_pair_table_bytes(spec.ctypes.data)
    assert need > 0
    table = np.empty(need, np.uint8)
    rc = lib.campx_pair_table_pack(spec.ctypes.data, arrays['trace'].ctypes.data, arrays['reward'].ctypes.data,
                                   arrays['done'].ctypes.data,
                                   arrays['perf'].ctypes.data if 'perf' in arrays else None, table.ctypes.data, None)
    bad_cell = bool(((arrays['trace'] & 0x7f) >= case['cells']).any())
    assert rc == -1 if bad_cell else rc in (-3, -2, 0), (rc, bad_cell)
    refused += rc == -1
  return refused


def main(argv):
  lib = ctypes.CDLL(argv[1])
  with open(argv[2], 'rb') as f:
    cases = pickle.load(f)
  n, seed = int(argv[3]), int(argv[4])
  rng = np.random.RandomState(seed)
  out = {}
  n_spec, n_shape = int(n * 0.6), int(n * 0.25)
  out['spec_mutants'] = n_spec
  out['spec_accepted'] = fuzz_specs(lib, cases['spec_dtype'], cases['specs'], n_spec, rng)
  out['shape_mutants'] = n_shape
  out['shape_accepted'], out['shape_tables_built'] = fuzz_shapes(lib, cases['shape_dtype'], cases['shapes'], n_shape, rng)
  out['wide_mutants'] = n - n_spec - n_shape
  out['wide_accepted'] = fuzz_wide(lib, cases['wide_dtype'], cases['wides'], out['wide_mutants'], rng)
  out['pack_mutants'] = max(200, n // 100)
  out['pack_refused'] = fuzz_pack(lib, cases['packs'], out['pack_mutants'], rng)
  # every valid case is accepted as it is
  for blob in cases['specs']:
    kept = np.array(blob, np.uint8)              # (a name: the buffer must outlive the call)
    assert lib.campx_spec_validate(kept.ctypes.data) == 0
  for blob in cases['shapes']:
    kept = np.array(blob, np.uint8)
    assert lib.campx_shape_spec_validate(kept.ctypes.data) == 0
  print(json.dumps(out))


if __name__ == '__main__':
  main(sys.argv)
