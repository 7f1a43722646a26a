"""The frame-major shape path's HOST side (csrc/k_shape.hip campx_shape_tables_build, pure C) and
its arithmetic, without a GPU: the row-word tables decoded and compared with numpy rolls of the
things' masks, and a numpy restatement of what the two kernels compute - the update pass's
offsets, trail words and keyframes; the render pass's slots (a W-bit row per (environment, layer,
board row), from the things' rotated row words front to back under a running `covered`, the art's
row under the trail words brought from the keyframe up to the frame) - replayed over the REFERENCE
engine's frames of Hello World and two shape-zoo games (`tests/golden/*.npz`): every layer of
every frame, trails, rebuilds after a quit and all.  The kernels themselves are compared with the
oracle and the serial kernel on the GPU (tests/test_shape_parity.py); this pins the scheme and the
tables they read."""

import ctypes

import numpy as np
import pytest

from campx_amd import _hip, gamespec
from games_under_test import SHAPE_GAMES

KEY = 4          # csrc/k_shape.hip kShapeKey


def _tables(spec):
  n = int(_hip.lib.campx_shape_tables_bytes(ctypes.byref(spec)))
  assert n > 0 and n % 8 == 0
  buf = np.zeros(n // 8, np.uint64)
  _hip.check(_hip.lib.campx_shape_tables_build(ctypes.byref(spec), ctypes.c_void_p(buf.ctypes.data), n),
             'campx_shape_tables_build')
  return buf


def _decode(spec, buf):
  """(static rows [L, H], {thing index: row words [W, H]}) from the blob's header."""
  head = buf.view(np.uint32)
  assert head[0] == 0x54485343 and tuple(head[1:6].view(np.int32)) == (
      spec.rows, spec.cols, spec.n_layers, spec.n_things, spec.first_drape)
  static_off, offs, total = int(head[6]), head[7:15], int(head[15])
  assert total == buf.nbytes
  H, W, L = spec.rows, spec.cols, spec.n_layers
  static = buf[static_off // 8: static_off // 8 + L * H].reshape(L, H)
  rows = {k: buf[int(o) // 8: int(o) // 8 + W * H].reshape(W, H) for k, o in enumerate(offs) if o}
  return static, rows


def _mask_of(spec, k):
  th = spec.things[k]
  m = np.zeros((spec.rows, spec.cols), bool)
  for i in range(th.n_cells):
    c = spec.cells[th.cell_begin + i]
    m[c >> 8, c & 0xff] = True
  return m


def _bits(mask_row):
  return sum(1 << c for c in np.flatnonzero(mask_row))


@pytest.mark.parametrize('name', ['hello_world', 'shape_zoo3', 'shape_zoo4'])
def test_row_word_tables_are_the_masks_in_every_column_rotation(name):
  spec = gamespec.lower_shapes(gamespec.describe(SHAPE_GAMES[name]()))
  static, rows = _decode(spec, _tables(spec))
  H, W = spec.rows, spec.cols
  backdrop = np.array(spec.backdrop[:H * W], np.uint8).reshape(H, W)
  for l in range(spec.n_layers):
    for r in range(H):
      assert int(static[l, r]) == _bits(backdrop[r] == l)
  multi = [k for k in range(spec.first_drape, spec.n_things)
           if spec.things[k].visible and spec.things[k].n_cells > 1]
  assert sorted(rows) == multi and multi
  for k in multi:
    mask = _mask_of(spec, k)
    for dc in range(W):
      rolled = np.roll(mask, dc, axis=1)
      for r in range(H):
        assert int(rows[k][dc, r]) == _bits(rolled[r]), (k, dc, r)


def test_games_whose_rows_are_not_16_to_64_cells_have_no_tables():
  for name in ('shape_zoo0', 'shape_zoo1', 'shape_zoo2'):
    spec = gamespec.lower_shapes(gamespec.describe(SHAPE_GAMES[name]()))
    assert _hip.lib.campx_shape_tables_bytes(ctypes.byref(spec)) == 0
    assert _hip.lib.campx_shape_scratch_bytes(ctypes.byref(spec), 64, 10) == 0


def _replay(spec, static, rows, actions):
  """What shape_update_split_kernel + shape_render_split_kernel compute, in numpy: uint8
  [T, N, L, H, W] observations of `actions` [T, N] from a fresh start."""
  H, W, L, N_things, FD = spec.rows, spec.cols, spec.n_layers, spec.n_things, spec.first_drape
  T, N = actions.shape
  things = [spec.things[k] for k in range(N_things)]
  art = [(spec.cells[t.cell_begin] >> 8, spec.cells[t.cell_begin] & 0xff) if t.n_cells else (0, 0) for t in things]
  trail = [k for k in range(FD) if things[k].visible]
  S = len(trail)
  front = [k for k in range(N_things - 1, FD - 1, -1) if things[k].visible and things[k].n_cells > 0]
  # ---- update pass: offsets per frame, trail words, keyframes, rebuild flags
  off_r = np.zeros((T, N, N_things), int)
  off_c = np.zeros((T, N, N_things), int)
  rebuilt = np.zeros((T, N), bool)
  pos = np.zeros((T, N, max(S, 1), 2), int)
  keys = {}
  cur_r, cur_c = np.zeros((N, N_things), int), np.zeros((N, N_things), int)
  words = np.zeros((N, max(S, 1), H), object)
  words[:] = 0
  over = np.zeros(N, bool)
  for t in range(T):
    for e in range(N):
      if over[e]:
        cur_r[e], cur_c[e], over[e], rebuilt[t, e] = 0, 0, False, True
        words[e] = 0
      a = int(actions[t, e])
      if 0 <= a < 5:
        for k, th in enumerate(things):
          cur_r[e, k] = (cur_r[e, k] + th.drow[a]) % H
          cur_c[e, k] = (cur_c[e, k] + th.dcol[a]) % W
          if (th.terminate_mask >> a) & 1:
            over[e] = True
      for s, k in enumerate(trail):              # back to front: mine, nobody else's
        r, c = (art[k][0] + cur_r[e, k]) % H, (art[k][1] + cur_c[e, k]) % W
        for q in range(S):
          words[e, q, r] = (words[e, q, r] | (1 << c)) if q == s else (words[e, q, r] & ~(1 << c))
        pos[t, e, s] = (r, c)
    off_r[t], off_c[t] = cur_r, cur_c
    if S and t % KEY == 0:
      keys[t // KEY] = words.copy()
  # ---- render pass: every slot of every frame
  obs = np.zeros((T, N, L, H, W), np.uint8)
  for t in range(T):
    k0 = t - t % KEY
    for e in range(N):
      for r in range(H):
        tw = [keys[k0 // KEY][e, s, r] for s in range(S)]
        for f in range(k0 + 1, t + 1):          # the frames since the keyframe, replayed
          if rebuilt[f, e]:
            tw = [0] * S
          for s in range(S):
            rr, cc = pos[f, e, s]
            bit = (1 << int(cc)) if rr == r else 0
            tw = [(w | bit) if q == s else (w & ~bit) for q, w in enumerate(tw)]
        covered, vis = 0, {}
        for k in front:                         # front to back
          dr, dc = off_r[t, e, k], off_c[t, e, k]
          if things[k].n_cells == 1:
            w = (1 << ((art[k][1] + dc) % W)) if (art[k][0] + dr) % H == r else 0
          else:
            w = int(rows[k][dc, (r - dr) % H])
          vis[k] = w & ~covered
          covered |= w
        any_trail = 0
        for w in tw:
          any_trail |= w
        for l in range(L):
          row = int(static[l, r]) & ~(covered | any_trail)
          for k in front:
            if things[k].layer == l:
              row |= vis[k]
          for s, k in enumerate(trail):
            if things[k].layer == l:
              row |= tw[s] & ~covered
          for c in range(W):
            obs[t, e, l, r, c] = (row >> c) & 1
  return obs


@pytest.mark.parametrize('name,envs,frames', [('hello_world', 3, 40), ('shape_zoo3', 2, 30), ('shape_zoo4', 2, 30)])
def test_the_row_word_scheme_reproduces_the_reference_engines_frames(name, envs, frames, golden):
  gold = golden(name)
  spec = gamespec.lower_shapes(gamespec.describe(SHAPE_GAMES[name]()))
  static, rows = _decode(spec, _tables(spec))
  actions = gold['actions'][:frames, :envs]
  got = _replay(spec, static, rows, actions)
  want = gold['layered'][1:frames + 1, :envs].astype(np.uint8)
  assert got.shape == want.shape
  assert np.array_equal(got, want)
  # the goldens exercise what the scheme is for: trails (two of the three games), and a quit
  if name != 'shape_zoo4':
    assert sum(spec.things[k].visible for k in range(spec.first_drape)) >= 1
