"""ctypes wrapper of oracle/_build/libcampx_oracle.so (see campx_oracle.c).

TEST INFRASTRUCTURE.  `OracleGame.from_description()` fills the C struct from a
`campx_amd.gamespec.GameDescription` - the plain reading of an Engine's set-up
(characters, masks, z-order, schedule, rule parameters).  Everything after that
(stepping, blocking, rewards, rendering) is the C restatement and shares no code
with the HIP path.
"""

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'libcampx_oracle.so')

MAX_CELLS, MAX_ENTITIES, MAX_CHARS, MAX_SET = 1024, 16, 32, 8
KINDS = {'fixed': 0, 'agent': 1, 'dir_hover': 2, 'box': 3, 'goal': 4,
         'rolling': 5, 'sliding_sprite': 6, 'translate': 7}


class _Entity(ctypes.Structure):
  _fields_ = [('kind', ctypes.c_int32), ('ch', ctypes.c_int32),
              ('group', ctypes.c_int32),
              ('n_blocking', ctypes.c_int32),
              ('blocking', ctypes.c_int32 * MAX_SET),
              ('n_reward_chars', ctypes.c_int32),
              ('reward_chars', ctypes.c_int32 * MAX_SET),
              ('has_step_reward', ctypes.c_int32),
              ('step_reward', ctypes.c_float),
              ('n_agents', ctypes.c_int32),
              ('agents', ctypes.c_int32 * MAX_SET),
              ('base_reward', ctypes.c_float),
              ('dctns', ctypes.c_float * 5),
              ('goal_reward', ctypes.c_float),
              ('is_sprite', ctypes.c_int32), ('visible', ctypes.c_int32),
              ('n_moves', ctypes.c_int32),
              ('roll_axis', ctypes.c_int32 * 4), ('roll_shift', ctypes.c_int32 * 4),
              ('dy', ctypes.c_int32 * 4), ('dx', ctypes.c_int32 * 4),
              ('quit_action', ctypes.c_int32),
              ('t_dy', ctypes.c_int32 * 5), ('t_dx', ctypes.c_int32 * 5),
              ('t_reward', ctypes.c_float * 5),
              ('t_has_reward', ctypes.c_int32), ('t_ends', ctypes.c_int32)]


class _Game(ctypes.Structure):
  _fields_ = [('rows', ctypes.c_int32), ('cols', ctypes.c_int32),
              ('n_entities', ctypes.c_int32), ('n_chars', ctypes.c_int32),
              ('chars', ctypes.c_int32 * MAX_CHARS),
              ('z_order', ctypes.c_int32 * MAX_ENTITIES),
              ('entities', _Entity * MAX_ENTITIES),
              ('backdrop', ctypes.c_uint8 * MAX_CELLS),
              ('curtains0', (ctypes.c_uint8 * MAX_CELLS) * MAX_ENTITIES),
              ('n_perf_masks', ctypes.c_int32), ('perf_char', ctypes.c_int32),
              ('perf_masks', (ctypes.c_uint8 * MAX_CELLS) * MAX_SET),
              ('n_pen_chars', ctypes.c_int32), ('pen_chars', ctypes.c_int32 * MAX_SET),
              ('pen_unit', ctypes.c_int32)]


def build(force=False):
  src = os.path.join(HERE, 'campx_oracle.c')
  if (force or not os.path.exists(LIB_PATH)
      or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)):
    subprocess.run(['make', '-C', HERE, '-B'], check=True,
                   stdout=subprocess.DEVNULL)
  return LIB_PATH


_lib = None


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    _lib.campx_oracle_rollout.restype = i32
    _lib.campx_oracle_rollout.argtypes = [
        ctypes.POINTER(_Game), i64, i32, vp, vp, vp, i32, vp, i64, vp, i64, vp,
        vp, vp, vp]
    _lib.campx_oracle_first_frame.restype = i32
    _lib.campx_oracle_first_frame.argtypes = [ctypes.POINTER(_Game), vp, vp]
    _lib.campx_oracle_sizeof_game.restype = i32
    _lib.campx_oracle_set_threads.restype = i32
    _lib.campx_oracle_set_threads.argtypes = [i32]
    assert _lib.campx_oracle_sizeof_game() == ctypes.sizeof(_Game)
  return _lib


def set_threads(n):
  """Set the OpenMP thread count of the rollout loop; returns the count in use."""
  return int(lib().campx_oracle_set_threads(int(n)))


def _np_ptr(a):
  return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


class OracleGame(object):
  """A game for the CPU oracle plus the batch state (drape curtains, done)."""

  def __init__(self, game, chars, rows, cols, n_entities):
    self._g = game
    self.chars = chars
    self.rows, self.cols, self.n_entities = rows, cols, n_entities
    self.curtains = None
    self.backdrops = None
    self.done = None

  @classmethod
  def from_description(cls, desc):
    g = _Game()
    g.rows, g.cols = desc.rows, desc.cols
    n = desc.rows * desc.cols
    assert n <= MAX_CELLS and len(desc.entities) <= MAX_ENTITIES
    g.n_entities, g.n_chars = len(desc.entities), len(desc.chars)
    for i, ch in enumerate(desc.chars):
      g.chars[i] = ord(ch)
    index = {e.char: i for i, e in enumerate(desc.entities)}
    for i, ch in enumerate(desc.z_order):
      g.z_order[i] = index[ch]
    for i in range(n):
      g.backdrop[i] = int(desc.backdrop.flat[i])
    for i, e in enumerate(desc.entities):
      en, p = g.entities[i], e.params
      kind = e.kind
      if kind == 'shape' and 'quit_actions' in p:
        # a user-written translating thing, as campx_amd.recognise describes it: offsets on
        # both axes, a reward per action, any set of ending actions (KIND_TRANSLATE)
        kind = 'translate'
      elif kind == 'shape':    # Hello World rules: a rolling drape or a sliding sprite
        kind = 'sliding_sprite' if p['sprite'] else 'rolling'
      en.kind, en.ch, en.group = KINDS[kind], ord(e.char), e.group
      en.visible, en.quit_action = 1, -1
      for j in range(n):
        g.curtains0[i][j] = int(e.mask.flat[j])
      if e.kind == 'agent':
        en.n_blocking = len(p['blocking'])
        for j, c in enumerate(p['blocking']):
          en.blocking[j] = ord(c)
        en.n_reward_chars = len(p['reward_chars'])
        for j, c in enumerate(p['reward_chars']):
          en.reward_chars[j] = ord(c)
        en.has_step_reward = int(p['step_reward'] is not None)
        en.step_reward = float(p['step_reward'] or 0.0)
      elif e.kind == 'dir_hover':
        en.n_agents = len(p['agents'])
        for j, c in enumerate(p['agents']):
          en.agents[j] = ord(c)
        en.base_reward = float(p['base_reward'])
        for j in range(5):
          en.dctns[j] = float(p['dctns'][j])
      elif e.kind == 'box':
        en.n_agents, en.agents[0] = 1, ord(p['agent'])
        en.n_blocking = len(p['blocking'])
        for j, c in enumerate(p['blocking']):
          en.blocking[j] = ord(c)
      elif e.kind == 'goal':
        en.n_agents, en.agents[0] = 1, ord(p['agent'])
        en.step_reward = float(p['step_reward'])
        en.goal_reward = float(p['goal_reward'])
      elif kind == 'translate':
        en.visible, en.is_sprite = int(p['visible']), int(p['sprite'])
        for a in range(5):
          en.t_dy[a], en.t_dx[a] = int(p['drow'][a]), int(p['dcol'][a])
          if p['rewards'][a] is not None:
            en.t_has_reward |= 1 << a
            en.t_reward[a] = float(p['rewards'][a])
        for a in p['quit_actions']:
          en.t_ends |= 1 << int(a)
      elif e.kind == 'shape':
        en.n_moves = len(p['drow'])
        assert en.n_moves <= 4
        en.visible = int(p['visible'])
        if p['sprite']:
          en.is_sprite = 1
          for a in range(en.n_moves):
            en.dy[a], en.dx[a] = int(p['drow'][a]), int(p['dcol'][a])
        else:
          for a in range(en.n_moves):
            assert (p['drow'][a] == 0) != (p['dcol'][a] == 0)    # one axis per action
            en.roll_axis[a] = 0 if p['drow'][a] else 1
            en.roll_shift[a] = int(p['drow'][a] or p['dcol'][a])
          rewards = set(p['rewards'])
          assert len(rewards) == 1 and None not in rewards
          en.has_step_reward, en.step_reward = 1, float(p['rewards'][0])
          en.quit_action = -1 if p['quit_action'] is None else int(p['quit_action'])
    if desc.performance is not None:
      agent, masks = desc.performance
      assert len(masks) <= MAX_SET
      g.n_perf_masks, g.perf_char = len(masks), ord(agent)
      for k, m in enumerate(masks):
        for j in range(n):
          g.perf_masks[k][j] = int(m.flat[j])
    if getattr(desc, 'penalty', None) is not None:
      who, masks, unit = desc.penalty
      assert desc.performance is None and len(masks) <= MAX_SET and len(who) <= MAX_SET
      g.n_perf_masks, g.n_pen_chars, g.pen_unit = len(masks), len(who), int(unit)
      for c, ch in enumerate(who):
        g.pen_chars[c] = ord(ch)
      for k, m in enumerate(masks):
        for j in range(n):
          g.perf_masks[k][j] = int(m.flat[j])
    return cls(g, list(desc.chars), desc.rows, desc.cols, len(desc.entities))

  def first_frame(self):
    L, H, W = len(self.chars), self.rows, self.cols
    obs = np.zeros((L, H, W), np.int8)
    board = np.zeros((H, W), np.int8)
    lib().campx_oracle_first_frame(ctypes.byref(self._g), _np_ptr(obs),
                                   _np_ptr(board))
    return obs, board

  def rollout(self, actions, reset_first=False, keep_obs=True, want_board=True):
    """actions int8 [T, B] -> dict(obs, board, reward, discount, done).

    With keep_obs the outputs are [T, B, ...]; otherwise only the last frame
    [B, ...] survives (every frame is still rendered).  State (curtains, done)
    persists across calls on this object; the first call starts from the art.
    """
    actions = np.ascontiguousarray(actions, dtype=np.int8)
    T, B = actions.shape
    L, H, W = len(self.chars), self.rows, self.cols
    if self.curtains is None or self.curtains.shape[0] != B:
      self.curtains = np.zeros((B, self.n_entities, H * W), np.uint8)
      self.backdrops = np.zeros((B, H * W), np.uint8)
      self.done = np.zeros((B,), np.uint8)
      reset_first = True
    Tk = T if keep_obs else 1
    obs = np.zeros((Tk, B, L, H, W), np.int8)
    board = np.zeros((Tk, B, H, W), np.int8) if want_board else None
    reward = np.zeros((T, B), np.float32)
    discount = np.zeros((T, B), np.float32)
    done = np.zeros((T, B), np.uint8)
    perf = np.zeros((T, B), np.int8) if self._g.n_perf_masks else None
    rc = lib().campx_oracle_rollout(
        ctypes.byref(self._g), B, T, _np_ptr(actions), _np_ptr(self.curtains),
        _np_ptr(self.done), int(reset_first), _np_ptr(obs),
        B * L * H * W if keep_obs else 0, _np_ptr(board),
        B * H * W if keep_obs else 0, _np_ptr(reward), _np_ptr(discount),
        _np_ptr(done), _np_ptr(perf), _np_ptr(self.backdrops))
    if rc != 0:
      raise ValueError('oracle: action id outside 0..4 (or a game without any drape)')
    if not keep_obs:
      obs, board = obs[0], (board[0] if want_board else None)
    return dict(obs=obs, board=board, reward=reward, discount=discount,
                done=done, perf=perf)
