"""Bindings of the native code: libcampx_hip.so and libcampx_torch.so.

* `lib` is the ctypes binding of the C ABI in include/campx_hip.h.  The fused tier
  uses it at set-up time (spec validation / compilation, error strings); the
  per-frame entry points are bound too, as the stub INTEGRATION.md shows a CampX
  maintainer, and tests call them directly.
* `ops` is `torch.ops.campx` after loading libcampx_torch.so: the custom ops
  `campx::reset / step / rollout / onehot_to_ids / check_actions` that
  `Engine.its_showtime() / play() / rollout()` lower to (csrc/campx_torch.cpp).

Importing this module loads both and fails loudly if either has not been built:
there is no CPU fallback for the fused tier.
"""

import ctypes
import os

from .gamespec import CampxSpec, CampxShapeSpec, CampxWideSpec

_LIB_PATH = os.environ.get('CAMPX_LIB') or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libcampx_hip.so')

EXPORTS = ('campx_spec_size', 'campx_flow_scratch_bytes', 'campx_spec_validate', 'campx_spec_compile',
           'campx_pair_table_bytes', 'campx_pair_table_build', 'campx_pair_table_pack',
           'campx_reset_launch',
           'campx_rollout_launch', 'campx_update_launch', 'campx_render_launch',
           'campx_update_render_launch', 'campx_update_render_shared', 'campx_flow_shared',
           'campx_shape_spec_size', 'campx_shape_spec_validate',
           'campx_shape_rollout_launch', 'campx_shape_tables_bytes', 'campx_shape_tables_build',
           'campx_shape_scratch_bytes',
           'campx_wide_spec_size', 'campx_wide_spec_validate', 'campx_wide_tables_bytes',
           'campx_wide_tables_build', 'campx_wide_reset_launch', 'campx_wide_rollout_launch',
           'campx_wide_rules_size', 'campx_wide_enumerate_launch',
           'campx_check_actions_launch',
           'campx_onehot_to_ids_launch', 'campx_config_set', 'campx_config_get',
           'campx_config_string', 'campx_write_probe_launch', 'campx_strerror',
           'campx_last_hip_error', 'campx_device_arch')


class CampxState(ctypes.Structure):
  _fields_ = [('pos', ctypes.c_void_p), ('done', ctypes.c_void_p),
              ('ret', ctypes.c_void_p), ('pair_table', ctypes.c_void_p)]


class CampxOutputs(ctypes.Structure):
  _fields_ = [('obs', ctypes.c_void_p), ('obs_t_stride', ctypes.c_int64),
              ('board', ctypes.c_void_p), ('board_t_stride', ctypes.c_int64),
              ('reward', ctypes.c_void_p), ('discount', ctypes.c_void_p),
              ('done', ctypes.c_void_p), ('perf', ctypes.c_void_p),
              ('trace', ctypes.c_void_p), ('obs_format', ctypes.c_int32),
              ('bad_count', ctypes.c_void_p), ('bad_flag', ctypes.c_void_p),
              ('scalar_pitch', ctypes.c_int64),
              ('overlap_ctl', ctypes.c_void_p), ('overlap_ctl_bytes', ctypes.c_int64),
              ('flow_state', ctypes.c_void_p), ('error_flag', ctypes.c_void_p)]


class CampxFlowState(ctypes.Structure):
  """include/campx_hip.h: what the library remembers about a scratch block of one-launch
  rollouts, in the caller's (host) memory."""
  _fields_ = [('tag', ctypes.c_int64), ('B', ctypes.c_int64), ('T', ctypes.c_int64),
              ('pitch', ctypes.c_int64), ('n_dyn', ctypes.c_int64), ('block', ctypes.c_int64)]


ERR_FLOW_TIMEOUT = 1


class CampxError(RuntimeError):
  pass


def _load():
  if not os.path.exists(_LIB_PATH):
    raise ImportError(
        '{} is missing: build it with `python -m campx_amd.build` (needs '
        'hipcc). The fused tier has no CPU fallback.'.format(_LIB_PATH))
  lib = ctypes.CDLL(_LIB_PATH)
  spec_p = ctypes.POINTER(CampxSpec)
  i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
  lib.campx_spec_size.restype = i32
  lib.campx_flow_scratch_bytes.restype = i64
  lib.campx_flow_scratch_bytes.argtypes = [i64, i32]
  lib.campx_spec_size.argtypes = []
  lib.campx_spec_validate.restype = i32
  lib.campx_spec_validate.argtypes = [spec_p]
  lib.campx_spec_compile.restype = i32
  lib.campx_spec_compile.argtypes = [spec_p, vp]
  lib.campx_pair_table_bytes.restype = i64
  lib.campx_pair_table_bytes.argtypes = [spec_p]
  lib.campx_pair_table_build.restype = i32
  lib.campx_pair_table_build.argtypes = [spec_p, vp, vp, vp]
  lib.campx_pair_table_pack.restype = i32
  lib.campx_pair_table_pack.argtypes = [spec_p, vp, vp, vp, vp, vp, vp]
  lib.campx_reset_launch.restype = i32
  lib.campx_reset_launch.argtypes = [spec_p, vp, CampxState, CampxOutputs, i64, vp]
  lib.campx_rollout_launch.restype = i32
  lib.campx_rollout_launch.argtypes = [spec_p, vp, CampxState, vp, CampxOutputs,
                                       i64, i32, i32, vp]
  lib.campx_update_launch.restype = i32
  lib.campx_update_launch.argtypes = [spec_p, vp, CampxState, vp, CampxOutputs,
                                      i64, i32, i32, vp]
  lib.campx_render_launch.restype = i32
  lib.campx_render_launch.argtypes = [spec_p, vp, CampxOutputs, i64, i32, vp]
  shape_p = ctypes.POINTER(CampxShapeSpec)
  lib.campx_update_render_launch.argtypes = [spec_p, vp, CampxState, vp, CampxOutputs, CampxOutputs,
                                             i64, i32, i32, vp]
  lib.campx_update_render_launch.restype = i32
  lib.campx_update_render_shared.argtypes = [spec_p, i64, i32]
  lib.campx_update_render_shared.restype = i32
  lib.campx_flow_shared.argtypes = [spec_p, i64, i32, i64]
  lib.campx_flow_shared.restype = i32
  lib.campx_shape_spec_size.restype = i32
  lib.campx_shape_spec_size.argtypes = []
  lib.campx_shape_spec_validate.restype = i32
  lib.campx_shape_spec_validate.argtypes = [shape_p]
  lib.campx_shape_rollout_launch.restype = i32
  lib.campx_shape_rollout_launch.argtypes = [shape_p, vp, vp, CampxState, vp, vp, CampxOutputs,
                                             i64, i32, i32, i32, vp]
  lib.campx_shape_tables_bytes.restype = i64
  lib.campx_shape_tables_bytes.argtypes = [shape_p]
  lib.campx_shape_tables_build.restype = i32
  lib.campx_shape_tables_build.argtypes = [shape_p, vp, i64]
  lib.campx_shape_scratch_bytes.restype = i64
  lib.campx_shape_scratch_bytes.argtypes = [shape_p, i64, i32]
  wide_p = ctypes.POINTER(CampxWideSpec)
  lib.campx_wide_spec_size.restype = i32
  lib.campx_wide_spec_size.argtypes = []
  lib.campx_wide_spec_validate.restype = i32
  lib.campx_wide_spec_validate.argtypes = [wide_p]
  lib.campx_wide_tables_bytes.restype = i64
  lib.campx_wide_tables_bytes.argtypes = [wide_p]
  lib.campx_wide_tables_build.restype = i32
  lib.campx_wide_tables_build.argtypes = [wide_p, vp, vp]
  lib.campx_wide_reset_launch.restype = i32
  lib.campx_wide_reset_launch.argtypes = [wide_p, vp, CampxState, CampxOutputs, i64, vp]
  lib.campx_wide_rollout_launch.restype = i32
  lib.campx_wide_rollout_launch.argtypes = [wide_p, vp, CampxState, vp, CampxOutputs, i64, i32,
                                            i32, vp]
  lib.campx_wide_rules_size.restype = i32
  lib.campx_wide_rules_size.argtypes = []
  lib.campx_wide_enumerate_launch.restype = i32
  lib.campx_wide_enumerate_launch.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp]
  lib.campx_check_actions_launch.restype = i32
  lib.campx_check_actions_launch.argtypes = [vp, i64, vp, vp]
  lib.campx_onehot_to_ids_launch.restype = i32
  lib.campx_onehot_to_ids_launch.argtypes = [vp, vp, i64, vp, vp]
  lib.campx_config_set.restype = i32
  lib.campx_config_set.argtypes = [ctypes.c_char_p, i64]
  lib.campx_config_get.restype = i32
  lib.campx_config_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(i64)]
  lib.campx_config_string.restype = i32
  lib.campx_config_string.argtypes = [ctypes.c_char_p, i32]
  lib.campx_write_probe_launch.restype = i32
  lib.campx_write_probe_launch.argtypes = [vp, i64, ctypes.c_uint32, vp]
  lib.campx_strerror.restype = ctypes.c_char_p
  lib.campx_strerror.argtypes = [i32]
  lib.campx_last_hip_error.restype = i32
  lib.campx_last_hip_error.argtypes = []
  lib.campx_device_arch.restype = i32
  lib.campx_device_arch.argtypes = [i32, ctypes.c_char_p, i32]
  if lib.campx_shape_spec_size() != ctypes.sizeof(CampxShapeSpec):
    raise ImportError('CampxShapeSpec layout mismatch between gamespec.py and '
                      'libcampx_hip.so: rebuild the library')
  if lib.campx_wide_spec_size() != ctypes.sizeof(CampxWideSpec):
    raise ImportError('CampxWideSpec layout mismatch between gamespec.py and '
                      'libcampx_hip.so: rebuild the library')
  if lib.campx_spec_size() != ctypes.sizeof(CampxSpec):
    raise ImportError('CampxSpec layout mismatch between gamespec.py ({} B) and '
                      'libcampx_hip.so ({} B): rebuild the library'.format(
                          ctypes.sizeof(CampxSpec), lib.campx_spec_size()))
  return lib


lib = _load()

_OPS_PATH = os.path.join(os.path.dirname(_LIB_PATH), 'libcampx_torch.so')


def _load_ops():
  if not os.path.exists(_OPS_PATH):
    raise ImportError(
        '{} is missing: build it with `python -m campx_amd.build` (needs hipcc '
        'and torch). The fused tier has no CPU fallback.'.format(_OPS_PATH))
  import torch
  torch.ops.load_library(_OPS_PATH)
  return torch.ops.campx


ops = _load_ops()
OP_NAMES = ('reset', 'step', 'rollout', 'update', 'render', 'rollout_pipelined', 'shape_rollout', 'wide_rollout',
            'onehot_to_ids', 'check_actions')


def check(code, what):
  if code != 0:
    raise CampxError('{} failed: {} (code {}, hipError {})'.format(
        what, lib.campx_strerror(code).decode(), code,
        lib.campx_last_hip_error()))


# ------------------------------------------------------------------ the library's settings

def config_get(name):
  """The effective value of a library setting (include/campx_hip.h campx_config_get)."""
  value = ctypes.c_int64()
  check(lib.campx_config_get(name.encode(), ctypes.byref(value)), 'campx_config_get({!r})'.format(name))
  return int(value.value)


def config_set(name, value):
  check(lib.campx_config_set(name.encode(), int(value)), 'campx_config_set({!r}, {})'.format(name, value))


def config_string():
  """'name=value name=value ...' of every setting as it is in effect (campx_config_string)."""
  need = lib.campx_config_string(None, 0)
  buf = ctypes.create_string_buffer(need)
  lib.campx_config_string(buf, need)
  return buf.value.decode()


class config(object):
  """`with _hip.config(flow=0, trace_chunk_mb=0): ...` - settings for the calls inside the block
  (process-wide while it lasts: the tests' way of reaching the paths that used to hide behind
  environment variables read once per process), restored afterwards."""

  def __init__(self, **settings):
    self.settings = settings
    self.before = {}

  def __enter__(self):
    for name, value in self.settings.items():
      self.before[name] = config_get(name)
      config_set(name, value)
    return self

  def __exit__(self, *exc):
    for name, value in self.before.items():
      config_set(name, value)
    return False
