"""ctypes loader for tests/emul/libble_emul.so (host build of the kernel's lane functions).

TEST TOOLING ONLY (numerics triage without a GPU); see ble_emul.cpp.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT)
from balloon_learning_environment_amd import _abi  # noqa: E402

_SO = os.path.join(_HERE, 'libble_emul.so')


def build():
  srcs = [os.path.join(_HERE, 'ble_emul.cpp'), os.path.join(_HERE, 'ble_intrinsics.h'),
          os.path.join(_ROOT, 'balloon_learning_environment_amd', 'csrc', 'ble_physics.h'),
          os.path.join(_ROOT, 'balloon_learning_environment_amd', 'csrc', 'ble_step_core.h'),
          os.path.join(_ROOT, 'balloon_learning_environment_amd', 'csrc', 'ble_noise.h'),
          os.path.join(_ROOT, 'balloon_learning_environment_amd', 'csrc', 'ble_decode.h'),
          os.path.join(_ROOT, 'balloon_learning_environment_amd', 'csrc', 'ble_reset.h')]
  if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-include', os.path.join(_HERE, 'ble_intrinsics.h'), '-o', _SO, srcs[0]])
  return _SO


_lib = None


def lib():
  global _lib
  if _lib is None:
    _lib = ctypes.CDLL(build())
  return _lib


def state_from_oracle(ost):
  """fp32 ABI-typed numpy state from an oracle (fp64) state dict."""
  st = {}
  for name in _abi.FIELD_NAMES:
    dt = _abi.FIELD_DTYPES[name]
    if name == 'sunrise_h_rel':
      st[name] = (ost['sunrise_h'] - ost['start_unix']).astype(dt)
    elif name == 'sunset_rel':
      st[name] = (ost['sunset'] - ost['start_unix']).astype(dt)
    else:
      st[name] = np.ascontiguousarray(ost[name]).astype(dt)
  return st


def oracle_from_state(st):
  """Oracle (fp64) state dict holding exactly the fp32 values of an ABI-typed state."""
  import oracle
  n = st['x'].size
  ost = oracle.new_state(n)
  for name in oracle.FLOAT_FIELDS:
    ost[name][:] = st[name].astype(np.float64)
  ost['start_unix'][:] = st['start_unix']
  ost['time_elapsed_s'][:] = st['time_elapsed_s']
  ost['sunrise_h'][:] = st['start_unix'] + st['sunrise_h_rel'].astype(np.int64)
  ost['sunset'][:] = st['start_unix'] + st['sunset_rel'].astype(np.int64)
  for name in oracle.U8_FIELDS:
    ost[name][:] = st[name]
  return ost


def step(st, action, field=None, wind_uv=None, substeps=18):
  n = st['x'].size
  cst = _abi.state_struct({k: v.ctypes.data for k, v in st.items()})
  action = np.ascontiguousarray(action, np.uint8)
  reward = np.empty(n, np.float32); terminal = np.empty(n, np.uint8); eff = np.empty(n, np.uint8)
  flags = np.zeros(1, np.uint32)
  fp = wp = None
  if field is not None:
    field = np.ascontiguousarray(field, np.float32); fp = field.ctypes.data_as(ctypes.c_void_p)
  if wind_uv is not None:
    wind_uv = np.ascontiguousarray(wind_uv, np.float32); wp = wind_uv.ctypes.data_as(ctypes.c_void_p)
  lib().emul_step_f32(ctypes.byref(cst), action.ctypes.data_as(ctypes.c_void_p), fp, wp,
                      reward.ctypes.data_as(ctypes.c_void_p), terminal.ctypes.data_as(ctypes.c_void_p),
                      eff.ctypes.data_as(ctypes.c_void_p), flags.ctypes.data_as(ctypes.c_void_p),
                      ctypes.c_int64(n), ctypes.c_int(substeps))
  return reward, terminal, eff, int(flags[0])
