"""ctypes mirror of include/ble_abi.h (struct ble_state_f32 and field metadata)."""
import ctypes

import numpy as np

# (name, numpy dtype, ctypes scalar type, mutable?) in the order of struct ble_state_f32.
STATE_FIELDS = (
    ('x', np.float32, ctypes.c_float), ('y', np.float32, ctypes.c_float),
    ('pressure', np.float32, ctypes.c_float), ('ambient_temperature', np.float32, ctypes.c_float),
    ('internal_temperature', np.float32, ctypes.c_float), ('envelope_volume', np.float32, ctypes.c_float),
    ('superpressure', np.float32, ctypes.c_float), ('mols_air', np.float32, ctypes.c_float),
    ('battery_charge', np.float32, ctypes.c_float),
    ('acs_power', np.float32, ctypes.c_float), ('acs_mass_flow', np.float32, ctypes.c_float),
    ('solar_charging', np.float32, ctypes.c_float), ('power_load', np.float32, ctypes.c_float),
    ('center_lat_deg', np.float32, ctypes.c_float), ('center_lng_deg', np.float32, ctypes.c_float),
    ('upwelling_infrared', np.float32, ctypes.c_float), ('alpha', np.float32, ctypes.c_float),
    ('start_unix', np.int64, ctypes.c_int64),
    ('time_elapsed_s', np.int32, ctypes.c_int32),
    ('sunrise_h_rel', np.int32, ctypes.c_int32), ('sunset_rel', np.int32, ctypes.c_int32),
    ('status', np.uint8, ctypes.c_uint8), ('last_command', np.uint8, ctypes.c_uint8),
    ('alt_fsm', np.uint8, ctypes.c_uint8), ('env_fsm', np.uint8, ctypes.c_uint8),
    ('power_paused', np.uint8, ctypes.c_uint8),
)
FIELD_NAMES = tuple(f[0] for f in STATE_FIELDS)
FIELD_DTYPES = {f[0]: f[1] for f in STATE_FIELDS}
MUTABLE_FLOATS = FIELD_NAMES[:9]
DERIVED_FLOATS = FIELD_NAMES[9:13]
EPISODE_CONSTS = FIELD_NAMES[13:18]


EPISODE_CACHE_ROWS = 7     # BLE_EPISODE_CACHE_ROWS


class BleStateF32(ctypes.Structure):
  # the per-env arrays, then the optional [EPISODE_CACHE_ROWS][n] float64 cache of per-episode derived constants
  _fields_ = [(name, ctypes.POINTER(ct)) for name, _, ct in STATE_FIELDS] + [('episode_cache', ctypes.POINTER(ctypes.c_double))]


def state_struct(pointers, episode_cache: int = 0):
  """Builds a BleStateF32 from a {field: integer address} mapping (+ the address of the optional episode cache)."""
  st = BleStateF32()
  for name, _, ct in STATE_FIELDS:
    setattr(st, name, ctypes.cast(ctypes.c_void_p(int(pointers[name])), ctypes.POINTER(ct)))
  st.episode_cache = ctypes.cast(ctypes.c_void_p(int(episode_cache) or None), ctypes.POINTER(ctypes.c_double))
  return st


class BleGpHistoryF32(ctypes.Structure):
  """struct ble_gp_history_f32."""
  _fields_ = [('xyp', ctypes.POINTER(ctypes.c_float)), ('elapsed_s', ctypes.POINTER(ctypes.c_int32)),
              ('err_uv', ctypes.POINTER(ctypes.c_float)), ('count', ctypes.POINTER(ctypes.c_int32)),
              ('chol', ctypes.POINTER(ctypes.c_double)), ('n_chol', ctypes.POINTER(ctypes.c_int32)),
              ('chol_stride', ctypes.c_int64)]


class BleNoiseGen(ctypes.Structure):
  """struct ble_noise_gen: the wind-noise generator of a fused rollout (ble_step_n_f32, ABI 3)."""
  _fields_ = [('seed', ctypes.c_uint64), ('episode', ctypes.c_void_p), ('harmonic_cache', ctypes.c_void_p),
              ('env_offset', ctypes.c_int64)]       # (ABI 4: the shard's first environment in the global batch; default 0)
