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


class BleVehicle(ctypes.Structure):
  """struct ble_vehicle (ABI 5): BalloonState's flight-vehicle constants (reference env/balloon/balloon.py:156-173), mols_lift_gas
  (:183) and power_safety_layer_enabled (:200) -- a HOST struct of doubles, one per call."""
  _fields_ = [('envelope_volume_base', ctypes.c_double), ('envelope_volume_dv_pressure', ctypes.c_double), ('envelope_mass', ctypes.c_double),
              ('envelope_max_superpressure', ctypes.c_double), ('envelope_cod', ctypes.c_double), ('payload_mass', ctypes.c_double),
              ('nighttime_power_load_w', ctypes.c_double), ('daytime_power_load_w', ctypes.c_double),
              ('acs_valve_hole_diameter_m', ctypes.c_double), ('battery_capacity_wh', ctypes.c_double), ('mols_lift_gas', ctypes.c_double),
              ('power_safety_layer_enabled', ctypes.c_int32), ('reserved_', ctypes.c_int32)]


VEHICLE_FIELDS = tuple(f[0] for f in BleVehicle._fields_[:-1])
# the reference's defaults (balloon.py:156-173,183,200) == what ble_vehicle_default() writes (tests/test_host_api.py)
VEHICLE_DEFAULTS = dict(envelope_volume_base=1804.0, envelope_volume_dv_pressure=0.0199, envelope_mass=68.5, envelope_max_superpressure=2380.0,
                        envelope_cod=0.25, payload_mass=92.5, nighttime_power_load_w=183.7, daytime_power_load_w=120.4,
                        acs_valve_hole_diameter_m=0.04, battery_capacity_wh=3058.56, mols_lift_gas=6830.0, power_safety_layer_enabled=1)


def vehicle_struct(**overrides):
  """A BleVehicle with the reference's defaults and the given fields replaced; None when nothing differs from the defaults
  (ble_state_f32.vehicle == NULL: the kernels with compile-time vehicle constants)."""
  unknown = set(overrides) - set(VEHICLE_DEFAULTS)
  if unknown:
    raise TypeError(f'unknown vehicle field(s) {sorted(unknown)}')
  values = dict(VEHICLE_DEFAULTS)
  values.update({k: (int(bool(v)) if k == 'power_safety_layer_enabled' else float(v)) for k, v in overrides.items()})
  if values == VEHICLE_DEFAULTS:
    return None
  return BleVehicle(reserved_=0, **values)


class BleStateF32(ctypes.Structure):
  # the per-env arrays, then the optional [EPISODE_CACHE_ROWS][n] float64 cache of per-episode derived constants, then the optional
  # HOST pointer to the vehicle (ABI 5)
  _fields_ = ([(name, ctypes.POINTER(ct)) for name, _, ct in STATE_FIELDS] + [('episode_cache', ctypes.POINTER(ctypes.c_double))] +
              [('vehicle', ctypes.POINTER(BleVehicle))])


def state_struct(pointers, episode_cache: int = 0, vehicle=None):
  """Builds a BleStateF32 from a {field: integer address} mapping (+ the address of the optional episode cache, + an optional
  BleVehicle, which the returned struct keeps alive)."""
  st = BleStateF32()
  for name, _, ct in STATE_FIELDS:
    setattr(st, name, ctypes.cast(ctypes.c_void_p(int(pointers[name])), ctypes.POINTER(ct)))
  st.episode_cache = ctypes.cast(ctypes.c_void_p(int(episode_cache) or None), ctypes.POINTER(ctypes.c_double))
  set_vehicle(st, vehicle)
  return st


def set_vehicle(st, vehicle) -> None:
  """Points st.vehicle at `vehicle` (a BleVehicle, kept alive by `st`) or at NULL (None: the reference's defaults)."""
  st._vehicle_keepalive = vehicle
  st.vehicle = ctypes.pointer(vehicle) if vehicle is not None else ctypes.POINTER(BleVehicle)()


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
