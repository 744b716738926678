"""Loader for libble_hip.so (the C ABI in include/ble_abi.h).

The HIP library is the ONLY compute path of this package.  If it is missing or cannot
be loaded the import of anything that needs it raises -- there is no CPU fallback.
"""
import ctypes
import os
import subprocess

# One HIP runtime per process: torch ships its own libamdhip64 (SONAME libamdhip64.so.7, found
# through torch/lib's RPATH).  Importing torch first makes libble_hip.so's NEEDED
# libamdhip64.so.7 resolve to that already-loaded runtime, so torch's device pointers and
# hipStream_t handles are valid in our launches.  Loaded the other way round the process gets
# two runtimes and launches fail with hipErrorNoDevice (100).
import torch  # noqa: F401  (must precede ctypes.CDLL(LIB_PATH))

from balloon_learning_environment_amd import _abi

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BLE_HIP_LIB') or os.path.join(_PKG_DIR, 'libble_hip.so')   # override: experiments only
_SOURCES = [os.path.join(_PKG_DIR, 'csrc', f) for f in ('ble_kernels.hip', 'ble_step_core.h', 'ble_physics.h', 'ble_intrinsics.h', 'ble_reset.h',
                                                          'ble_observe.h', 'ble_noise.h', 'ble_decode.h', 'ble_step_split.h')]
_HEADER = os.path.join(os.path.dirname(_PKG_DIR), 'include', 'ble_abi.h')

ABI_VERSION = 5
NOISE_PRIMITIVE_VERSION = 2     # BLE_NOISE_PRIMITIVE_VERSION this mirror, oracle/noise_oracle.py and tests/golden/f14 were made with
BLE_OK = 0
FLAG_PRESSURE_RANGE, FLAG_ABSORPTIVITY, FLAG_SOLAR_RANGE, FLAG_POWER_TABLE, FLAG_NONFINITE = 1, 2, 4, 16, 32
FLAG_GP_WINDOW, FLAG_PRESSURE_SEARCH, FLAG_DAY_CYCLE = 64, 128, 256
OBS_DIM, GP_CAPACITY, GP_CHOL_STRIDE = 1099, 128, 7620
ROW_DOUBLES = 26        # BLE_ROW_DOUBLES
NOISE_CACHE_ROWS = 53

# every symbol include/ble_abi.h declares
EXPORTS = ('ble_abi_version', 'ble_noise_primitive_version', 'ble_vehicle_default', 'ble_last_hip_error', 'ble_device_count', 'ble_set_step_form', 'ble_step_f32', 'ble_step_n_f32', 'ble_reset_f32', 'ble_reset_at_f32', 'ble_wind_noise_at_f32', 'ble_observe_f32', 'ble_observe_forecast_f32', 'ble_decode_flow_fields_f32', 'ble_wind_noise_f32', 'ble_forecast_f32',
           'ble_forecast_column_f32', 'ble_state_rows_f64', 'ble_power_table_f32', 'ble_probe_atmosphere_f32', 'ble_probe_atmosphere_at_height_f64', 'ble_probe_solar_f32', 'ble_probe_latlng_f64',
           'ble_probe_solar_power_f32', 'ble_probe_thermal_f32', 'ble_probe_sp_volume_f32', 'ble_probe_thermal_vehicle_f32', 'ble_probe_sp_volume_vehicle_f32', 'ble_probe_acs_f32', 'ble_probe_safety_f32',
           'ble_probe_f64_prims')


class BleLibraryError(RuntimeError):
  pass


def build(force: bool = False, verbose: bool = False) -> str:
  """Compiles csrc/ble_kernels.hip for gfx950 into libble_hip.so (in-tree)."""
  deps = _SOURCES + [_HEADER]
  stale = (not os.path.exists(LIB_PATH) or
           any(os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps))
  if force or stale:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
      hipcc = 'hipcc'
    # -ffp-contract=on: a * b + c fuses where the SOURCE writes it in one expression and nowhere else.  hipcc's default
    # ("fast") also fuses across statements wherever the optimiser happens to see a product next to a sum, which depends on the
    # code around it (inlining, vectorisation): the same lane function then rounds differently in two kernels.  The transition
    # exists in three kernels that must agree bit for bit (one lane per environment, four waves per environment, noise
    # generated in-kernel); measured neutral on both hot kernels (profiles/r04_notes.md).
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=on', '-fPIC', '-shared', '-I', os.path.join(_PKG_DIR, 'csrc'),
           '-o', LIB_PATH, _SOURCES[0]]
    if verbose:
      cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    subprocess.check_call(cmd)
  return LIB_PATH


_lib = None
_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


def lib():
  """Returns the loaded library; raises BleLibraryError if it is not there."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise BleLibraryError(
        f'{LIB_PATH} not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
        '(hipcc --offload-arch=gfx950). This package has no CPU fallback.')
  try:
    l = ctypes.CDLL(LIB_PATH)
  except OSError as e:  # e.g. libamdhip64 missing
    raise BleLibraryError(f'cannot load {LIB_PATH}: {e}') from e
  for name in EXPORTS:
    if not hasattr(l, name):
      raise BleLibraryError(f'{LIB_PATH} does not export {name}')
  if l.ble_abi_version() != ABI_VERSION:
    raise BleLibraryError('ABI version mismatch between libble_hip.so and the Python mirror')
  if l.ble_noise_primitive_version() != NOISE_PRIMITIVE_VERSION:
    raise BleLibraryError(f'libble_hip.so evaluates wind-noise primitive version {l.ble_noise_primitive_version()}, this package (its oracle and '
                          f'fixtures) was made with {NOISE_PRIMITIVE_VERSION}: recorded noise seeds would fly another wind')
  st = ctypes.POINTER(_abi.BleStateF32)
  l.ble_step_f32.argtypes = [st, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]
  l.ble_step_n_f32.argtypes = [st, _vp, _vp, _i64, ctypes.POINTER(_abi.BleNoiseGen), _vp, _vp, _vp, _vp, _i64, _int, _int, _vp]
  l.ble_reset_f32.argtypes = [st, _vp, ctypes.c_uint64, _vp, _int, _vp, _i64, _vp]
  l.ble_reset_at_f32.argtypes = [st, _vp, ctypes.c_uint64, _vp, _int, _vp, _i64, _i64, _vp]
  l.ble_observe_f32.argtypes = [st, _vp, _i64, _vp, _vp, ctypes.POINTER(_abi.BleGpHistoryF32), _int, _vp, _vp, _i64, _vp]
  l.ble_observe_forecast_f32.argtypes = [st, _vp, _i64, _vp, _vp, _vp, ctypes.POINTER(_abi.BleGpHistoryF32), _int, _vp, _vp, _i64, _vp]
  l.ble_decode_flow_fields_f32.argtypes = [_vp, _vp, _i64, _vp]
  l.ble_wind_noise_f32.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_uint64, _vp, _int, _vp, _vp, _i64, _vp]
  l.ble_wind_noise_at_f32.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_uint64, _vp, _int, _vp, _vp, _i64, _i64, _vp]
  l.ble_forecast_f32.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]
  l.ble_forecast_column_f32.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp]
  l.ble_power_table_f32.argtypes = [_vp, _vp, _vp, _vp, _i64, _vp]
  l.ble_state_rows_f64.argtypes = [st, _i64, _i64, _vp, _i64, _vp]
  l.ble_probe_atmosphere_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _vp]
  l.ble_probe_atmosphere_at_height_f64.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _vp]
  l.ble_probe_solar_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]
  l.ble_probe_latlng_f64.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]
  l.ble_probe_solar_power_f32.argtypes = [_vp, _vp, _vp, _vp, _i64, _vp]
  l.ble_probe_thermal_f32.argtypes = [_vp] * 9 + [_i64, _vp]
  l.ble_probe_sp_volume_f32.argtypes = [_vp] * 5 + [_i64, _vp]
  l.ble_probe_thermal_vehicle_f32.argtypes = [ctypes.POINTER(_abi.BleVehicle)] + [_vp] * 9 + [_i64, _vp]
  l.ble_probe_sp_volume_vehicle_f32.argtypes = [ctypes.POINTER(_abi.BleVehicle)] + [_vp] * 5 + [_i64, _vp]
  l.ble_probe_acs_f32.argtypes = [_vp] * 4 + [_i64, _vp]
  l.ble_probe_safety_f32.argtypes = [_int, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _i64, _vp]
  l.ble_probe_f64_prims.argtypes = [_vp, _vp, _int, _i64, _vp]
  l.ble_set_step_form.argtypes = [_int]
  l.ble_vehicle_default.argtypes = [ctypes.POINTER(_abi.BleVehicle)]
  for name in EXPORTS:
    getattr(l, name).restype = _int
  _lib = l
  return l


def check(code: int, what: str) -> None:
  if code != BLE_OK:
    hip = _lib.ble_last_hip_error() if _lib is not None else 0
    raise BleLibraryError(f'{what} failed with BLE error {code} (hipError_t {hip})')


class step_form:
  """`with _lib.step_form(waves): ...` forces the transition kernel's form (0 automatic, 1 one lane per environment,
  4 / 2 wavefronts per environment: `ble_set_step_form`) for the launches inside the block and restores the previous
  setting.  A/B runs and the bit-identity tests; the automatic choice is by batch size (BLE_SPLIT_MAX_ENVS).  (2 exists in
  experiment builds only: the product library refuses it since ABI 5.)"""

  def __init__(self, waves_per_env: int):
    self.waves = int(waves_per_env)

  def __enter__(self):
    self.before = lib().ble_set_step_form(self.waves)
    if self.before < 0:
      raise ValueError(f'step form must be 0, 1 or 4, not {self.waves}')
    return self

  def __exit__(self, *exc):
    lib().ble_set_step_form(self.before)
    return False


def set_step_form(mode) -> int:
  """`ble_set_step_form` with the spelling of the former BLE_STEP_SPLIT switch: None / 'auto' -> automatic, '0' -> one lane per
  environment, '1' / '4' -> four wavefronts ('2' -> two: experiment builds only, BLE_E_INVALID_ARG from the product library).  Returns
  the previous setting; raises ValueError when the library refuses the form."""
  waves = {None: 0, 'auto': 0, '0': 1, '1': 4, '4': 4, '2': 2}[mode if mode is None else str(mode)]
  before = lib().ble_set_step_form(waves)
  if before < 0:
    raise ValueError(f'this library has no step form {mode!r}')
  return before
