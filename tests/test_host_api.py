"""Host-side logic and the C-ABI surface, CPU only (no compute calls without a GPU)."""
import ctypes
import datetime as dt
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from balloon_learning_environment_amd import _abi, _lib
from balloon_learning_environment_amd.env.balloon import balloon, control
from balloon_learning_environment_amd.env import balloon_env, simulator_data
from balloon_learning_environment_amd.utils import units

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
  header = open(os.path.join(ROOT, 'include', 'ble_abi.h')).read()
  declared = set(re.findall(r'^int (ble_\w+)\(', header, re.M))
  assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
  path = _lib.build()                     # hipcc cross-compiles gfx950 without a GPU
  out = subprocess.check_output(['nm', '-D', '--defined-only', path]).decode()
  exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
  assert declared <= exported
  lib = ctypes.CDLL(path)                 # loads on a CPU-only box (libamdhip64 is present)
  assert lib.ble_abi_version() == _lib.ABI_VERSION


def test_state_struct_matches_header_order():
  header = open(os.path.join(ROOT, 'include', 'ble_abi.h')).read()
  body = header[header.index('typedef struct ble_state_f32 {'):header.index('} ble_state_f32;')]
  names = re.findall(r'\*\s*(\w+);', body)
  assert tuple(names) == _abi.FIELD_NAMES + ('episode_cache', 'vehicle')        # the per-env arrays, the optional cache, the optional vehicle (ABI 5)
  assert [f[0] for f in _abi.BleStateF32._fields_] == names
  assert ctypes.sizeof(_abi.BleStateF32) == 8 * len(names)
  assert int(re.search(r'#define BLE_EPISODE_CACHE_ROWS (\d+)', header).group(1)) == _abi.EPISODE_CACHE_ROWS


def test_vehicle_struct_matches_header_and_reference_defaults():
  """struct ble_vehicle (ABI 5) field for field; its defaults are the reference's BalloonState defaults (balloon.py:156-173,183,200 --
  tests/golden/f16_vehicles.npz records them from the reference's own dataclass), what ble_vehicle_default() writes and what the oracle
  uses."""
  header = open(os.path.join(ROOT, 'include', 'ble_abi.h')).read()
  body = header[header.index('typedef struct ble_vehicle {'):header.index('} ble_vehicle;')]
  names = re.findall(r'(?:double|int32_t)\s+(\w+);', body)
  assert names == [f[0] for f in _abi.BleVehicle._fields_]
  assert ctypes.sizeof(_abi.BleVehicle) == 8 * 11 + 8
  assert _abi.VEHICLE_DEFAULTS == oracle.VEHICLE_DEFAULTS and tuple(_abi.VEHICLE_DEFAULTS) == _abi.VEHICLE_FIELDS
  lib = ctypes.CDLL(_lib.build())          # a host function: callable without a GPU
  v = _abi.BleVehicle()
  lib.ble_vehicle_default.argtypes = [ctypes.POINTER(_abi.BleVehicle)]
  assert lib.ble_vehicle_default(ctypes.byref(v)) == 0
  assert {k: getattr(v, k) for k in _abi.VEHICLE_DEFAULTS} == _abi.VEHICLE_DEFAULTS
  import helpers
  d = helpers.golden('f16_vehicles')
  assert tuple(str(k) for k in d['vehicle_fields']) == _abi.VEHICLE_FIELDS
  for vi in range(len(d['vehicles'])):          # every field F16 does not vary holds the reference's default
    row = dict(zip(_abi.VEHICLE_FIELDS, d['vehicles'][vi]))
    changed = helpers.fixture_vehicle(d, vi)
    assert all(row[k] == _abi.VEHICLE_DEFAULTS[k] for k in row if k not in changed)
  assert set().union(*(helpers.fixture_vehicle(d, vi) for vi in range(len(d['vehicles'])))) == set(_abi.VEHICLE_FIELDS)      # ... and F16 varies them all
  assert _abi.vehicle_struct() is None and _abi.vehicle_struct(envelope_mass=68.5) is None
  assert _abi.vehicle_struct(envelope_mass=70.0).envelope_mass == 70.0
  with pytest.raises(TypeError):
    _abi.vehicle_struct(mass=1.0)


def test_noise_primitive_version_is_one_number_everywhere():
  """The wind-noise primitive is this repository's own: the library, the header, the Python mirror, the oracle and the committed
  fixture F14 carry ONE version of its bit pattern (include/ble_abi.h::BLE_NOISE_PRIMITIVE_VERSION)."""
  import helpers
  import noise_oracle
  header = open(os.path.join(ROOT, 'include', 'ble_abi.h')).read()
  v = int(re.search(r'#define BLE_NOISE_PRIMITIVE_VERSION (\d+)', header).group(1))
  lib = ctypes.CDLL(_lib.build())
  assert v == _lib.NOISE_PRIMITIVE_VERSION == noise_oracle.PRIMITIVE_VERSION == lib.ble_noise_primitive_version()
  assert int(helpers.golden('f14_wind_noise')['noise_primitive_version']) == v


def test_gp_history_struct_matches_header_order():
  header = open(os.path.join(ROOT, 'include', 'ble_abi.h')).read()
  body = header[header.index('typedef struct ble_gp_history_f32 {'):header.index('} ble_gp_history_f32;')]
  names = re.findall(r'(?:\*|int64_t)\s*(\w+);', body)       # the pointer members, then the int64 slab stride (ABI 2)
  assert names == [f[0] for f in _abi.BleGpHistoryF32._fields_] and names[-1] == 'chol_stride'
  assert ctypes.sizeof(_abi.BleGpHistoryF32) == 8 * len(names)
  assert int(re.search(r'#define BLE_ABI_VERSION (\d+)', header).group(1)) == _lib.ABI_VERSION == 5
  assert int(re.search(r'#define BLE_OBS_DIM (\d+)', header).group(1)) == _lib.OBS_DIM
  assert int(re.search(r'#define BLE_GP_CAPACITY (\d+)', header).group(1)) == _lib.GP_CAPACITY
  assert int(re.search(r'#define BLE_GP_CHOL_STRIDE (\d+)', header).group(1)) == _lib.GP_CHOL_STRIDE


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
  monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
  monkeypatch.setattr(_lib, '_lib', None)
  with pytest.raises(_lib.BleLibraryError, match='no CPU fallback'):
    _lib.lib()


def test_no_cpu_path_without_gpu():
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from balloon_learning_environment_amd import vec_state
  with pytest.raises(RuntimeError, match='no CPU path'):
    vec_state.VecSimulator(4, 'cuda:0')
  with pytest.raises(RuntimeError, match='no CPU path'):
    vec_state.VecSimulator(4, 'cpu')


def test_reference_module_mirrors_have_no_cpu_path_either():
  """env/balloon/{standard_atmosphere, solar, thermal, acs, stable_init}.py mirror the reference's modules on the device
  functions of the transition: importable anywhere, but a call without a HIP device raises instead of computing on the host."""
  import torch
  from balloon_learning_environment_amd.env.balloon import acs, solar, stable_init, standard_atmosphere, thermal    # noqa: F401
  a = standard_atmosphere.Atmosphere(np.array([0, 7], np.uint32))
  assert 0.0 <= a.alpha < 1.0 and standard_atmosphere.Atmosphere(np.array([0, 7], np.uint32)).alpha == a.alpha
  assert solar.balloon_shadow(45.0, 3.0) == 0.4392 and thermal.absorptivity_ir(210.0) == pytest.approx(0.04587)
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from balloon_learning_environment_amd.env.balloon import altitude_safety, envelope_safety, power_safety    # noqa: F401
  from balloon_learning_environment_amd.env.balloon import power_table
  for call in (lambda: a.at_pressure(8000.0), lambda: a.at_height(units.Distance(feet=50000.0)), lambda: power_table.lookup(1.1, 0.5),
               lambda: solar.solar_power(30.0, 8000.0), lambda: acs.get_most_efficient_power(1.1),
               lambda: envelope_safety.EnvelopeSafetyLayer(2380.0).get_action(control.AltitudeControlCommand.DOWN, 100.0),
               lambda: altitude_safety.AltitudeSafetyLayer().get_action(control.AltitudeControlCommand.DOWN, a, 9000.0),
               lambda: thermal.d_balloon_temperature_dt(1804.0, 68.5, 210.0, 215.0, 8000.0, 30.0, 1360.0, 250.0)):
    with pytest.raises(RuntimeError, match='no CPU path'):
      call()


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, 'balloon_learning_environment_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.h', '.hip')):
        src = open(os.path.join(dirpath, f)).read()
        assert 'import oracle' not in src and 'libble_oracle' not in src and 'ble_oracle' not in src, f


def test_units_behave_like_the_reference():
  d = units.Distance(km=1.5) + units.Distance(feet=1000.0)
  assert d.m == pytest.approx(1500 + 304.8)
  assert (units.Velocity(mps=3.0) * dt.timedelta(seconds=10)).m == 30.0
  assert (units.Power(watts=360.0) * dt.timedelta(seconds=10)).watt_hours == pytest.approx(1.0)
  assert units.Energy(watt_hours=5.0) / units.Energy(watt_hours=10.0) == 0.5
  assert units.relative_distance(units.Distance(m=3.0), units.Distance(m=4.0)).m == 5.0
  with pytest.raises(NotImplementedError):
    _ = units.Distance(m=1.0) + 3.0
  assert units.datetime(2013, 9, 21, 18).timestamp() == 1379786400


def test_balloon_state_row_roundtrip_and_properties():
  row = dict(x=1234.5, y=-777.0, pressure=8123.0, ambient_temperature=201.0, internal_temperature=215.0,
             envelope_volume=1810.0, superpressure=1234.0, mols_air=1500.0, battery_charge=2900.0, acs_power=100.0,
             acs_mass_flow=0.01, solar_charging=300.0, power_load=220.4, center_lat_deg=2.5, center_lng_deg=-70.0,
             upwelling_infrared=260.0, alpha=0.5, start_unix=1364203532, time_elapsed_s=360, sunrise_h_rel=5000,
             sunset_rel=40000, status=0, last_command=0, alt_fsm=1, env_fsm=3, power_paused=1)
  s = balloon.state_from_row(row)
  assert s.x.m == 1234.5 and s.time_elapsed == dt.timedelta(seconds=360) and s.status == balloon.BalloonStatus.OK
  assert s.navigation_is_paused and s.altitude_safety_layer.navigation_is_paused
  assert s.pressure_ratio == pytest.approx((8123.0 + 1234.0) / 8123.0)   # balloon_test.py:85-91 semantics
  s.superpressure = -52.0
  assert s.pressure_ratio == 1.0
  s.superpressure = 1234.0
  back = balloon.row_from_state(s, 0.5)
  for k, v in row.items():
    assert back[k] == pytest.approx(v), k
  # (BalloonState.latlng is a device probe since round 5: its comparison with the oracle's spherical offset is a -m gpu test)


@pytest.mark.gpu          # (BalloonState.excess_energy reads the sun through the device probe since round 5)
@pytest.mark.parametrize('x,y,batt,acs_w,cmd,t', [
    (1000.0, -1000.0, 2900.0, 0.0, 1, '2013-03-25T12:00:00'),
    (60000.0, 20000.0, 2900.0, 0.0, 1, '2013-03-25T12:00:00'),
    (0.0, 0.0, 1000.0, 250.0, 0, '2013-03-25T00:00:00'),
    (120000.0, 0.0, 3050.0, 100.0, 0, '2013-03-25T12:00:00'),
])
def test_host_reward_mirror_matches_oracle(x, y, batt, acs_w, cmd, t):
  now = dt.datetime.fromisoformat(t).replace(tzinfo=dt.timezone.utc)
  s = balloon.BalloonState(center_latlng=balloon.LatLng(0.0, 0.0), date_time=now, x=units.Distance(m=x),
                           y=units.Distance(m=y), pressure=8000.0, battery_charge=units.Energy(watt_hours=batt),
                           acs_power=units.Power(watts=acs_w), last_command=control.AltitudeControlCommand(cmd))
  got = balloon_env.perciatelli_reward_function(simulator_data.SimulatorState(s, None, simulator_data.Atmosphere(0.5)))
  ref = oracle.reward_only(x, y, 8000.0, batt, acs_w, cmd, 0.0, 0.0, int(now.timestamp()), 0)
  assert got == pytest.approx(ref, rel=1e-12)


def test_gym_registration_module_without_gym():
  """env/gym.py mirrors the reference's register_env(); gym itself is optional."""
  import importlib
  import pytest
  mod = importlib.import_module('balloon_learning_environment_amd.env.gym')
  assert mod.ENV_ID == 'BalloonLearningEnvironment-v0' and mod.ENTRY_POINT.endswith(':BalloonEnv')
  try:
    import gym  # noqa: F401
  except ImportError:
    with pytest.raises(ImportError):
      mod.register_env()
  else:
    mod.register_env(); mod.register_env()      # idempotent


def test_package_holds_one_implementation():
  """no NumPy / SciPy twin of the device paths inside the product package (the host sampler, WindGP,
  feature constructor and pressure-range search of rounds 1-4 live next to the tests now)."""
  pkg = os.path.join(ROOT, 'balloon_learning_environment_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith('.py'):
        src = open(os.path.join(dirpath, f)).read()
        assert 'scipy' not in src and 'np.linalg' not in src and 'numpy.linalg' not in src, f
        assert 'reset_host' not in src.replace('tests/reset_host.py', '') and 'wind_gp' not in src, f
  for gone in ('reset_host.py', 'env/wind_gp.py', 'env/balloon/pressure_range_builder.py'):
    assert not os.path.exists(os.path.join(pkg, gone)), gone
