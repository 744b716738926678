"""Run-time flight vehicles (ABI 5: ble_state_f32.vehicle) through the C ABI, against the reference-generated fixture F16 and the
oracle -- BalloonState's vehicle constants are dataclass fields in the reference (env/balloon/balloon.py:156-173,183) and
power_safety_layer_enabled a per-state switch (:200,305)."""
import ctypes

import numpy as np
import pytest
import torch

import helpers
import oracle
from helpers import FLOORS, STATE_FLOATS, golden, rel_err, traj_state_at
from test_gpu_parity import RTOL, _dev, abi_state_from_oracle, compare_states, oracle_state_from_abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ble():
  from balloon_learning_environment_amd import vec_state
  return vec_state


def test_f16_vehicles_teacher_forced(ble):
  """Every step of every F16 trajectory from the reference's own state: device vs oracle (same float32 inputs, same vehicle) at 1e-5 /
  discrete exact, and device vs the fixture's next state directly (bounded by the reference's own sensitivity to the float32 rounding
  of its inputs, as for F8)."""
  d = golden('f16_vehicles')
  n, steps = d['actions'].shape
  zero_grid = np.zeros((21, 21, 10, 9, 2), np.float32)
  worst_all = {k: 0.0 for k in STATE_FLOATS}
  flown = 0
  for vi in range(len(d['vehicles'])):
    veh = helpers.fixture_vehicle(d, vi)
    mine = np.nonzero(d['vehicle_index'] == vi)[0]
    for s in range(steps):
      rows = mine[d['valid'][mine, s] == 1]
      if rows.size == 0:
        continue
      ost = traj_state_at(d, s, rows)
      sim = ble.VecSimulator(rows.size)
      sim.set_vehicle(**veh)
      sim.set_state(abi_state_from_oracle(ost))
      o2 = oracle_state_from_abi(sim.get_state())
      act = d['actions'][rows, s]
      sim.set_grid(zero_grid)
      w = d['wind_uv'][rows, s].astype(np.float32)
      reward, terminal = sim.step(_dev(act, np.uint8), noise_uv=_dev(w, np.float32))
      ro, to, eo, err = oracle.step(o2, act, wind_uv=w.astype(np.float64), vehicle=veh)
      torch.cuda.synchronize()
      sim.check_errors()
      assert err == 0
      got = sim.get_state()
      worst = compare_states(got, o2, ctx=f'f16 vehicle {vi} step {s}')
      for k, v in worst.items():
        worst_all[k] = max(worst_all[k], v)
      nxt = traj_state_at(d, s + 1, rows)
      for k in STATE_FLOATS:
        direct = rel_err(got[k], nxt[k], FLOORS[k]); sens = rel_err(o2[k], nxt[k], FLOORS[k])
        assert (direct - sens).max() <= RTOL, f'f16 vehicle {vi} step {s} {k}: {direct.max():.3g} vs the fixture (sensitivity {sens.max():.3g})'
      for k in ('status', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s'):
        same = o2[k] == nxt[k]
        np.testing.assert_array_equal(got[k][same], nxt[k][same], err_msg=f'f16 vehicle {vi} step {s} {k} vs the fixture')
      np.testing.assert_array_equal(sim.effective_action.cpu().numpy(), eo)
      np.testing.assert_array_equal(terminal.cpu().numpy(), to)
      np.testing.assert_allclose(reward.cpu().numpy(), ro, rtol=RTOL, atol=RTOL)
      flown += rows.size
  assert flown == int(d['valid'].sum())
  print('worst relative errors (f16):', {k: f'{v:.2g}' for k, v in worst_all.items()})


def test_f16_cold_start_with_the_vehicle(ble):
  """stable_init.cold_start_to_stable_params with the vehicle's own volume / masses / lift gas (stable_init.py:132-157): the
  device reset (sample = 0) against F16's cold-start values."""
  d = golden('f16_vehicles')
  for vi in range(len(d['vehicles'])):
    veh = helpers.fixture_vehicle(d, vi)
    mine = np.nonzero(d['vehicle_index'] == vi)[0]
    ost = traj_state_at(d, 0, mine)
    sim = ble.VecSimulator(mine.size)
    sim.set_vehicle(**veh)
    sim.set_state(abi_state_from_oracle(ost))
    sim.reset_device(seed=0, sample=False)
    torch.cuda.synchronize(); sim.check_errors()
    got = sim.get_state()
    for k in ('ambient_temperature', 'internal_temperature', 'mols_air', 'envelope_volume', 'superpressure'):
      e = rel_err(got[k], d['cold_' + k][mine], FLOORS[k])
      assert e.max() <= RTOL, (vi, k, e.max())


def test_run_time_vehicle_with_default_values_flies_the_default_kernels_bits(ble):
  """The second instantiation reads from scalar registers exactly the numbers the default one folds at compile time: handed the
  reference's defaults explicitly (a non-NULL vehicle), it produces the bits of the NULL-vehicle kernels -- one lane per environment,
  fused and unfused, with and without the in-kernel noise generator."""
  from balloon_learning_environment_amd import _abi, _lib
  n, k = 4096 + 17, 6
  field = (np.random.default_rng(5).standard_normal((21, 21, 10, 9, 2)) * 6.0).astype(np.float32)
  acts = torch.from_numpy(np.random.default_rng(6).integers(0, 3, (k, n)).astype(np.uint8)).cuda()
  outs = []
  for explicit in (False, True):
    sim = ble.VecSimulator(n)
    sim.set_grid(field)
    if explicit:      # (set_vehicle() maps all-default fields to NULL: build the struct by hand)
      veh = _abi.BleVehicle(reserved_=0, **_abi.VEHICLE_DEFAULTS)
      _abi.set_vehicle(sim._struct, veh)
    sim.reset_device(seed=77)
    r = torch.zeros(k, n, device='cuda'); t = torch.zeros(k, n, dtype=torch.uint8, device='cuda')
    with _lib.step_form(1):
      sim.step_n(acts, r, t)
      sim.step_n(acts, r, t, noise_seed=9)
      rr, tt = sim.step(acts[0])
    torch.cuda.synchronize(); sim.check_errors()
    outs.append((sim.get_state(), r.cpu().numpy(), t.cpu().numpy(), rr.cpu().numpy().copy(), sim.effective_action.cpu().numpy().copy()))
  a, b = outs
  for key in a[0]:
    np.testing.assert_array_equal(a[0][key], b[0][key], err_msg=key)
  for x, y in zip(a[1:], b[1:]):
    np.testing.assert_array_equal(x, y)


def test_default_vehicle_struct_and_validation(ble):
  from balloon_learning_environment_amd import _abi, _lib
  lib = _lib.lib()
  v = _abi.BleVehicle()
  assert lib.ble_vehicle_default(ctypes.byref(v)) == 0
  assert {k: getattr(v, k) for k in _abi.VEHICLE_DEFAULTS} == _abi.VEHICLE_DEFAULTS
  assert lib.ble_vehicle_default(None) == -1
  sim = ble.VecSimulator(64)
  sim.set_grid(np.zeros((21, 21, 10, 9, 2), np.float32))
  sim.reset_device(seed=1)
  for bad in (dict(envelope_volume_base=0.0), dict(battery_capacity_wh=-1.0), dict(envelope_max_superpressure=200.0), dict(envelope_cod=float('nan'))):
    sim.set_vehicle(**bad)
    with pytest.raises(_lib.BleLibraryError):
      sim.step(torch.ones(64, dtype=torch.uint8, device='cuda'))
    with pytest.raises(_lib.BleLibraryError):
      sim.reset_device(seed=1)
  with pytest.raises(TypeError):
    sim.set_vehicle(no_such_field=1.0)
  sim.set_vehicle()
  sim.step(torch.ones(64, dtype=torch.uint8, device='cuda'))
  torch.cuda.synchronize(); sim.check_errors()


def test_large_batch_with_a_vehicle_matches_oracle_sampled(ble):
  """A 65 536-environment batch of a non-default vehicle -- its own cold start on the device, then one agent step with random
  actions in a grid wind: 512 sampled environments against the oracle at the parity bar (the reset's draws do not depend on the
  vehicle, its cold start does)."""
  d = golden('f16_vehicles')
  veh = helpers.fixture_vehicle(d, 4)
  n = 65536
  field = (np.random.default_rng(11).standard_normal((21, 21, 10, 9, 2)) * 6.0).astype(np.float32)
  sim = ble.VecSimulator(n)
  sim.set_vehicle(**veh)
  sim.set_grid(field)
  sim.reset_device(seed=2024)
  ref = ble.VecSimulator(n)
  ref.set_grid(field)
  ref.reset_device(seed=2024)
  rows = np.random.default_rng(1).choice(n, 512, replace=False)
  before = {key: v[rows] for key, v in sim.get_state().items()}
  default_before = {key: v[rows] for key, v in ref.get_state().items()}
  for key in ('x', 'y', 'pressure', 'center_lat_deg', 'upwelling_infrared', 'alpha', 'start_unix'):
    np.testing.assert_array_equal(before[key], default_before[key])
  assert np.abs(before['mols_air'] - default_before['mols_air']).max() > 10.0
  out, err = oracle.stable_init(before['pressure'], before['center_lat_deg'], before['center_lng_deg'], before['x'], before['y'],
                                before['start_unix'], before['upwelling_infrared'], before['alpha'], vehicle=veh)
  for key, v in out.items():
    assert rel_err(before[key], v, FLOORS[key]).max() <= RTOL, key
  acts = np.random.default_rng(2).integers(0, 3, n).astype(np.uint8)
  reward, terminal = sim.step(torch.from_numpy(acts).cuda())
  torch.cuda.synchronize(); sim.check_errors()
  got = {key: v[rows] for key, v in sim.get_state().items()}
  o = oracle_state_from_abi(before)
  ro, to, eo, err = oracle.step(o, acts[rows], field=field, vehicle=veh)
  assert err == 0
  compare_states(got, o, ctx='65536 envs, vehicle 4')
  np.testing.assert_array_equal(sim.effective_action.cpu().numpy()[rows], eo)
  np.testing.assert_allclose(reward.cpu().numpy()[rows], ro, rtol=RTOL, atol=RTOL)
