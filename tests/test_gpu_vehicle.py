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


def test_f16_observation_with_the_vehicle(ble):
  """ble_observe_f32 with ble_state_f32.vehicle set: the reference's own feature vectors of F16's five vehicles (battery_soc, excess
  energy, the vehicle's reachable pressure range) -- against the feature oracle on the same float32 inputs at 1e-5, and against the
  fixture within the reference's own sensitivity to that rounding.  One simulator per vehicle (a batch flies one vehicle)."""
  import features_oracle
  from test_gpu_observe import check, row32, rows_to_arrays
  g = golden('f16_vehicles')
  field = helpers.fixture_field(g)
  n_steps = g['obs_features'].shape[1]
  for j in range(len(g['vehicles'])):
    veh = helpers.fixture_vehicle(g, j)
    sim = ble.VecSimulator(1)
    sim.set_vehicle(**veh)
    sim.set_grid(torch.from_numpy(field).cuda())
    fo = features_oracle.FeatureOracle(field, float(np.float32(g['obs_alpha'][j])), vehicle=veh)
    got = np.zeros((1, n_steps, 1099), np.float32); same = np.zeros_like(got)
    for i in range(n_steps):
      row = {k: float(g['obs_' + k][j, i]) for k in STATE_FLOATS}
      for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s'):
        row[k] = int(g['obs_' + k][j, i])
      for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha'):
        row[k] = float(g['obs_' + k][j])
      row['start_unix'] = int(g['obs_start_unix'][j])
      row['sunrise_h_rel'] = int(g['obs_sunrise_h'][j, i] - g['obs_start_unix'][j]); row['sunset_rel'] = int(g['obs_sunset'][j, i] - g['obs_start_unix'][j])
      sim.set_state(rows_to_arrays([row]))
      fu, fv = oracle.wind_forecast(field, [row['x']], [row['y']], [row['pressure']], [row['time_elapsed_s']])
      noise = np.array([[g['obs_wind_measured'][j, i, 0] - fu[0], g['obs_wind_measured'][j, i, 1] - fv[0]]], np.float32)
      got[0, i] = sim.observe(torch.from_numpy(noise).cuda()).cpu().numpy()[0]
      sim.check_errors()
      fo.observe(row32(row), noise[0].astype(np.float64))
      same[0, i] = fo.features()
    check(got, same, f'F16 vehicle {j} vs the oracle on the same float32 inputs')
    sens = np.abs(same.astype(np.float64) - g['obs_features'][j][None].astype(np.float64))
    check(got, g['obs_features'][j][None], f'F16 vehicle {j} vs the reference', slack=sens)


def test_reference_style_objects_carry_the_vehicle(ble):
  """The reference-shaped mirrors with non-default vehicles, as a user of the reference writes them: BalloonState(...vehicle fields...),
  stable_init.cold_start_to_stable_params / calculate_stable_params_for_pressure, Balloon.simulate_step, calculate_superpressure_and_volume,
  thermal.d_balloon_temperature_dt(balloon_mass), test_helpers.create_balloon(power_safety_layer_enabled=False), and an arena whose balloon
  state is replaced by such a state -- F16's first steps, its cold starts and the oracle are the references."""
  import datetime as dt
  from balloon_learning_environment_amd.env import simulator_data, wind_field
  from balloon_learning_environment_amd.env.balloon import balloon, control, stable_init, thermal
  from balloon_learning_environment_amd.utils import test_helpers, units
  d = golden('f16_vehicles')
  for vi in range(len(d['vehicles'])):
    veh = helpers.fixture_vehicle(d, vi)
    j = int(np.nonzero(d['vehicle_index'] == vi)[0][0])
    atm = simulator_data.Atmosphere(float(d['alpha'][j]))
    kw = balloon._vehicle_kwargs(veh)
    start = units.datetime_from_timestamp(int(d['start_unix'][j]))
    st = balloon.BalloonState(center_latlng=balloon.LatLng.from_degrees(float(d['center_lat_deg'][j]), float(d['center_lng_deg'][j])), date_time=start,
                              x=units.Distance(m=float(d['x'][j, 0])), y=units.Distance(m=float(d['y'][j, 0])), pressure=float(d['pressure'][j, 0]),
                              upwelling_infrared=float(d['upwelling_infrared'][j]), **kw)
    assert balloon.vehicle_of(st) == {**ble._abi.VEHICLE_DEFAULTS, **veh, 'power_safety_layer_enabled': bool(veh.get('power_safety_layer_enabled', 1))}
    stable_init.cold_start_to_stable_params(st, atm)
    for k in ('ambient_temperature', 'internal_temperature', 'mols_air', 'envelope_volume', 'superpressure'):
      assert rel_err(np.array([getattr(st, k)]), d['cold_' + k][j:j + 1], FLOORS[k]).max() <= RTOL, (vi, k)
    sp = stable_init.calculate_stable_params_for_pressure(st.pressure, st.envelope_volume_base, st.envelope_volume_dv_pressure, st.envelope_mass,
                                                          st.payload_mass, st.mols_lift_gas, st.latlng, st.date_time, st.upwelling_infrared, atm)
    assert abs(sp.mols_air - st.mols_air) <= 1e-3 * max(1.0, st.mols_air) and abs(sp.superpressure - st.superpressure) <= 1.0
    vol, spr = balloon.calculate_superpressure_and_volume(st.mols_lift_gas, st.mols_air, st.internal_temperature, st.pressure,
                                                          st.envelope_volume_base, st.envelope_volume_dv_pressure)
    assert abs(vol - st.envelope_volume) <= 1e-5 * st.envelope_volume and abs(spr - st.superpressure) <= 0.2
    # thermal.d_balloon_temperature_dt is inversely proportional to its balloon_mass argument (thermal.py:221-230)
    a = thermal.d_balloon_temperature_dt(1800.0, 68.5, 210.0, 215.0, 8000.0, 40.0, 1360.0, 260.0)
    b = thermal.d_balloon_temperature_dt(1800.0, st.envelope_mass, 210.0, 215.0, 8000.0, 40.0, 1360.0, 260.0)
    assert b == pytest.approx(a * 68.5 / st.envelope_mass, rel=1e-5)
    # the fixture's first three steps, free-running from ITS initial state (battery as the fixture set it)
    st.battery_charge = units.Energy(watt_hours=float(d['battery_charge'][j, 0]))
    bal = balloon.Balloon(st)
    for s in range(3):
      w = wind_field.WindVector(units.Velocity(mps=float(np.float32(d['wind_uv'][j, s, 0]))), units.Velocity(mps=float(np.float32(d['wind_uv'][j, s, 1]))))
      bal.simulate_step(w, atm, control.AltitudeControlCommand(int(d['actions'][j, s])), dt.timedelta(minutes=3))
    for k, ref in (('pressure', d['pressure'][j, 3]), ('superpressure', d['superpressure'][j, 3]), ('internal_temperature', d['internal_temperature'][j, 3])):
      assert abs(getattr(bal.state, k) - ref) <= 2e-4 * max(abs(ref), FLOORS[k]), (vi, k, getattr(bal.state, k), ref)
    assert bal.state.battery_charge.watt_hours == pytest.approx(d['battery_charge'][j, 3], rel=1e-4)
    assert bal.state.envelope_mass == st.envelope_mass and bal.state.power_safety_layer_enabled == bool(veh.get('power_safety_layer_enabled', 1))
  # create_balloon(power_safety_layer_enabled=False): at night with an almost empty battery the layer would pause DOWN into STAY
  atm = simulator_data.Atmosphere(0.5)
  night = units.datetime(2021, 9, 9, 0)
  eff = {}
  for enabled in (True, False):
    b = test_helpers.create_balloon(date_time=night, power_percent=0.01, power_safety_layer_enabled=enabled, atmosphere=atm)
    b.simulate_step(wind_field.WindVector(units.Velocity(mps=1.0), units.Velocity(mps=1.0)), atm, control.AltitudeControlCommand.DOWN, dt.timedelta(minutes=3))
    eff[enabled] = (b.state.acs_power.watts, b.state.power_safety_layer.navigation_is_paused)
  assert eff[True] == (0.0, True) and eff[False][0] > 0.0 and eff[False][1] is False
  # an arena handed such a state flies and observes that vehicle from then on
  from balloon_learning_environment_amd.env import balloon_env
  env = balloon_env.BalloonEnv(seed=5)
  state = env.arena.get_balloon_state()
  state.battery_capacity = units.Energy(watt_hours=2000.0); state.battery_charge = units.Energy(watt_hours=1500.0)
  env.arena.set_balloon_state(state)
  obs, _, _, _ = env.step(1)
  assert env.arena.get_balloon_state().battery_capacity.watt_hours == 2000.0
  assert obs[1] == pytest.approx(env.arena.get_balloon_state().battery_charge.watt_hours / 2000.0, abs=1e-6)      # battery_soc with ITS capacity


def test_checkpoint_carries_the_vehicle(ble):
  """state_dict() / load_state_dict() of a simulator that flies a non-default vehicle: the restored one flies on bit for bit (the vehicle is
  part of the checkpoint; a fresh simulator would otherwise continue as the default balloon)."""
  d = golden('f16_vehicles')
  veh = helpers.fixture_vehicle(d, 1)
  n, k = 2048, 5
  field = (np.random.default_rng(21).standard_normal((21, 21, 10, 9, 2)) * 5.0).astype(np.float32)
  acts = torch.from_numpy(np.random.default_rng(22).integers(0, 3, (2 * k, n)).astype(np.uint8)).cuda()
  a = ble.VecSimulator(n); a.set_vehicle(**veh); a.set_grid(field); a.reset_device(seed=4)
  r = torch.zeros(k, n, device='cuda'); t = torch.zeros(k, n, dtype=torch.uint8, device='cuda')
  a.step_n(acts[:k], r, t, noise_seed=3)
  ckpt = a.state_dict()
  assert ckpt['vehicle'] == a.vehicle and set(a.vehicle) == set(veh)
  b = ble.VecSimulator(n); b.set_grid(field)
  b.load_state_dict(ckpt)
  assert b.vehicle == a.vehicle
  rb = torch.zeros(k, n, device='cuda'); tb = torch.zeros(k, n, dtype=torch.uint8, device='cuda')
  a.step_n(acts[k:], r, t, noise_seed=3); b.step_n(acts[k:], rb, tb, noise_seed=3)
  torch.cuda.synchronize(); a.check_errors(); b.check_errors()
  assert torch.equal(r, rb) and torch.equal(t, tb)
  sa, sb = a.get_state(), b.get_state()
  for name in sa:
    np.testing.assert_array_equal(sa[name], sb[name], err_msg=name)
  c = ble.VecSimulator(n); c.set_grid(field); c.load_state_dict(dict(ckpt, vehicle={}))      # the same state flown as the default balloon goes elsewhere
  rc = torch.zeros(k, n, device='cuda'); tc = torch.zeros(k, n, dtype=torch.uint8, device='cuda')
  c.step_n(acts[k:], rc, tc, noise_seed=3); torch.cuda.synchronize()
  assert not np.array_equal(c.get_state()['pressure'], sa['pressure'])


def test_random_vehicles_every_env_matches_oracle(ble):
  """Eight vehicles drawn at random inside +-20 % of the reference's constants (every field varied at once, the power layer on or off), 2 048
  environments each: the vehicle's own cold start on the device (every environment against the oracle's), then two agent steps with random
  actions in a grid wind, every environment against the oracle from the device's own pre-step state -- 1e-5, discrete exact."""
  from balloon_learning_environment_amd import _abi
  rng = np.random.default_rng(2026)
  field = (rng.standard_normal((21, 21, 10, 9, 2)) * 6.0).astype(np.float32)
  n = 2048
  checked = 0
  for trial in range(8):
    veh = {k: float(v * rng.uniform(0.8, 1.2)) for k, v in _abi.VEHICLE_DEFAULTS.items() if k != 'power_safety_layer_enabled'}
    veh['power_safety_layer_enabled'] = int(trial % 2)
    sim = ble.VecSimulator(n)
    sim.set_vehicle(**veh)
    sim.set_grid(field)
    sim.reset_device(seed=100 + trial)
    torch.cuda.synchronize(); sim.check_errors()
    st = sim.get_state()
    out, err = oracle.stable_init(st['pressure'], st['center_lat_deg'], st['center_lng_deg'], st['x'], st['y'], st['start_unix'],
                                  st['upwelling_infrared'], st['alpha'], vehicle=veh)
    for key, v in out.items():
      assert rel_err(st[key], v, FLOORS[key]).max() <= RTOL, (trial, key)
    # the sampler's pressures suit the reference's vehicle: some of these start burst / deflated -- the transition must say so exactly like the oracle
    for step in range(2):
      before = sim.get_state()
      live = before['status'] == 0
      acts = rng.integers(0, 3, n).astype(np.uint8)
      reward, terminal = sim.step(torch.from_numpy(acts).cuda())
      torch.cuda.synchronize(); sim.check_errors()
      got = sim.get_state()
      o = oracle_state_from_abi({k: v[live] for k, v in before.items()})
      ro, to, eo, err = oracle.step(o, acts[live], field=field, vehicle=veh)
      assert err == 0
      compare_states({k: v[live] for k, v in got.items()}, o, ctx=f'vehicle trial {trial} step {step}')
      np.testing.assert_array_equal(sim.effective_action.cpu().numpy()[live], eo)
      np.testing.assert_array_equal(terminal.cpu().numpy()[live], to)
      np.testing.assert_allclose(reward.cpu().numpy()[live], ro, rtol=RTOL, atol=RTOL)
      checked += int(live.sum())
  assert checked > 20000


def test_random_vehicles_observation_matches_oracle(ble):
  """The observation of three random vehicles (as above), six environments each flown and observed for four steps: every 1099-vector
  against the feature oracle with the same vehicle on the device's own float32 state."""
  import features_oracle
  from balloon_learning_environment_amd import _abi
  from test_gpu_observe import check
  rng = np.random.default_rng(77)
  field = (rng.standard_normal((21, 21, 10, 9, 2)) * 6.0).astype(np.float32)
  n = 6
  compared = 0
  for trial in range(3):
    veh = {k: float(v * rng.uniform(0.85, 1.15)) for k, v in _abi.VEHICLE_DEFAULTS.items() if k != 'power_safety_layer_enabled'}
    veh['power_safety_layer_enabled'] = int(trial % 2)
    sim = ble.VecSimulator(n)
    sim.set_vehicle(**veh); sim.set_grid(field); sim.reset_device(seed=500 + trial)
    alpha = sim.state['alpha'].cpu().numpy().astype(np.float64)
    oracles = [features_oracle.FeatureOracle(field, alpha[j], vehicle=veh) for j in range(n)]
    for i in range(4):
      if i > 0:
        sim.step(torch.from_numpy(rng.integers(0, 3, n).astype(np.uint8)).cuda())
      noise = (rng.standard_normal((n, 2)) * 1.5).astype(np.float32)
      obs = sim.observe(torch.from_numpy(noise).cuda()).cpu().numpy()
      sim.check_errors()
      state = sim.get_state()
      for j in range(n):
        row = {k: float(state[k][j]) for k in STATE_FLOATS}
        for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha'):
          row[k] = float(state[k][j])
        for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s', 'start_unix'):
          row[k] = int(state[k][j])
        oracles[j].observe(row, noise[j].astype(np.float64))
        if state['status'][j] == 0:
          try:
            want = oracles[j].features()
          except ValueError:        # the reference raises ("no safe pressure") for a vehicle that cannot float anywhere in the band: the device flags it
            continue
          check(obs[j], want, f'vehicle trial {trial} env {j} step {i}')
          compared += 1
  assert compared >= 40, compared
