"""Pins the CPU oracle (oracle/ble_oracle.c) to the reference.

Two sources, both committed under tests/golden/:
  * *.npz  -- outputs of the reference's own Python (imported under container-type
              shims by tests/golden/make_golden.py), fp64;
  * reference_known_answers.json -- literals from the reference's own unit tests.
CPU only; runs everywhere.
"""
import numpy as np
import pytest

import oracle
import helpers
from helpers import (STATE_FLOATS, STATE_INTS, STATE_U8, golden, known_answers, traj_state_at, unix)

KA = known_answers()
RT = 1e-12  # function-level agreement with the reference (fp64 vs fp64)


def close(a, b, rtol=RT, atol=0.0):
  np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ---------------------------------------------------------------- atmosphere (F1)
def test_f1_atmosphere_tables_and_lookups():
  d = golden('f1_atmosphere')
  for i, a in enumerate(d['alphas']):
    lapse, ttr, ptr = oracle.atm_tables(float(a))
    close(lapse, d['lapse'][i]); close(ttr, d['temperature_transitions'][i]); close(ptr, d['pressure_transitions'][i])
    h, t, rho, err = oracle.at_pressure(float(a), d['pressures'])
    assert err == 0
    close(h, d['h_of_p'][i], rtol=1e-11); close(t, d['t_of_p'][i]); close(rho, d['rho_of_p'][i])
    p, t2, _, err = oracle.at_height(float(a), d['heights'])
    assert err == 0
    close(p, d['p_of_h'][i]); close(t2, d['t_of_h'][i])


# The reference asserts these ranges over alphas drawn from 10 PRNG keys (not the extremes 0 and 1).
@pytest.mark.parametrize('alpha', [0.15, 0.37, 0.5, 0.85])
def test_atmosphere_reference_test_ranges(alpha):
  for h, pr, tr, dr in KA['atmosphere_at_height']['cases']:
    p, t, rho, err = oracle.at_height(alpha, [h])
    assert err == 0 and pr[0] <= p[0] <= pr[1] and tr[0] <= t[0] <= tr[1] and dr[0] <= rho[0] <= dr[1]
  for p, hr, tr, dr in KA['atmosphere_at_pressure']['cases']:
    h, t, rho, err = oracle.at_pressure(alpha, [p])
    assert err == 0 and hr[0] <= h[0] <= hr[1] and tr[0] <= t[0] <= tr[1] and dr[0] <= rho[0] <= dr[1]
  for h in KA['atmosphere_out_of_range']['heights_raise']:
    assert oracle.at_height(alpha, [h])[3] & oracle.ERR_PRESSURE_RANGE
  for p in KA['atmosphere_out_of_range']['pressures_raise']:
    assert oracle.at_pressure(alpha, [p])[3] & oracle.ERR_PRESSURE_RANGE


# ---------------------------------------------------------------- solar (F2)
def test_f2_solar_calculator():
  d = golden('f2_solar')
  el, az, fl, err = oracle.solar_calculator(d['lat_rad'], d['lng_rad'], d['unix_s'])
  assert err == 0
  close(el, d['el'], rtol=0, atol=2e-11); close(fl, d['flux']); close(az, d['az'], rtol=0, atol=1e-7)


def test_f2_attenuation_power_shadow_latlng():
  d = golden('f2_solar')
  att, err = oracle.solar_attenuation(d['att_el'].ravel(), d['att_p'].ravel())
  assert err == 0
  close(att, d['attenuation'].ravel())
  pw, _ = oracle.solar_power(d['att_el'].ravel(), d['att_p'].ravel())
  close(pw, d['power'].ravel(), atol=1e-12)
  np.testing.assert_array_equal(oracle.balloon_shadow(d['shadow_el'], np.full(d['shadow_el'].size, 3.3)), d['shadow33'])
  np.testing.assert_array_equal(oracle.balloon_shadow(d['shadow_el'], np.full(d['shadow_el'].size, 2.7)), d['shadow27'])
  lat, lng = oracle.latlng_from_offset(d['off_lat0'], d['off_lng0'], d['off_x'], d['off_y'])
  close(lat, d['off_lat'], rtol=0, atol=1e-15); close(lng, d['off_lng'], rtol=0, atol=1e-15)


def test_solar_reference_known_answers():
  for lat, lng, ts, el_e, az_e, fl_e in KA['solar_calculator']['cases']:
    el, az, fl, err = oracle.solar_calculator(np.radians(lat), np.radians(lng), ts)
    assert err == 0 and abs(el[0] - el_e) < 0.05
    if az_e is not None:
      assert abs(az[0] - az_e) < 0.05
    if fl_e is not None:
      assert abs(fl[0] - fl_e) < 0.05
  for el, p, e in KA['solar_attenuation']['cases']:
    assert abs(oracle.solar_attenuation(el, p)[0][0] - e) < 5e-6
  for el in KA['solar_attenuation_raises']['el_raise']:
    assert oracle.solar_attenuation(el, 5000.0)[1] & oracle.ERR_SOLAR_RANGE
  for p in KA['solar_attenuation_raises']['pressure_raise']:
    assert oracle.solar_attenuation(0.0, p)[1] & oracle.ERR_SOLAR_RANGE
  for el, h, e in KA['balloon_shadow']['cases']:
    assert abs(oracle.balloon_shadow(el, h)[0] - e) < 5e-4
  f = KA['features_solar_elevation']
  el, _, _, _ = oracle.solar_calculator(np.radians(f['lat']), np.radians(f['lng']), unix(f['time']))
  assert abs(el[0] - f['el_deg']) < 1e-12
  b = KA['appendix_b']
  el, az, fl, _ = oracle.solar_calculator(np.radians(b['solar'][0]), np.radians(b['solar'][1]), b['solar'][2])
  close([el[0], az[0], fl[0]], b['solar'][3:], rtol=1e-13)
  assert abs(oracle.solar_attenuation(30.0, 20000.0)[0][0] - b['attenuation_30_20000']) < 1e-15


def test_sunrise_sunset_reference_known_answers():
  for now, sr, ss in KA['sunrise_sunset']['cases']:
    a, b = oracle.next_sunrise_sunset(0.0, 0.0, unix(now))
    assert a[0] == unix(sr) and b[0] == unix(ss)


def test_spherical_geometry_reference_known_answers():
  g = KA['spherical_geometry']
  for lat, lng in g['one_degree_lat']:
    la, _ = oracle.latlng_from_offset(np.radians(lat), np.radians(lng), 0.0, 111000.0)
    assert abs(np.degrees(la[0]) - (lat + 1.0)) < 5e-3
  for lng, deg, expected in g['lng_wrap']:
    _, lo = oracle.latlng_from_offset(0.0, np.radians(lng), 6371000.0 * np.radians(deg), 0.0)
    assert abs(np.degrees(lo[0]) - expected) < 1e-7


# ---------------------------------------------------------------- thermal / volume / ACS (F3-F5)
def test_f3_thermal():
  d = golden('f3_thermal')
  out, err = oracle.thermal_dtdt(d['volume'], d['t_int'], d['t_amb'], d['pressure'], d['el'], d['flux'], d['ir'])
  assert err == 0
  close(out, d['dtdt'], rtol=1e-11, atol=1e-16)


def test_f4_superpressure_volume():
  d = golden('f4_sp_volume')
  vol, sp = oracle.sp_volume(d['mols_air'], d['t_int'], d['pressure'])
  close(vol, d['volume']); close(sp, d['superpressure'], rtol=1e-10, atol=1e-9)
  assert (d['superpressure'] == 0).sum() > 8  # the not-fully-inflated branch is exercised
  b = KA['appendix_b']['sp_volume']
  vol, sp = oracle.sp_volume(b[1], b[2], b[3])
  close([vol[0], sp[0]], b[4:], rtol=1e-13)


def test_f5_acs_and_power_table():
  d = golden('f5_acs_power_table')
  power, eff, mdot = oracle.acs(d['pr'])
  close(power, d['power']); close(eff, d['eff'], atol=1e-15); close(mdot, d['mass_flow'], atol=1e-17)
  close(oracle.acs_efficiency(d['pr2'], d['power2']), d['eff2'], atol=1e-15)
  w, err = oracle.power_table(d['pt_pr'], d['pt_soc'])
  assert err == 0
  np.testing.assert_array_equal(w, d['pt_watts'])


def test_acs_power_table_reference_known_answers():
  a = KA['acs']
  for pr, w in a['power_eq']:
    assert oracle.acs(pr)[0][0] == w
  for pr, w in a['power_le']:
    assert oracle.acs(pr)[0][0] <= w
  for pr, w in a['power_ge']:
    assert oracle.acs(pr)[0][0] >= w
  for pr, w, e in a['eff_eq']:
    assert oracle.acs_efficiency(pr, w)[0] == pytest.approx(e, abs=1e-15)
  for pr, w, e in a['eff_ge']:
    assert oracle.acs_efficiency(pr, w)[0] >= e - 1e-15
  for pr, w, e in a['eff_le']:
    assert oracle.acs_efficiency(pr, w)[0] <= e + 1e-15
  for pr, w in KA['appendix_b']['acs_power']:
    assert oracle.acs(pr)[0][0] == pytest.approx(w, rel=1e-13)
  t = KA['power_table']
  for pr, soc, w in t['cases']:
    got, err = oracle.power_table(pr, soc)
    assert err == 0 and got[0] == w
  for pr in t['raises']:
    assert oracle.power_table(pr, 1.0)[1] & oracle.ERR_POWER_TABLE


# ---------------------------------------------------------------- safety layers (F6)
def test_f6_envelope_altitude_power_traces():
  d = golden('f6_safety')
  oa, of = oracle.envelope_safety_trace(d['env_action'], d['env_sp'])
  np.testing.assert_array_equal(oa, d['env_out_action']); np.testing.assert_array_equal(of, d['env_out_fsm'])
  assert set(np.unique(of)) == {0, 1, 2, 3, 4}
  for tag in 'ab':
    oa, of, err = oracle.altitude_safety_trace(float(d[f'alt_{tag}_alpha']), d[f'alt_{tag}_action'], d[f'alt_{tag}_p'])
    assert err == 0
    np.testing.assert_array_equal(oa, d[f'alt_{tag}_out_action']); np.testing.assert_array_equal(of, d[f'alt_{tag}_out_fsm'])
    assert set(np.unique(of)) == {0, 1, 2}
  for k in range(3):
    oa, osr, oss, op = oracle.power_safety_trace(d[f'pow_{k}_action'], d[f'pow_{k}_now'], d[f'pow_{k}_batt'],
                                                 d[f'pow_{k}_sunrise_h0'], d[f'pow_{k}_sunset0'])
    np.testing.assert_array_equal(oa, d[f'pow_{k}_out_action']); np.testing.assert_array_equal(op, d[f'pow_{k}_out_paused'])
    np.testing.assert_array_equal(osr, d[f'pow_{k}_out_sunrise_h']); np.testing.assert_array_equal(oss, d[f'pow_{k}_out_sunset'])
    # PowerSafetyLayer.__init__ (sunrise search) agrees too
    sr, ss = oracle.next_sunrise_sunset(np.radians(float(d[f'pow_{k}_lat'])), np.radians(float(d[f'pow_{k}_lng'])),
                                        int(d[f'pow_{k}_now'][0]))
    assert sr[0] + 1800 == d[f'pow_{k}_sunrise_h0'] and ss[0] == d[f'pow_{k}_sunset0']
    assert op.max() == 1 and op.min() == 0


def test_safety_reference_known_answers():
  for sp, a, e in KA['envelope_safety']['cases']:
    assert oracle.envelope_safety_trace([a], [sp])[0][0] == e
  alt = KA['altitude_safety']
  for alpha in (0.0, 0.6, 1.0):
    pr = {k: oracle.at_height(alpha, [v * 0.3048])[0][0] for k, v in alt['altitudes_ft'].items()}
    for name, a, e in alt['action_cases']:
      assert oracle.altitude_safety_trace(alpha, [a], [pr[name]])[0][0] == e
    for name, paused in alt['paused_cases']:
      assert (oracle.altitude_safety_trace(alpha, [0], [pr[name]])[1][0] != 0) == paused
    for seq, paused in alt['hysteresis']:
      assert (oracle.altitude_safety_trace(alpha, [0, 0], [pr[s] for s in seq])[1][-1] != 0) == paused
  for c in KA['power_safety']['cases']:
    t0 = unix(c['start'])
    sr, ss = oracle.next_sunrise_sunset(0.0, 0.0, t0)
    oa, _, _, _ = oracle.power_safety_trace([0], [t0], [c['batt_wh']], sr[0] + 1800, ss[0], 0, c['load_w'], c['cap_wh'])
    assert oa[0] == c['expected']
  c = KA['power_safety']['sunrise_comment']
  assert oracle.next_sunrise_sunset(0.0, 0.0, unix(c['start']))[0][0] == unix(c['sunrise'])


# ---------------------------------------------------------------- wind (F7)
def test_f7_wind_interpolation():
  d = golden('f7_wind')
  u, v = oracle.wind_forecast(d['field'], d['x'], d['y'], d['pressure'], d['elapsed_s'])
  close(u, d['u'], rtol=1e-13, atol=1e-14); close(v, d['v'], rtol=1e-13, atol=1e-14)
  field = np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)).astype(np.float32)
  q = np.array(KA['wind_grid_spot']['query_f32'], np.float32).astype(np.float64)
  u, v = oracle.wind_forecast(field, q[0] * 1000.0, q[1] * 1000.0, q[2], int(round(q[3] * 3600)))
  # (the elapsed time is rounded to a whole second here, hence the looser bound)
  assert abs(u[0] - KA['wind_grid_spot']['uv'][0]) < 2e-4 and abs(v[0] - KA['wind_grid_spot']['uv'][1]) < 2e-4


def test_wind_properties_of_reference_tests():
  """Properties asserted by env/grid_based_wind_field_test.py:86-234 (linearity, clamp, boomerang)."""
  field = golden('f7_wind')['field']
  f = lambda x, y, p, t: np.array(oracle.wind_forecast(field, x, y, p, t)).ravel()
  a, b, m = f(0, 0, 9000, 0), f(50000, 0, 9000, 0), f(25000, 0, 9000, 0)
  close(m, 0.5 * (a + b), rtol=1e-12)
  a, b, m = f(0, 0, 9000, 0), f(0, 0, 10000, 0), f(0, 0, 9500, 0)
  close(m, 0.5 * (a + b), rtol=1e-12)
  a, b, m = f(0, 0, 9000, 6 * 3600), f(0, 0, 9000, 12 * 3600), f(0, 0, 9000, 9 * 3600)
  close(m, 0.5 * (a + b), rtol=1e-12)
  close(f(7e5, -9e5, 20000, 3600), f(5e5, -5e5, 14000, 3600))      # clamp beyond the grid
  close(f(1e4, 2e4, 7000, 50 * 3600), f(1e4, 2e4, 7000, 46 * 3600))  # boomerang 48+2 == 48-2
  close(f(1e4, 2e4, 7000, 98 * 3600), f(1e4, 2e4, 7000, 2 * 3600))   # second leg goes forward again


# ---------------------------------------------------------------- transition (F8, F9)
def _check_traj(d, use_field):
  n, steps = d['actions'].shape
  valid = d['valid'] if 'valid' in d.files else np.ones((n, steps), np.uint8)
  field = None
  if use_field:
    field = (np.random.default_rng(int(d['field_seed'])).standard_normal((21, 21, 10, 9, 2)) *
             float(d['field_scale'])).astype(np.float32)
  worst = 0.0
  for s in range(steps):
    rows = np.nonzero(valid[:, s])[0]
    if rows.size == 0:
      continue
    st = traj_state_at(d, s, rows)   # teacher forcing: start every step from the reference's state
    wind = None if use_field else d['wind_uv'][rows, s]
    reward, terminal, _, err = oracle.step(st, d['actions'][rows, s], field=field, wind_uv=wind)
    assert err == 0
    for k in STATE_FLOATS:
      ref = d[k][rows, s + 1]
      e = np.abs(st[k] - ref) / np.maximum(np.abs(ref), 1.0)
      worst = max(worst, e.max())
      np.testing.assert_allclose(st[k], ref, rtol=2e-9, atol=2e-9, err_msg=f'{k} step {s}')
    for k in STATE_INTS + STATE_U8:
      np.testing.assert_array_equal(st[k], d[k][rows, s + 1], err_msg=f'{k} step {s}')
    np.testing.assert_allclose(reward, d['reward'][rows, s], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(terminal, d['status'][rows, s + 1] != 0)
  return worst


def test_f8_simulate_step_trajectories_teacher_forced():
  d = golden('f8_trajectories')
  worst = _check_traj(d, use_field=False)
  assert worst < 2e-9
  # the fixture covers every terminal status and every safety state
  assert set(np.unique(d['status'])) == {0, 1, 2, 3}
  assert d['alt_fsm'].max() == 2 and d['env_fsm'].max() == 4 and d['power_paused'].max() == 1


def test_f8_free_running_trajectories():
  """Oracle run open-loop for 40 agent steps from the reference's initial state."""
  d = golden('f8_trajectories')
  n, steps = d['actions'].shape
  st = traj_state_at(d, 0)
  for s in range(steps):
    live = st['status'] == 0
    assert np.array_equal(live, d['valid'][:, s] == 1)
    reward, terminal, _, _ = oracle.step(st, d['actions'][:, s], wind_uv=d['wind_uv'][:, s])
    for k in STATE_FLOATS:
      np.testing.assert_allclose(st[k], d[k][:, s + 1], rtol=1e-7, atol=1e-7, err_msg=f'{k} step {s}')
    for k in STATE_U8 + STATE_INTS:
      np.testing.assert_array_equal(st[k], d[k][:, s + 1], err_msg=f'{k} step {s}')
    np.testing.assert_allclose(reward[live], d['reward'][live, s], rtol=1e-9)


def test_f16_non_default_vehicles_teacher_forced():
  """F16: the reference's Balloon.simulate_step on BalloonStates whose flight-vehicle constants (balloon.py:156-173,183) differ
  from the defaults, two of them with power_safety_layer_enabled=False (:200,305), each after ITS OWN cold start
  (stable_init.py:132-157) -- the oracle with the same vehicle struct."""
  d = golden('f16_vehicles')
  assert len(d['vehicles']) >= 3 and d['actions'].shape[1] >= 40
  n, steps = d['actions'].shape
  worst = 0.0
  for vi in range(len(d['vehicles'])):
    veh = helpers.fixture_vehicle(d, vi)
    assert veh, 'every F16 vehicle differs from the default one'
    mine = np.nonzero(d['vehicle_index'] == vi)[0]
    # the vehicle's own cold start
    out, err = oracle.stable_init(d['pressure'][mine, 0], d['center_lat_deg'][mine], d['center_lng_deg'][mine], d['x'][mine, 0], d['y'][mine, 0],
                                  d['start_unix'][mine], d['upwelling_infrared'][mine], d['alpha'][mine], vehicle=veh)
    assert err == 0
    for k, v in out.items():
      np.testing.assert_allclose(v, d['cold_' + k][mine], rtol=1e-9, atol=1e-9, err_msg=f'vehicle {vi} cold start {k}')
    for s in range(steps):
      rows = mine[d['valid'][mine, s] == 1]
      if rows.size == 0:
        continue
      st = traj_state_at(d, s, rows)
      reward, terminal, _, err = oracle.step(st, d['actions'][rows, s], wind_uv=d['wind_uv'][rows, s], vehicle=veh)
      assert err == 0
      for k in STATE_FLOATS:
        ref = d[k][rows, s + 1]
        worst = max(worst, (np.abs(st[k] - ref) / np.maximum(np.abs(ref), 1.0)).max())
        np.testing.assert_allclose(st[k], ref, rtol=2e-9, atol=2e-9, err_msg=f'vehicle {vi} {k} step {s}')
      for k in STATE_INTS + STATE_U8:
        np.testing.assert_array_equal(st[k], d[k][rows, s + 1], err_msg=f'vehicle {vi} {k} step {s}')
      np.testing.assert_allclose(reward, d['reward'][rows, s], rtol=1e-12, atol=1e-15)
      np.testing.assert_array_equal(terminal, d['status'][rows, s + 1] != 0)
  assert worst < 2e-9
  # the switch matters in the fixture: with the layer off nobody is ever paused, with it on somebody is; an episode ends out of power
  off = np.isin(d['vehicle_index'], [vi for vi in range(len(d['vehicles'])) if not helpers.fixture_vehicle(d, vi).get('power_safety_layer_enabled', 1)])
  assert off.any() and d['power_paused'][off].max() == 0 and d['power_paused'][~off].max() == 1 and (d['status'][off, -1] == 1).any()
  # ... and so do the constants: flown as the DEFAULT vehicle the same states go elsewhere
  rows = np.nonzero(d['valid'][:, 0])[0]
  st = traj_state_at(d, 0, rows)
  oracle.step(st, d['actions'][rows, 0], wind_uv=d['wind_uv'][rows, 0])
  assert np.abs(st['pressure'] - d['pressure'][rows, 1]).max() > 1.0


def test_f16_observation_of_non_default_vehicles():
  """F16's observation part: the reference's PerciatelliFeatureConstructor on BalloonStates of other vehicles (battery_soc and
  excess_energy read the capacity and the daytime load, get_pressure_range the envelope / masses / lift gas / maximum
  superpressure) against the feature oracle with the same vehicle."""
  import features_oracle
  g = golden('f16_vehicles')
  field = helpers.fixture_field(g)
  worst = 0.0
  for j in range(len(g['vehicles'])):
    veh = helpers.fixture_vehicle(g, j)
    fo = features_oracle.FeatureOracle(field, g['obs_alpha'][j], vehicle=veh)
    plain = features_oracle.FeatureOracle(field, g['obs_alpha'][j])
    for i in range(g['obs_features'].shape[1]):
      row = {k: float(g['obs_' + k][j, i]) for k in STATE_FLOATS}
      for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s'):
        row[k] = int(g['obs_' + k][j, i])
      for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha'):
        row[k] = float(g['obs_' + k][j])
      row['start_unix'] = int(g['obs_start_unix'][j])
      fu, fv = oracle.wind_forecast(field, row['x'], row['y'], row['pressure'], row['time_elapsed_s'])
      err = (g['obs_wind_measured'][j, i, 0] - fu[0], g['obs_wind_measured'][j, i, 1] - fv[0])
      fo.observe(row, err); plain.observe(row, err)
      got = fo.features()
      worst = max(worst, float(np.abs(got.astype(np.float64) - g['obs_features'][j, i]).max()))
    if 'power_safety_layer_enabled' not in veh or len(veh) > 1:      # (a vehicle that differs only by the switch observes like the default one)
      assert np.abs(plain.features().astype(np.float64) - g['obs_features'][j, -1]).max() > 1e-3
  assert worst <= 2e-6, worst


def test_f17_features_over_a_forecast_that_is_not_a_grid():
  """F17: the reference's PerciatelliFeatureConstructor over its unit-test wind field, SimpleStaticWindField (a step function of
  pressure, wind_field.py:149-184) -- the feature oracle with that forecast's own column."""
  import features_oracle
  g = golden('f17_static_wind_features')
  worst = 0.0
  for j in range(g['features'].shape[0]):
    fo = features_oracle.FeatureOracle(None, g['alpha'][j], forecast_column=features_oracle.simple_static_wind_column)
    for i in range(g['features'].shape[1]):
      row = helpers.feature_row(g, j, i)
      fu, fv = features_oracle.simple_static_wind_column(row['x'], row['y'], [row['pressure']], row['time_elapsed_s'])
      assert (fu[0], fv[0]) == tuple(g['forecast_at_balloon'][j, i])
      fo.observe(row, (g['wind_measured'][j, i, 0] - fu[0], g['wind_measured'][j, i, 1] - fv[0]))
      worst = max(worst, float(np.abs(fo.features().astype(np.float64) - g['features'][j, i]).max()))
  assert worst <= 2e-6, worst
  assert len({tuple(v) for v in g['forecast_at_balloon'].reshape(-1, 2)}) >= 2      # the balloons cross a sheet boundary


def test_f9_arena_step_with_grid_wind_field():
  d = golden('f9_arena')
  assert _check_traj(d, use_field=True) < 2e-9


def test_appendix_b_trajectory():
  t = KA['appendix_b']['trajectory50']
  st = oracle.new_state(1)
  su = unix(t['start'])
  init, err = oracle.stable_init([t['pressure']], [0.0], [0.0], [0.0], [0.0], [su], [250.0], [t['alpha']])
  assert err == 0
  for k, v in init.items():
    st[k][:] = v
  st['pressure'][:] = t['pressure']; st['alpha'][:] = t['alpha']; st['upwelling_infrared'][:] = 250.0
  st['battery_charge'][:] = 2905.6; st['start_unix'][:] = su; st['last_command'][:] = 1
  sr, ss = oracle.next_sunrise_sunset(0.0, 0.0, su)
  st['sunrise_h'][:] = sr + 1800; st['sunset'][:] = ss
  for i in range(50):
    oracle.step(st, [i % 3], wind_uv=np.array([t['wind']]))
  got = [st[k][0] for k in ('x', 'y', 'pressure', 'internal_temperature', 'superpressure', 'mols_air', 'battery_charge')]
  np.testing.assert_allclose(got, [t['x'], t['y'], t['p'], t['t_int'], t['sp'], t['mols_air'], t['batt']], rtol=1e-9)
  assert st['status'][0] == 0


def test_pressure_ratio_and_reward_known_answers():
  r = KA['reward']
  noon = unix('2013-03-25T12:00:00')
  for x_km, y_km in r['in_radius_one']:
    assert oracle.reward_only(x_km * 1e3, y_km * 1e3, 8000.0, 2900.0, 0.0, 1, 0.0, 0.0, noon, 0) == 1.0
  j = r['just_outside_dropoff']
  d_m = (j['radius_km'] + j['extra_km']) * 1e3
  got = oracle.reward_only(d_m * np.cos(j['angle']), d_m * np.sin(j['angle']), 8000.0, 2900.0, 0.0, 1, 0.0, 0.0, noon, 0)
  assert abs(got - j['dropoff']) < j['delta']
  # DOWN at night with no excess energy and acs_power <= 100 W: multiplier 0.95
  got = oracle.reward_only(0.0, 0.0, 8000.0, 1000.0, 100.0, 0, 0.0, 0.0, unix('2013-03-25T00:00:00'), 0)
  assert abs(got - r['power_regularization']['no_excess_down']) < 1e-12


# ---------------------------------------------------------------- reset path (F10)
def test_f10_stable_init_and_sunrise():
  d = golden('f10_reset')
  out, err = oracle.stable_init(d['pressure'], d['center_lat_deg'], d['center_lng_deg'], d['x'], d['y'], d['unix_s'],
                                d['upwelling_infrared'], d['alpha'])
  assert err == 0
  for k, v in out.items():
    np.testing.assert_allclose(v, d[k], rtol=1e-9, atol=1e-9, err_msg=k)
  sr, ss = oracle.next_sunrise_sunset(d['balloon_lat_rad'], d['balloon_lng_rad'], d['unix_s'])
  np.testing.assert_array_equal(sr, d['sunrise']); np.testing.assert_array_equal(ss, d['sunset'])


# ---------------------------------------------------------------------------------------------
# Observation path oracle (oracle/features_oracle.py) vs the reference's own feature vectors.
def _oracle_features(name, j, keep_last=None):
  import features_oracle
  g = helpers.golden(name)
  field = helpers.fixture_field(g)
  fo = features_oracle.FeatureOracle(field, g['alpha'][j])
  n = g['x'].shape[1]
  out = []
  for i in range(n):
    row = helpers.feature_row(g, j, i)
    fu, fv = oracle.wind_forecast(field, row['x'], row['y'], row['pressure'], row['time_elapsed_s'])
    fo.observe(row, (g['wind_measured'][j, i, 0] - fu[0], g['wind_measured'][j, i, 1] - fv[0]))
    if keep_last is None or i >= n - keep_last:
      out.append(fo.features())
  return np.array(out), g['features'][j]


@pytest.mark.parametrize('j', [0, 1, 2])
def test_feature_oracle_matches_reference(j):
  got, want = _oracle_features('f11_features', j)
  assert got.shape == want.shape
  np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


def test_feature_oracle_long_horizon():
  """136 observations: the 6 h window (120 observations) must drop the oldest ones."""
  got, want = _oracle_features('f12_features_long', 0, keep_last=16)
  np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


# ---------------------------------------------------------------------------------------------
# F13: BASELINE.json configs[0] in closed loop -- the reference's StationSeekerAgent flying the
# reference's arena loop for micro_eval's 960 steps (generated by tests/golden/make_golden.py).
def test_f13_station_seeker_restatement_reproduces_every_action():
  """oracle/station_seeker_oracle.py vs the reference agent on the reference's own observations."""
  import station_seeker_oracle as sso
  g = golden('f13_station_seeker')
  n = int(g['n_flown'])
  assert n == 960 and (g['status'][0] == 0).all()              # the controller keeps the balloon alive all episode
  feats, actions, levels = g['features'][0], g['actions'][0], g['levels']
  for i in range(n):
    assert sso.best_level(feats[i]) == levels[i], i
    assert sso.pick_action(feats[i]) == actions[i], i
  assert set(np.unique(actions)) == {0, 1, 2}


def test_f13_transition_oracle_teacher_forced():
  """The C oracle's transition on the closed-loop episode: ground-truth wind (forecast + noise) at the
  pre-step state, the agent's action, 960 steps."""
  d = golden('f13_station_seeker')
  field = (np.random.default_rng(int(d['field_seed'])).standard_normal((21, 21, 10, 9, 2)) * float(d['field_scale'])).astype(np.float32)
  n = int(d['n_flown'])
  st = traj_state_at(d, 0, np.arange(n) * 0)            # n copies of env 0 ...
  for k in STATE_FLOATS + STATE_INTS + STATE_U8:        # ... each at its own step of the episode
    st[k][:] = d[k][0, :n]
  reward, terminal, _, err = oracle.step(st, d['actions'][0, :n], field=field, noise_uv=d['noise_uv'][0, :n])
  assert err == 0
  for k in STATE_FLOATS:
    np.testing.assert_allclose(st[k], d[k][0, 1:n + 1], rtol=2e-9, atol=2e-9, err_msg=k)
  for k in STATE_INTS + STATE_U8:
    np.testing.assert_array_equal(st[k], d[k][0, 1:n + 1], err_msg=k)
  np.testing.assert_allclose(reward, d['reward'][0, :n], rtol=1e-12, atol=1e-15)
  # wind_measured really is forecast + noise at the recorded states
  fu, fv = oracle.wind_forecast(field, d['x'][0], d['y'][0], d['pressure'][0], d['time_elapsed_s'][0])
  np.testing.assert_allclose(d['wind_measured'][0, :, 0], fu + d['noise_uv'][0, :, 0], rtol=0, atol=1e-12)


def test_f13_feature_oracle_closed_loop():
  """oracle/features_oracle.py along the whole episode (the WindGP window slides 840 times): every
  100th vector and the last 20 against the reference's, and the restated agent's action from the
  ORACLE's observation equals the reference agent's on every step."""
  import features_oracle
  import station_seeker_oracle as sso
  g = helpers.golden('f13_station_seeker')
  field = helpers.fixture_field(g)
  fo = features_oracle.FeatureOracle(field, g['alpha'][0])
  n = int(g['n_flown'])
  for i in range(n + 1):
    row = helpers.feature_row(g, 0, i)
    fo.observe(row, tuple(g['noise_uv'][0, i]))
    if i < n or i % 100 == 0:
      f = fo.features()
      if i % 100 == 0 or i > n - 20:
        np.testing.assert_allclose(f, g['features'][0, i], rtol=0, atol=1e-6, err_msg=f'step {i}')
      if i < n:
        assert sso.pick_action(f.astype(np.float32)) == g['actions'][0, i], i


# ---------------------------------------------------------------- wind noise composition (F14) and VAE decoder (F15)
def test_f14_noise_composition_oracle_matches_reference():
  """oracle/noise_oracle.py::wind_noise == the reference's SimplexWindNoise.get_wind_noise (wind_field.py:187-218,
  simplex_wind_noise.py:82-211) on the recorded seeds / offsets, both around the same stand-in primitive: pins the harmonic
  tables, spacings, offsets, NOISE_MAGNITUDE and the variance adjustment.  (The primitive itself is not opensimplex 0.3.)"""
  import math
  import noise_oracle
  d = golden('f14_wind_noise')
  assert d['noise_magnitude'] == math.sqrt(1.02 / 0.0569) == noise_oracle.REFERENCE_MAGNITUDE     # simplex_wind_noise.py:76
  for e in range(d['x'].shape[0]):
    got = noise_oracle.wind_noise(d['x'][e], d['y'][e], d['pressure'][e], d['elapsed_s'][e], d['seeds'][e], d['offsets'][e],
                                  np.float64, magnitude=noise_oracle.REFERENCE_MAGNITUDE)
    np.testing.assert_allclose(got, d['noise'][e], rtol=0, atol=1e-12)
    # WindField.get_ground_truth = get_forecast + noise (wind_field.py:125-145)
    np.testing.assert_allclose(d['ground_truth'][e], d['forecast'][e] + d['noise'][e], rtol=0, atol=1e-12)
  assert np.abs(d['noise']).max() > 1.0 and 0 < d['seeds'].min() and d['seeds'].max() < 1634753849
  assert len(np.unique(d['seeds'])) == d['seeds'].size


def test_f14_kernel_source_host_build_matches_reference_composition():
  """The kernel's own noise source (csrc/ble_noise.h::wind_noise_cached, host build) fed the recorded seeds / offsets
  through its harmonic cache == the reference's output scaled by the ratio of the two variance normalisations
  (noise_oracle.MAGNITUDE_RATIO: the reference divides by the variance of opensimplex, 0.0569; the kernel by its own
  primitive's, 0.088392 -- both aim at 1.02 (m/s)^2).  Tolerance: 1e-5 + the float32-vs-float64 sensitivity of the noise to
  its coordinates, computed here with the oracle in both precisions (pressure / 66.553 has an ulp of 1.5e-5 in float32)."""
  import ctypes
  import noise_oracle
  from emul import emul
  d = golden('f14_wind_noise')
  worst = 0.0
  for e in range(d['x'].shape[0]):
    n = d['x'].shape[1]
    cache = np.ascontiguousarray(helpers.noise_cache_from_draws(d['seeds'][e], d['offsets'][e], n, seed=77, episode=e))
    xs, ys, ps = (np.ascontiguousarray(d[k][e], np.float32) for k in ('x', 'y', 'pressure'))
    ts = np.ascontiguousarray(d['elapsed_s'][e], np.int32); ep = np.full(n, e, np.uint32)
    out = np.zeros((n, 2), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    emul.lib().emul_wind_noise_cached(ctypes.c_int64(n), vp(xs), vp(ys), vp(ps), vp(ts), ctypes.c_uint64(77), vp(ep), vp(cache), vp(out))
    assert np.array_equal(cache, helpers.noise_cache_from_draws(d['seeds'][e], d['offsets'][e], n, seed=77, episode=e))   # key matched: not redrawn
    want = d['noise'][e] * noise_oracle.MAGNITUDE_RATIO
    o32 = noise_oracle.wind_noise(xs, ys, ps, ts, d['seeds'][e], d['offsets'][e], np.float32)
    o64 = noise_oracle.wind_noise(xs.astype(np.float64), ys.astype(np.float64), ps.astype(np.float64), ts, d['seeds'][e], d['offsets'][e], np.float64)
    # |o64 - want|: what rounding the fixture's float64 positions to the ABI's float32 does to the REFERENCE'S value;
    # |o32 - o64|: float32 coordinate arithmetic (the kernel's) against float64 on the same float32 inputs
    bound = 1e-5 + np.abs(o32 - o64) + np.abs(o64 - want)
    err = np.abs(out.astype(np.float64) - want)
    assert (err <= bound).all(), (err.max(), bound.max())
    assert np.abs(o64 - want).max() < 2e-4 and np.abs(o32 - o64).max() < 2e-4     # (and neither sensitivity is large)
    assert np.abs(out - o32).max() < 1e-5                             # kernel source == float32 oracle up to fma / op order
    worst = max(worst, err.max())
  print(f'F14 host build vs reference x ratio: worst {worst:.2e}')


def test_f15_decoder_oracle_matches_reference():
  """oracle/vae_oracle.py == the reference's Decoder.__call__ / GenerativeWindFieldSampler.sample_field (generative/vae.py:140-186,
  env/generative_wind_field.py:49-62) on synthetic weights: layer order, ReLUs, (7, 7, 90) reshape, roll / slice order, signs,
  (21, 21, 10, 9) reshape, stack axis.  The resize operator is the fixture's stated assumption."""
  import vae_oracle
  d = golden('f15_decoder')
  params = helpers.hashed_decoder_params(int(d['param_seed']))
  flow = vae_oracle.mlp(d['latents'], params)
  np.testing.assert_allclose(flow, d['flow'], rtol=1e-12, atol=1e-12)
  fields = vae_oracle.decode_flow(d['flow'])
  scale = np.abs(d['fields']).max()
  assert np.abs(fields - d['fields']).max() <= 2e-7 * scale           # the fixture stores float32
  # it would notice: u / v swapped, a sign, the transposed reshape
  assert np.abs(fields[..., ::-1] - d['fields']).max() > 0.1 * scale
  assert np.abs(vae_oracle.decode_flow(d['flow'].reshape(-1, 90, 49).transpose(0, 2, 1).reshape(-1, 4410)) - d['fields']).max() > 0.1 * scale


def test_f15_kernel_tail_host_build_matches_reference():
  """The decoder tail's own source (csrc/ble_decode.h, host build) on the fixture's MLP output == the reference's fields."""
  import ctypes
  from emul import emul
  d = golden('f15_decoder')
  flow = np.ascontiguousarray(d['flow'], np.float32)
  grid = np.empty((flow.shape[0], 21, 21, 10, 9, 2), np.float32)
  emul.lib().emul_decode_flow(ctypes.c_int64(flow.shape[0]), flow.ctypes.data_as(ctypes.c_void_p), grid.ctypes.data_as(ctypes.c_void_p))
  scale = np.abs(d['fields']).max()
  assert np.abs(grid - d['fields']).max() <= 2e-6 * scale
