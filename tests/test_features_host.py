"""Host observation path: PerciatelliFeatureConstructor / WindGP /
pressure_range_builder against golden vectors produced by the reference's own
PerciatelliFeatureConstructor (tests/golden/make_golden.py::f11_features).

CPU: the forecast is a test double backed by the oracle's wind interpolation, so the GP,
sunrise feature, pressure range and column layout are pinned without a GPU.  The GPU test in
test_gpu_parity.py repeats it with the device forecast kernel behind GridBasedWindField.
Tolerance: features are float32; 1e-6 absolute (measured: <= 1e-11, i.e. identical after the float32 store).
"""
import datetime as dt

import numpy as np
import pytest

import helpers
import oracle
from balloon_learning_environment_amd.env import features, simulator_data, wind_field
from balloon_learning_environment_amd.env.balloon import balloon
import features_host
import pressure_range_host as pressure_range_builder
from balloon_learning_environment_amd.utils import transforms, units

ATOL = 1e-6


class OracleForecast:
  """wind_field.WindField forecast methods on top of oracle.wind_forecast (test double)."""

  def __init__(self, field):
    self.field = field

  def get_forecast(self, x, y, pressure, elapsed_time):
    u, v = oracle.wind_forecast(self.field, x.m, y.m, pressure, int(elapsed_time.total_seconds()))
    return wind_field.WindVector(units.Velocity(mps=float(u[0])), units.Velocity(mps=float(v[0])))

  def get_forecast_column(self, x, y, pressures, elapsed_time):
    n = len(pressures)
    u, v = oracle.wind_forecast(self.field, np.full(n, x.m), np.full(n, y.m), np.asarray(pressures, np.float64),
                                np.full(n, int(elapsed_time.total_seconds()), np.int64))
    return [wind_field.WindVector(units.Velocity(mps=float(a)), units.Velocity(mps=float(b))) for a, b in zip(u, v)]


def state_at(g, j, i):
  row = {k: g[k][j, i] for k in helpers.STATE_FLOATS}
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s'):
    row[k] = g[k][j, i]
  for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha'):
    row[k] = g[k][j]
  row['start_unix'] = g['start_unix'][j]
  row['sunrise_h_rel'] = g['sunrise_h'][j, i] - g['start_unix'][j]
  row['sunset_rel'] = g['sunset'][j, i] - g['start_unix'][j]
  return balloon.state_from_row(row)


def run_constructor(g, forecast, j, n_steps=None):
  fc = features_host.PerciatelliFeatureConstructor(forecast, simulator_data.Atmosphere(float(g['alpha'][j])))
  n = g['features'].shape[1] if n_steps is None else n_steps
  out = np.zeros((n, 1099), np.float32)
  for i in range(n):
    meas = wind_field.WindVector(units.Velocity(mps=float(g['wind_measured'][j, i, 0])),
                                 units.Velocity(mps=float(g['wind_measured'][j, i, 1])))
    fc.observe(simulator_data.SimulatorObservation(state_at(g, j, i), meas))
    out[i] = fc.get_features()
  return out


def field_of(g):
  return (np.random.default_rng(int(g['field_seed'])).standard_normal((21, 21, 10, 9, 2)) * float(g['field_scale'])).astype(np.float32)


@pytest.mark.parametrize('j', [0, 1, 2])
def test_perciatelli_features_match_reference(j):
  g = helpers.golden('f11_features')
  got = run_constructor(g, OracleForecast(field_of(g)), j)
  want = g['features'][j]
  assert got.shape == want.shape
  # unreachable-level pattern must be identical (discrete)
  unreach = lambda f: (f[:, 16::3] == 0) & (f[:, 17::3] == 1) & (f[:, 18::3] == 1)
  np.testing.assert_array_equal(unreach(got), unreach(want))
  np.testing.assert_array_equal(got[:, 8:14], want[:, 8:14])          # one-hots / flags
  err = np.abs(got.astype(np.float64) - want.astype(np.float64))
  assert err.max() <= ATOL, (np.unravel_index(err.argmax(), err.shape), err.max())


def test_feature_vector_properties():
  g = helpers.golden('f11_features')
  got = run_constructor(g, OracleForecast(field_of(g)), 0, n_steps=3)
  fc = features_host.PerciatelliFeatureConstructor(OracleForecast(field_of(g)), simulator_data.Atmosphere(0.5))
  assert fc.num_features == 1099 and fc.observation_space.shape == (1099,)
  assert fc.observation_space.contains(got[2])
  named = features.NamedPerciatelliFeatures(got[2])
  assert named.num_pressure_levels == 361 and named.wind_column_center() == 180
  assert named.level_is_valid(180)
  np.testing.assert_allclose(named.balloon_pressure, g['pressure'][0, 2], rtol=1e-6)
  assert not named.level_is_valid(0)
  w = features.convert_wind_feature_to_real_wind(named.wind_at(180))
  assert 0 <= w.bearing <= np.pi + 1e-6 and w.magnitude >= 0


def test_wind_gp_empty_and_horizon():
  import wind_gp_host as wind_gp
  g = helpers.golden('f11_features')
  fc = OracleForecast(field_of(g))
  gp = wind_gp.WindGP(fc)
  x, y = units.Distance(km=3.0), units.Distance(km=-7.0)
  q = np.array([[x.m, y.m, p, 600.0] for p in (6000.0, 9000.0, 12000.0)])
  means, dev = gp.query_batch(q)
  col = fc.get_forecast_column(x, y, q[:, 2], dt.timedelta(seconds=600))
  np.testing.assert_allclose(means[:, 0], [c.u.mps for c in col])
  assert (dev == 0).all()
  # one observation with a +1 m/s error: queried at the same point the posterior mean error is
  # s^2 / (s^2 + noise) and the deviation is 1 - s^2/(s^2+noise)
  f = fc.get_forecast(x, y, 9000.0, dt.timedelta(seconds=600))
  gp.observe(x, y, 9000.0, dt.timedelta(seconds=600), wind_field.WindVector(f.u + units.Velocity(mps=1.0), f.v))
  (mu, mv), d = gp.query(x, y, 9000.0, dt.timedelta(seconds=600))
  s2 = 3.6 ** 2
  np.testing.assert_allclose(mu - f.u.mps, s2 / (s2 + 0.05), rtol=1e-12)
  np.testing.assert_allclose(mv - f.v.mps, 0.0, atol=1e-12)
  np.testing.assert_allclose(d, 1 - s2 / (s2 + 0.05), rtol=1e-9)
  # observations older than 6 h are dropped: only the prior remains
  (mu, _), d = gp.query(x, y, 9000.0, dt.timedelta(seconds=600 + 6 * 3600))
  f2 = fc.get_forecast(x, y, 9000.0, dt.timedelta(seconds=600 + 6 * 3600))
  np.testing.assert_allclose(mu, f2.u.mps, atol=1e-12)
  np.testing.assert_allclose(d, 1.0)


def test_power_table_host_known_answers():
  # power_table_test.py of the reference (tests/golden/reference_known_answers.json)
  ka = helpers.known_answers()['power_table']
  for pr, soc, want in ka['cases']:
    assert features_host.power_table_lookup(pr, soc) == want
  with pytest.raises(AssertionError):
    features_host.power_table_lookup(ka['raises'][0], 0.5)
  g = helpers.golden('f5_acs_power_table')
  for pr, soc, w in zip(g['pt_pr'], g['pt_soc'], g['pt_watts']):
    assert features_host.power_table_lookup(float(pr), float(soc)) == w


def test_transforms_known_answers():
  # transforms_test.py of the reference
  assert transforms.linear_rescale_with_extrapolation(5.0, 0.0, 10.0) == 0.5
  assert transforms.linear_rescale_with_extrapolation(15.0, 0.0, 10.0) == 1.5
  assert transforms.linear_rescale_with_saturation(15.0, 0.0, 10.0) == 1.0
  assert transforms.linear_rescale_with_saturation(-5.0, 0.0, 10.0) == 0.0
  assert transforms.undo_linear_rescale_with_extrapolation(0.5, 0.0, 10.0) == 5.0
  assert transforms.squash_to_unit_interval(1.0, 1.0) == 0.5
  np.testing.assert_allclose(transforms.undo_squash_to_unit_interval(0.5, 30.0), 30.0)
  with pytest.raises(ValueError):
    transforms.squash_to_unit_interval(-1.0, 1.0)
  with pytest.raises(ValueError):
    transforms.squash_to_unit_interval(1.0, 0.0)
  with pytest.raises(ValueError):
    transforms.linear_rescale_with_extrapolation(1.0, 2.0, 1.0)


def test_pressure_range_is_safe_band():
  g = helpers.golden('f11_features')
  st = state_at(g, 0, 0)
  r = pressure_range_builder.get_pressure_range(st, simulator_data.Atmosphere(float(g['alpha'][0])))
  assert 4000 < r.min_pressure < r.max_pressure < 20000
  with pytest.raises(ValueError):
    pressure_range_builder._x_crossing(2.0, 0.0, 1.0, 1.0, 0.5)
  assert pressure_range_builder._x_crossing(0.0, 0.0, 2.0, 4.0, 1.0) == 0.5


# ---------------------------------------------------------------------------------------------
# The reference's own feature tests (env/features_test.py:92-496, env/wind_gp_test.py:44-64),
# restated against this package's constructor with the reference's unit-test wind field.
# The reference fixes its atmosphere with jax.random.PRNGKey(38) (jax absent: alpha unknown),
# so index ranges that depend on alpha are asserted through the pressure range instead.
START = units.datetime(2013, 3, 25, 9, 25, 32)
INVALID = (0.0, 1.0, 1.0)
ALPHA = 0.5


def create_observation(pressure=9000.0, charge_percent=1.0, x_km=0.0, y_km=0.0, lat=0.0, lng=0.0,
                       last_command=1, datetime=START, navigation_is_paused=False):
  import reset_host
  from balloon_learning_environment_amd.env.balloon import control
  import math
  center = balloon.LatLng(lat, lng)
  st = balloon.BalloonState(center_latlng=center, date_time=datetime, x=units.Distance(km=x_km), y=units.Distance(km=y_km),
                            pressure=pressure)
  lat_r, lng_r = features_host._latlng_rad(st)
  sp = reset_host.stable_params(np.array([pressure]), lat_r, lng_r,
                                np.array([int(datetime.timestamp())], np.int64), np.array([250.0]),
                                reset_host.AtmosphereTables(np.array([ALPHA])))
  st.ambient_temperature = float(sp['ambient_temperature'][0]); st.internal_temperature = float(sp['internal_temperature'][0])
  st.mols_air = float(sp['mols_air'][0]); st.envelope_volume = float(sp['envelope_volume'][0])
  st.superpressure = float(sp['superpressure'][0])
  st.battery_charge = st.battery_capacity * charge_percent
  st.last_command = control.AltitudeControlCommand(last_command)
  st.power_safety_layer = balloon.SafetyLayerView(navigation_is_paused)
  wv = wind_field.SimpleStaticWindField().get_forecast(units.Distance(km=0.0), units.Distance(km=0.0), pressure, dt.timedelta())
  return simulator_data.SimulatorObservation(st, wv)


def make_features(**kw):
  fc = features_host.PerciatelliFeatureConstructor(wind_field.SimpleStaticWindField(), simulator_data.Atmosphere(ALPHA))
  obs = create_observation(**kw)
  fc.observe(obs)
  return fc.get_features(), fc, obs


def test_ref_make_features_and_range():
  v, fc, _ = make_features()
  assert isinstance(v, np.ndarray) and v.shape == (1099,)
  space = fc.observation_space
  assert (v >= space.low).all() and (v <= space.high).all()


def test_ref_invalid_range_is_padded_and_valid_winds():
  v, fc, obs = make_features()
  col = v[16:].reshape(-1, 3)
  # balloon at level 80: [0, 100) and [281, 361) are padding
  for i in list(range(100)) + list(range(281, 361)):
    assert tuple(col[i]) == INVALID
  valid = ~np.all(col == np.array(INVALID, np.float32), axis=1)
  assert valid.any() and valid[180]
  # SimpleStaticWindField blows at exactly 10 m/s: magnitude 10 / (10 + 30)
  assert (col[valid, 2] == 0.25).all()
  rng = pressure_range_builder.get_pressure_range(obs.balloon_observation, simulator_data.Atmosphere(ALPHA))
  levels = fc.pressure_levels
  want = np.zeros(361, bool)
  want[100:281] = (levels >= rng.min_pressure) & (levels <= rng.max_pressure)
  np.testing.assert_array_equal(valid, want)
  named = features.NamedPerciatelliFeatures(v)
  np.testing.assert_allclose(named.balloon_pressure, 9000.0, atol=1e-3)
  for i in range(361):
    assert named.level_is_valid(i) == bool(want[i])


def test_ref_extreme_pressures_pad_correctly():
  v, _, _ = make_features(pressure=5000.0)
  assert (v[16:16 + 3 * 179].reshape(-1, 3) == np.array(INVALID, np.float32)).all()
  v, _, _ = make_features(pressure=14000.0)
  assert (v[16 + 181 * 3:16 + 361 * 3].reshape(-1, 3) == np.array(INVALID, np.float32)).all()


def test_ref_unreachable_altitude_is_marked():
  import reset_host
  p = float(reset_host.AtmosphereTables(np.array([ALPHA])).at_height(reset_host.MIN_ALTITUDE_M)[0][0])
  v, fc, _ = make_features(pressure=p)
  col = v[16:].reshape(-1, 3)
  # the balloon's own level (always index 180) is reachable iff the nearest 50 Pa level does not
  # round to below the altitude floor (true for the reference's PRNGKey(38) atmosphere)
  own_level_pressure = fc.pressure_levels[fc._nearest_pressure_level(p)]
  assert (tuple(col[180]) != INVALID) == (own_level_pressure <= p)
  assert tuple(col[179]) != INVALID
  assert (col[181:] == np.array(INVALID, np.float32)).all()


@pytest.mark.parametrize('pressure,expected', [(14000.0, 1.0), (14100.0, 1.0), (5000.0, 0.0), (4900.0, 0.0), (9500.0, 0.5)])
def test_ref_pressure_feature(pressure, expected):
  assert abs(make_features(pressure=pressure)[0][0] - expected) < 1e-7


@pytest.mark.parametrize('charge', [1.0, 0.0, 0.32])
def test_ref_power_feature(charge):
  assert abs(make_features(charge_percent=charge)[0][1] - charge) < 1e-7


@pytest.mark.parametrize('lat,lng,when,low,high', [
    (0.0, 0.0, units.datetime(2022, 3, 20, 12, 7, 27), 0.99, 1.0),
    (0.0, 180.0, units.datetime(2022, 3, 20, 12, 7, 27), 0.0, 0.01),
    (0.0, 0.0, units.datetime(2022, 3, 20, 8, 32, 12), 0.6, 0.9)])
def test_ref_solar_angle_feature(lat, lng, when, low, high):
  assert low <= make_features(lat=lat, lng=lng, datetime=when)[0][2] <= high


@pytest.mark.parametrize('x_km,y_km,sin_h,cos_h', [(1.0, 0.0, -1.0, 0.0), (-1.0, 0.0, 1.0, 0.0), (0.0, -1.0, 0.0, 1.0), (0.0, 1.0, 0.0, -1.0)])
def test_ref_heading_features(x_km, y_km, sin_h, cos_h):
  v = make_features(x_km=x_km, y_km=y_km)[0]
  assert abs(v[5] - sin_h) < 1e-7 and abs(v[6] - cos_h) < 1e-7


@pytest.mark.parametrize('x_km,y_km,expected', [(0.0, 1.0, 1 / 251), (-3.67, 0.0, 3.67 / 253.67), (300.0, 400.0, 2 / 3), (0.0, 0.0, 0.0)])
def test_ref_distance_feature(x_km, y_km, expected):
  assert abs(make_features(x_km=x_km, y_km=y_km)[0][7] - expected) < 1e-7


@pytest.mark.parametrize('cmd', [0, 1, 2])
def test_ref_last_command_features(cmd):
  v = make_features(last_command=cmd)[0]
  assert (v[8], v[9], v[10]) == (float(cmd == 2), float(cmd == 1), float(cmd == 0))


@pytest.mark.parametrize('paused', [True, False])
def test_ref_navigation_paused_features(paused):
  v = make_features(navigation_is_paused=paused)[0]
  assert (v[11], v[12]) == (float(paused), float(not paused))


@pytest.mark.parametrize('when,charge,expected', [
    (units.datetime(2020, 6, 21, 12, 0, 0), 1.0, 1.0), (units.datetime(2020, 6, 21, 12, 0, 0), 0.5, 0.0),
    (units.datetime(2020, 6, 21, 0, 0, 0), 1.0, 0.0), (units.datetime(2020, 6, 21, 0, 0, 0), 0.1, 0.0)])
def test_ref_excess_energy_feature(when, charge, expected):
  assert make_features(datetime=when, charge_percent=charge)[0][13] == expected


def test_ref_acs_power_feature():
  assert 0.0 <= make_features()[0][14] <= 1.0
  assert make_features(pressure=5000.0)[0][14] > make_features(pressure=12000.0)[0][14]


def test_ref_compute_solar_angle():
  ka = helpers.known_answers()['features_solar_elevation']
  st = create_observation(pressure=5000.0).balloon_observation
  st.date_time = units.datetime(2013, 9, 21, 12, 0, 0); assert features_host.compute_solar_angle(st) > 80
  st.date_time = units.datetime(2013, 9, 21, 0, 0, 0); assert features_host.compute_solar_angle(st) < -80
  st.date_time = units.datetime(2013, 9, 21, 18, 0, 0)
  assert abs(features_host.compute_solar_angle(st) - ka['el_deg']) < 1e-7     # assertAlmostEqual: 7 places


def test_ref_wind_gp_cases():
  import wind_gp_host as wind_gp
  zero, t0 = units.Distance(m=0.0), dt.timedelta(seconds=0)
  model = wind_gp.WindGP(wind_field.SimpleStaticWindField())
  pre = model.query(zero, zero, 0.0, t0)
  model.observe(zero, zero, 0.0, t0, wind_field.WindVector(units.Velocity(mps=1.0), units.Velocity(mps=1.0)))
  post = model.query(zero, zero, 0.0, t0)
  assert abs(float(post[1]) - 0.003843) < 5e-4           # wind_gp_test.py:44-53 (places=3)
  near = model.query(units.Distance(km=0.05), zero, 0.0, t0)
  assert (pre[0] != near[0]).all()


def test_ref_simple_static_wind_field():
  # env/wind_field_test.py:33-70
  wf = wind_field.SimpleStaticWindField()
  zero, t0 = units.Distance(m=0.0), dt.timedelta()
  col = [wf.get_forecast(zero, zero, p, t0) for p in np.arange(5000.0, 14001.0, 500.0)]
  assert any(w.v.mps > 0 and w.u.mps == 0 for w in col) and any(w.v.mps < 0 and w.u.mps == 0 for w in col)
  assert any(w.u.mps > 0 and w.v.mps == 0 for w in col) and any(w.u.mps < 0 and w.v.mps == 0 for w in col)
  pressures = [6000.0, 9000.0, 11000.0, 13000.0]
  assert wf.get_forecast_column(zero, zero, pressures, t0) == [wf.get_forecast(zero, zero, p, t0) for p in pressures]
  assert wf.get_ground_truth(zero, zero, 9000.0, t0) == wf.get_forecast(zero, zero, 9000.0, t0)


def test_perciatelli_features_long_horizon():
  """F12: 136 observations -- the WindGP must drop those older than 6 h (wind_gp.py:179-185)."""
  g = helpers.golden('f12_features_long')
  got = run_constructor(g, OracleForecast(field_of(g)), 0, n_steps=g['x'].shape[1])
  np.testing.assert_allclose(got[-16:], g['features'][0], rtol=0, atol=ATOL)
