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
from balloon_learning_environment_amd.env.balloon import balloon, power_table, pressure_range_builder
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
  fc = features.PerciatelliFeatureConstructor(forecast, simulator_data.Atmosphere(float(g['alpha'][j])))
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
  fc = features.PerciatelliFeatureConstructor(OracleForecast(field_of(g)), simulator_data.Atmosphere(0.5))
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
  from balloon_learning_environment_amd.env import wind_gp
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
    assert power_table.lookup(pr, soc) == want
  with pytest.raises(AssertionError):
    power_table.lookup(ka['raises'][0], 0.5)
  g = helpers.golden('f5_acs_power_table')
  for pr, soc, w in zip(g['pt_pr'], g['pt_soc'], g['pt_watts']):
    assert power_table.lookup(float(pr), float(soc)) == w


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
