"""env/grid_based_wind_field_test.py:33-234 and env/wind_field_test.py:25-70 of the reference, test by test, on this
package's GridBasedWindField -- whose lookups are the device kernels (`ble_forecast_f32`, `ble_forecast_column_f32`) and
whose ground truth adds the device wind-noise kernel.  Same calls and literals; the reference's SimpleWindFieldSampler draws
its normal field with jax, this one with NumPy (any field does: the tests are properties of the interpolation)."""
import datetime as dt

import numpy as np
import pytest
import torch

from balloon_learning_environment_amd.utils import units

pytestmark = pytest.mark.gpu
km, hours = (lambda v: units.Distance(km=v)), (lambda v: dt.timedelta(hours=v))


@pytest.fixture(scope='module')
def wf():
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device; none visible')
  from balloon_learning_environment_amd.env import grid_based_wind_field, grid_wind_field_sampler
  from balloon_learning_environment_amd.utils import test_helpers

  class SimpleWindFieldSampler(grid_wind_field_sampler.GridWindFieldSampler):      # grid_based_wind_field_test.py:33-45
    @property
    def field_shape(self):
      return grid_wind_field_sampler.FieldShape()

    def sample_field(self, key, date_time):
      return np.random.default_rng(int(np.asarray(key).ravel()[-1])).standard_normal(self.field_shape.grid_shape()).astype(np.float32)
  field = grid_based_wind_field.GridBasedWindField(SimpleWindFieldSampler())
  field.reset(np.array([0, 0], np.uint32), test_helpers.START_DATE_TIME)
  return field


X, Y, PRESSURE, T0 = units.Distance(m=0.0), units.Distance(m=0.0), 9000.0, dt.timedelta(seconds=0)


def test_grid_based_wind_field_returns_consistent_forecast(wf):             # :60-65
  assert wf.get_forecast(X, Y, PRESSURE, T0) == wf.get_forecast(X, Y, PRESSURE, T0)


def test_grid_based_wind_field_returns_consistent_true_wind(wf):            # :67-74
  assert wf.get_ground_truth(X, Y, PRESSURE, T0) == wf.get_ground_truth(X, Y, PRESSURE, T0)


def test_grid_based_wind_field_returns_different_forecast_and_true_wind(wf):    # :76-84
  assert wf.get_forecast(X, Y, PRESSURE, T0) != wf.get_ground_truth(X, Y, PRESSURE, T0)


@pytest.mark.parametrize('x1,x2,y1,y2,p1,p2,t1,t2', [
    (km(-300.0), km(-250.0), km(0.0), km(0.0), 9000.0, 9000.0, hours(0), hours(0)),       # x_direction
    (km(0.0), km(0.0), km(0.0), km(50.0), 9000.0, 9000.0, hours(0), hours(0)),            # y_direction
    (km(0.0), km(0.0), km(0.0), km(0.0), 9000.0, 8000.0, hours(0), hours(0)),             # pressure_direction
    (km(0.0), km(0.0), km(0.0), km(0.0), 9000.0, 9000.0, hours(6), hours(12)),            # time_direction
])
def test_grid_based_wind_field_interpolates_correctly_between_grid_points(wf, x1, x2, y1, y2, p1, p2, t1, t2):    # :86-158
  fc1, fc2 = wf.get_forecast(x1, y1, p1, t1), wf.get_forecast(x2, y2, p2, t2)
  mid = wf.get_forecast((x1 + x2) / 2.0, (y1 + y2) / 2.0, (p1 + p2) / 2.0, (t1 + t2) / 2.0)
  assert mid.u.meters_per_second == pytest.approx(((fc1.u + fc2.u) / 2.0).meters_per_second, abs=5e-6)       # places=5
  assert mid.v.meters_per_second == pytest.approx(((fc1.v + fc2.v) / 2.0).meters_per_second, abs=5e-6)


def test_grid_based_wind_field_boomerangs_correctly(wf):                    # :160-182
  fc1, fc2 = wf.get_forecast(X, Y, PRESSURE, hours(46)), wf.get_forecast(X, Y, PRESSURE, hours(50))
  fc3, fc4 = wf.get_forecast(X, Y, PRESSURE, hours(46 + 48 * 2)), wf.get_forecast(X, Y, PRESSURE, hours(50 + 48 * 2))
  assert fc1 == fc2 and fc1 == fc3 and fc1 == fc4
  assert fc1 != wf.get_forecast(X, Y, PRESSURE, hours(49))


@pytest.mark.parametrize('x1,x2,y1,y2,p1,p2', [
    (km(-500.0), km(-550.0), km(0.0), km(0.0), 9000.0, 9000.0),
    (km(0.0), km(0.0), km(500.0), km(900.0), 9000.0, 9000.0),
    (km(0.0), km(0.0), km(0.0), km(0.0), 5000.0, 1000.0),
])
def test_grid_based_wind_field_extends_wind_field_beyond_grid(wf, x1, x2, y1, y2, p1, p2):     # :184-223
  assert wf.get_forecast(x1, y1, p1, T0) == wf.get_forecast(x2, y2, p2, T0)


def test_grid_based_wind_field_get_wind_column_matches_get_ground_truth(wf):        # :225-234
  pressures = tuple(range(5_000, 15_000, 1_000))
  assert [wf.get_forecast(X, Y, p, T0) for p in pressures] == wf.get_forecast_column(X, Y, pressures, T0)


def test_forecast_before_reset_raises_like_the_reference():                 # grid_based_wind_field.py:86-87
  from balloon_learning_environment_amd.env import grid_based_wind_field, grid_wind_field_sampler
  fresh = grid_based_wind_field.GridBasedWindField(grid_wind_field_sampler.GaussianFieldSampler())
  with pytest.raises(RuntimeError, match='reset'):
    fresh.get_forecast(X, Y, PRESSURE, T0)


def test_simple_static_wind_field_like_the_reference():                     # wind_field_test.py:33-70
  from balloon_learning_environment_amd.env import wind_field
  x, y, delta = units.Distance(km=2.1), units.Distance(km=2.2), dt.timedelta(minutes=3)
  field = wind_field.SimpleStaticWindField()
  vec = lambda u, v: wind_field.WindVector(units.Velocity(mps=u), units.Velocity(mps=v))
  assert field.get_forecast(x, y, 9323.0, delta) == vec(0.0, 10.0)         # north
  assert field.get_forecast(x, y, 13999.0, delta) == vec(0.0, -10.0)       # south
  assert field.get_forecast(x, y, 5523.0, delta) == vec(10.0, 0.0)         # east
  assert field.get_forecast(x, y, 11212.0, delta) == vec(-10.0, 0.0)       # west
  column = field.get_forecast_column(x, y, [10_000.0, 11_000.0], delta)
  assert column[0] == field.get_forecast(x, y, 10_000.0, delta) and column[1] == field.get_forecast(x, y, 11_000.0, delta)
