"""WindField interface of the reference (env/wind_field.py:39-145) over a device-resident grid."""
import abc
import datetime as dt
from typing import List, NamedTuple, Sequence

from balloon_learning_environment_amd.utils import units


class WindVector(NamedTuple):
  u: units.Velocity
  v: units.Velocity

  def add(self, other: 'WindVector') -> 'WindVector':
    if not isinstance(other, WindVector):
      raise NotImplementedError(f'Cannot add WindVector with {type(other)}')
    return WindVector(self.u + other.u, self.v + other.v)

  def __str__(self) -> str:
    return f'({self.u}, {self.v})'


class WindField(abc.ABC):
  """Point lookups in a wind field.  Like the reference (env/wind_field.py:58-59,125-145) every
  WindField carries a SimplexWindNoise and `get_ground_truth` = forecast + noise; `noise=False` is the
  explicit opt-out (parity tests with forecast == truth).  The noise model is created at the first
  reset() -- it lives on the GPU (csrc/ble_noise.h; its 4-D primitive is not opensimplex 0.3's, which is
  absent: parity unpinned), while constructing a field and reading its forecast need none."""

  def __init__(self, noise: bool = True, device='cuda:0'):
    self._noise_enabled, self._noise_device = bool(noise), device
    self.noise_model = None

  def _ensure_noise_model(self):
    if getattr(self, '_noise_enabled', True) and getattr(self, 'noise_model', None) is None:
      from balloon_learning_environment_amd.env import simplex_wind_noise
      self.noise_model = simplex_wind_noise.SimplexWindNoise(getattr(self, '_noise_device', 'cuda:0'))
    return getattr(self, 'noise_model', None)

  @abc.abstractmethod
  def reset_forecast(self, key, date_time: dt.datetime) -> None: ...

  @abc.abstractmethod
  def get_forecast(self, x: units.Distance, y: units.Distance, pressure: float,
                   elapsed_time: dt.timedelta) -> WindVector: ...

  def get_forecast_column(self, x, y, pressures: Sequence[float], elapsed_time) -> List[WindVector]:
    return [self.get_forecast(x, y, p, elapsed_time) for p in pressures]

  def reset(self, key, date_time: dt.datetime) -> None:
    model = self._ensure_noise_model()
    if model is not None:
      model.reset(key)
    self.reset_forecast(key, date_time)

  def get_wind_noise(self, x, y, pressure, elapsed_time) -> WindVector:
    model = getattr(self, 'noise_model', None)
    if model is None:       # noise=False, or no reset() yet
      return WindVector(units.Velocity(mps=0.0), units.Velocity(mps=0.0))
    return model.get_wind_noise(x, y, pressure, elapsed_time)

  def get_ground_truth(self, x, y, pressure, elapsed_time) -> WindVector:
    return self.get_forecast(x, y, pressure, elapsed_time).add(self.get_wind_noise(x, y, pressure, elapsed_time))


class SimpleStaticWindField(WindField):
  """Four horizontal sheets blowing E / N / W / S by pressure band (env/wind_field.py:149-184).
  Host-only forecast object (the reference's unit-test wind field); an arena that flies in it
  hands the kernel the looked-up vector through the additive wind input."""

  def reset_forecast(self, unused_key, unused_date_time) -> None:
    pass

  def get_forecast(self, unused_x, unused_y, pressure: float, unused_elapsed_time) -> WindVector:
    band = int(pressure >= 8000.0) + int(pressure >= 10000.0) + int(pressure >= 12000.0)
    u, v = ((10.0, 0.0), (0.0, 10.0), (-10.0, 0.0), (0.0, -10.0))[band]
    return WindVector(units.Velocity(mps=u), units.Velocity(mps=v))
