"""FeatureConstructor interface (env/features.py:106-144).

`PerciatelliFeatureConstructor` (env/features.py:269-581): 16 ambient features + a 361-level relative wind column from the
WindGP, computed by `ble_observe_f32` (csrc/ble_observe.h) for a one-environment batch -- the WindGP history and its factor
live on the device.  There is no host implementation in this package: a forecast that is not a device grid is asked for its column
above the balloon and the kernel takes that as an input (`ble_observe_forecast_f32`).  (The
NumPy restatement the parity tests check the kernel against is oracle/features_oracle.py; the former host constructor lives
next to the tests, tests/features_host.py.)  `StateFeatureConstructor` is a compact raw-state observation for vectorised
consumers.
"""
import abc
import dataclasses
import math

import numpy as np

from balloon_learning_environment_amd.env import simulator_data
from balloon_learning_environment_amd.env.balloon import control
from balloon_learning_environment_amd.utils import constants
from balloon_learning_environment_amd.utils import transforms
from balloon_learning_environment_amd.utils import units

TOLERANCE = units.Distance(meters=1e-5)


def compute_solar_angle(balloon_state) -> float:
  """Solar elevation [deg] at the balloon (env/features.py:56-70), by the transition's device function."""
  from balloon_learning_environment_amd.env.balloon import solar
  return solar.solar_calculator(balloon_state.latlng, balloon_state.date_time)[0]


def compute_sunrise_time(balloon_state) -> float:
  """Normalised solar-cycle time (env/features.py:73-104): [sunrise, sunset] -> [0, pi],
  [sunset, next sunrise] -> [pi, 2 pi]; the two events by the reset kernel's search."""
  from balloon_learning_environment_amd.env.balloon import solar
  now = int(balloon_state.date_time.timestamp())
  sunrise, sunset = solar.get_next_sunrise_sunset(balloon_state.latlng, balloon_state.date_time)
  sunrise, sunset = int(sunrise.timestamp()), int(sunset.timestamp())
  day = constants.NUM_SECONDS_PER_DAY
  assert sunrise - day <= now <= sunrise
  assert sunset - day <= now <= sunset
  if sunset < sunrise:       # day: sunset is next
    sunrise -= day
    return math.pi * (now - sunrise) / (sunset - sunrise)
  sunset -= day              # night: sunrise is next
  return math.pi + math.pi * (now - sunset) / (sunrise - sunset)


class Box:
  """Stand-in for gym.spaces.Box (gym is not installed here)."""

  def __init__(self, low, high, dtype=np.float32):
    self.low = np.asarray(low, dtype); self.high = np.asarray(high, dtype)
    self.shape = self.low.shape; self.dtype = dtype

  def contains(self, x):
    x = np.asarray(x)
    return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

  def sample(self, rng=np.random):
    lo = np.where(np.isfinite(self.low), self.low, -1e6); hi = np.where(np.isfinite(self.high), self.high, 1e6)
    return (lo + (hi - lo) * rng.random_sample(self.shape)).astype(self.dtype)


class Discrete:
  def __init__(self, n): self.n = n
  def contains(self, x): return 0 <= int(x) < self.n

  def sample(self, rng=np.random): return int(rng.randint(self.n))


class FeatureConstructor(abc.ABC):
  @abc.abstractmethod
  def __init__(self, forecast, atmosphere) -> None: ...

  @abc.abstractmethod
  def observe(self, observation: simulator_data.SimulatorObservation) -> None: ...

  @abc.abstractmethod
  def get_features(self) -> np.ndarray: ...

  @property
  @abc.abstractmethod
  def observation_space(self): ...


class StateFeatureConstructor(FeatureConstructor):
  """[x km, y km, pressure kPa, T_amb, T_int, superpressure kPa, mols_air/1000, soc, u, v, paused]."""

  def __init__(self, forecast, atmosphere):
    self._last = None

  def observe(self, observation):
    self._last = observation

  def get_features(self):
    b = self._last.balloon_observation; w = self._last.wind_at_balloon
    return np.array([b.x.km, b.y.km, b.pressure / 1000.0, b.ambient_temperature, b.internal_temperature,
                     b.superpressure / 1000.0, b.mols_air / 1000.0, b.battery_soc, w.u.mps, w.v.mps,
                     float(b.navigation_is_paused)], dtype=np.float32)

  @property
  def observation_space(self):
    return Box(np.full(11, -np.inf), np.full(11, np.inf))


@dataclasses.dataclass
class PerciatelliWindFeature:
  uncertainty: float
  bearing: float
  magnitude: float

  def is_valid_wind(self):
    return self.magnitude != 1.0 or self.bearing != 1.0 or self.uncertainty != 0.0


class NamedPerciatelliFeatures:
  """Names for the 1099-vector (env/features.py:167-257)."""

  def __init__(self, features: np.ndarray):
    assert len(features) == 1099
    self._winds = features[16:]
    self.num_pressure_levels = len(self._winds) // 3
    a = features[:16]
    self.balloon_pressure = transforms.undo_linear_rescale_with_extrapolation(
        a[0], constants.PERCIATELLI_PRESSURE_RANGE_MIN, constants.PERCIATELLI_PRESSURE_RANGE_MAX)
    self.battery_charge = a[1]
    self.solar_elevation = a[2]
    self.sin_normalized_solar_cycle, self.cos_normalized_solar_cycle = a[3], a[4]
    self.sin_heading_to_station, self.cos_heading_to_station = a[5], a[6]
    self.distance_to_station = a[7]
    # feature order is up, stay, down (the reverse of the command enum)
    self.last_command = (control.AltitudeControlCommand.UP, control.AltitudeControlCommand.STAY,
                         control.AltitudeControlCommand.DOWN)[int(np.argmax(a[8:11]))]
    self.navigation_enabled, self.navigation_paused = a[11], a[12]
    self.has_excess_energy, self.descent_cost, self.internal_pressure_ratio = a[13], a[14], a[15]

  def wind_at(self, level: int) -> PerciatelliWindFeature:
    if 0 > level >= self.num_pressure_levels:
      raise ValueError(f'Invalid wind level: {level}')
    return PerciatelliWindFeature(*self._winds[level * 3:level * 3 + 3])

  def level_is_valid(self, level: int) -> bool: return self.wind_at(level).is_valid_wind()
  def magnitude(self, level: int) -> float: return self.wind_at(level).magnitude
  def bearing(self, level: int) -> float: return self.wind_at(level).bearing
  def uncertainty(self, level: int) -> float: return self.wind_at(level).uncertainty

  def wind_column_center(self) -> int:
    assert self.num_pressure_levels % 2 == 1
    return self.num_pressure_levels // 2


def convert_wind_feature_to_real_wind(wind: PerciatelliWindFeature) -> PerciatelliWindFeature:
  return PerciatelliWindFeature(wind.uncertainty,
                                transforms.undo_linear_rescale_with_extrapolation(wind.bearing, 0.0, math.pi),
                                transforms.undo_squash_to_unit_interval(wind.magnitude, 30.0))


_UNREACHABLE = (0.0, 1.0, 1.0)   # certain, wrong way, infinitely fast


class PerciatelliFeatureConstructor(FeatureConstructor):
  """env/features.py:269-581, computed by `ble_observe_f32` for a one-environment batch (the WindGP
  history and its factor live on the device).  Like the reference's it takes ANY wind_field.WindField as its forecast
  (features.py:290-299): a grid-based one is interpolated inside the kernel; any other (the reference's unit-test field,
  SimpleStaticWindField) is asked for its column above the balloon, `get_forecast_column(x, y, 181 levels, elapsed)`
  (features.py:499-503), which the kernel takes as an input (`ble_observe_forecast_f32`).  For N environments use
  `VecBalloonArena.observe` directly."""

  LEVELS = tuple(5000.0 + 50.0 * k for k in range(181))       # features.py:288-289: np.linspace(5000, 14000, 181)

  def __init__(self, forecast, atmosphere, device=None) -> None:
    from balloon_learning_environment_amd import vec_state       # (device module: imported on use)
    self._forecast, self._alpha = forecast, float(atmosphere.alpha)
    self._grid_based = getattr(forecast, 'grid', None) is not None
    self._sim = vec_state.VecSimulator(1, getattr(forecast, 'device', None) or device or 'cuda:0')
    self._sim.set_grid(forecast.grid if self._grid_based else np.zeros(vec_state.GRID_SHAPE, np.float32))
    self._features = None
    self.num_features = 1099

  def bind_state(self, sim) -> None:
    """The arena that owns this constructor flies its balloon in `sim` (a one-environment VecSimulator on the same
    device).  The private simulator then ALIASES that state: `observe_bound` reads it where it already is instead of
    copying 26 fields host -> device per step (0.36 ms of the single-env facade's 1.03 ms)."""
    from balloon_learning_environment_amd import device as dev
    assert sim.n == 1 and sim.device == self._sim.device
    self._sim.state = sim.state
    self._sim._struct = sim._struct          # (the arena's own struct: its device pointers AND its vehicle)
    self._bound = True

  def observe_bound(self, observation: simulator_data.SimulatorObservation) -> None:
    """observe() for the owning arena: `observation` describes the bound simulator's current state."""
    assert getattr(self, '_bound', False)
    self._observe(observation, copy_state=False)

  def _unbind(self) -> None:
    """Back to a private copy of the state (the WindGP history stays): from here on observe() honours the
    observation it is handed, like the reference's FeatureConstructor.observe(observation) contract."""
    import torch
    from balloon_learning_environment_amd import device as dev
    self._sim.state = {name: t.clone() for name, t in self._sim.state.items()}
    self._sim._struct = dev.state_struct(self._sim.state, self._sim.episode_cache)
    self._sim.set_vehicle(**self._sim.vehicle)
    self._bound = False

  def observe(self, observation: simulator_data.SimulatorObservation) -> None:
    """The reference's public contract (features.py:301-330): the observation handed in is what is observed.
    A constructor bound to its arena's state (bind_state) that is called from outside -- e.g. with a noisy or
    edited observation -- gives the alias up first and copies the observation's state like an unbound one."""
    if getattr(self, '_bound', False):
      self._unbind()
    self._observe(observation, copy_state=True)

  def _observe(self, observation: simulator_data.SimulatorObservation, copy_state: bool) -> None:
    import torch
    from balloon_learning_environment_amd.env.balloon import balloon as balloon_lib
    b = observation.balloon_observation
    if copy_state:
      row = balloon_lib.row_from_state(b, self._alpha)
      self._sim.set_state({k: np.array([v]) for k, v in row.items()})
      self._sim.set_vehicle(**balloon_lib.vehicle_of(b))        # battery_soc, excess_energy and get_pressure_range read the vehicle
    fc = self._forecast.get_forecast(b.x, b.y, b.pressure, b.time_elapsed)
    w = observation.wind_at_balloon
    noise = torch.tensor([[w.u.mps - fc.u.mps, w.v.mps - fc.v.mps]], dtype=torch.float32, device=self._sim.device)
    column = None
    if not self._grid_based:      # a forecast that is not a grid: its own column above the balloon (features.py:499-503)
      col = self._forecast.get_forecast_column(b.x, b.y, list(self.LEVELS), b.time_elapsed)
      column = torch.tensor([[[c.u.mps, c.v.mps] for c in col]], dtype=torch.float32, device=self._sim.device)
    self._features = self._sim.observe(noise, forecast_levels=column)[0].cpu().numpy()
    self._sim.check_errors()

  def get_features(self) -> np.ndarray:
    return self._features.copy()

  @property
  def observation_space(self) -> Box:
    low = np.zeros(1099, np.float32); high = np.ones(1099, np.float32)
    low[[3, 4, 5, 6]] = -1.0
    low[15], high[15] = 1.0, np.inf
    return Box(low, high)


DevicePerciatelliFeatureConstructor = PerciatelliFeatureConstructor      # (the name rounds 2-4 used)


def perciatelli_feature_constructor(forecast, atmosphere) -> FeatureConstructor:
  """Default factory of BalloonEnv / BalloonArena: the device observation (`ble_observe_f32`; `ble_observe_forecast_f32` for a
  forecast object that is not a device grid)."""
  return PerciatelliFeatureConstructor(forecast, atmosphere)
