"""FeatureConstructor interface (env/features.py:106-144).

The 1099-float PerciatelliFeatureConstructor (wind-column + GP) is the next scope row after
the transition and is not built yet.  `StateFeatureConstructor` is a small
placeholder observation so that the gym surface is complete; agents that need Perciatelli
features cannot drop in until that row lands.
"""
import abc

import numpy as np

from balloon_learning_environment_amd.env import simulator_data


class Box:
  """Stand-in for gym.spaces.Box (gym is not installed here)."""

  def __init__(self, low, high, dtype=np.float32):
    self.low = np.asarray(low, dtype); self.high = np.asarray(high, dtype)
    self.shape = self.low.shape; self.dtype = dtype

  def contains(self, x):
    x = np.asarray(x)
    return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


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
