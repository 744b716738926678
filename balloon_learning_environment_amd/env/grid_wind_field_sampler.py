"""GridWindFieldSampler (env/grid_wind_field_sampler.py:26-42) + FieldShape (generative/vae.py:26-93)."""
import abc
import dataclasses
import datetime as dt

import numpy as np


@dataclasses.dataclass
class FieldShape:
  latlng_slices: int = 21
  flow_field_width: int = 7
  pressure_slices: int = 10
  time_slices: int = 9
  latlng_displacement_km: float = 500.
  max_pressure_pa: float = 14000.
  min_pressure_pa: float = 5000.
  time_horizon_hours: int = 48

  def grid_shape(self): return (self.latlng_slices, self.latlng_slices, self.pressure_slices, self.time_slices, 2)
  def num_grid_points(self): return self.latlng_slices ** 2 * self.pressure_slices * self.time_slices
  def latlng_grid_points(self): return np.linspace(-self.latlng_displacement_km, self.latlng_displacement_km, self.latlng_slices, dtype=np.float32)
  def pressure_grid_points(self): return np.linspace(self.min_pressure_pa, self.max_pressure_pa, self.pressure_slices, dtype=np.float32)
  def time_grid_points(self): return np.linspace(0, self.time_horizon_hours, self.time_slices).astype(np.int32)


class GridWindFieldSampler(abc.ABC):
  @property
  @abc.abstractmethod
  def field_shape(self) -> FieldShape: ...

  @abc.abstractmethod
  def sample_field(self, key, date_time: dt.datetime) -> np.ndarray: ...


class GaussianFieldSampler(GridWindFieldSampler):
  """Synthetic stand-in for GenerativeWindFieldSampler (the VAE decoder weights are absent,
  /root/reference/.MISSING_LARGE_BLOBS): N(0, scale^2) m/s float32 field seeded by `key`."""

  def __init__(self, scale_mps: float = 5.0):
    self._shape = FieldShape()
    self._scale = scale_mps

  @property
  def field_shape(self): return self._shape

  def sample_field(self, key, date_time=None) -> np.ndarray:
    seed = int(np.asarray(key).ravel()[-1]) if key is not None else 0
    return (np.random.default_rng(seed).standard_normal(self._shape.grid_shape()) * self._scale).astype(np.float32)
