"""env/balloon/standard_atmosphere.py:52-202 of the reference: `Atmosphere(key)` with `at_pressure` / `at_height`.

`at_pressure` -- the lookup the transition makes every stride -- runs the kernel's own device function
(`ble_probe_atmosphere_f32`); `at_height` (needed once per episode: the 50 000 ft pressure bound of the initial-condition
sampler, utils/sampling.py:86-117, which the reset kernel has inline) INVERTS that same device function by bracketing --
four launches of 4 096 pressures -- so that the package holds one implementation of the atmosphere, the device's."""
import dataclasses

import numpy as np

from balloon_learning_environment_amd.utils import units

DRY_AIR_SPECIFIC_GAS_CONSTANT = 8.3144621 / 0.028964922481160      # utils/constants.py: R / M_dry_air


@dataclasses.dataclass
class AtmosphericValues:
  height: units.Distance
  temperature: float  # K
  pressure: float     # Pa
  density: float      # kg/m^3


class AtmosphereOps:
  """at_pressure / at_height for anything that carries the lapse-rate mix `alpha` (standard_atmosphere.py:76-87)."""
  alpha: float

  def at_pressure(self, pressure: float) -> AtmosphericValues:
    from balloon_learning_environment_amd.env.balloon import _probes
    height, temperature = _probes.atmosphere(self.alpha, pressure)       # raises AssertionError out of range, like :126-127
    return AtmosphericValues(units.Distance(meters=height), temperature, float(pressure),
                             float(pressure) / (DRY_AIR_SPECIFIC_GAS_CONSTANT * temperature))

  def at_height(self, height: units.Distance) -> AtmosphericValues:
    from balloon_learning_environment_amd.env.balloon import _probes
    h = float(height.meters)
    assert -610.0 <= h < 85000.0, 'Atmosphere.at_height: height out of range (standard_atmosphere.py:94-95)'
    # height falls monotonically with pressure: bracket h on a geometric pressure grid and refine; the device evaluates in
    # fp64 and returns float32 (resolution 1 mm at 15 km), a float32 pressure resolves 6e-8 relative
    lo, hi = 0.3, 120000.0                                   # Pa: 85 km .. -610 m
    for _ in range(4):
      ps = np.geomspace(lo, hi, 4096)
      hs, _ = _probes.atmosphere_column(self.alpha, ps)
      ok = np.isfinite(hs)
      k = int(np.searchsorted(-hs[ok], -h))                   # first grid pressure whose height is <= h
      ps_ok = ps[ok]
      k = min(max(k, 1), ps_ok.size - 1)
      lo, hi = float(ps_ok[k - 1]), float(ps_ok[k])
    (h0, h1), (t0, t1) = _probes.atmosphere_column(self.alpha, [lo, hi])
    w = 0.0 if h0 == h1 else (h0 - h) / (h0 - h1)
    pressure = lo * (hi / lo) ** w
    temperature = t0 + w * (t1 - t0)
    return AtmosphericValues(units.Distance(meters=h), float(temperature), float(pressure),
                             float(pressure) / (DRY_AIR_SPECIFIC_GAS_CONSTANT * float(temperature)))


class Atmosphere(AtmosphereOps):
  """`Atmosphere(key)`: a new lapse-rate mix per reset(key) (the reference draws alpha ~ U[0, 1) from a jax key; here from a
  NumPy Philox stream keyed the same way -- the JAX streams themselves are not reproduced)."""

  def __init__(self, key):
    self.reset(key)

  def reset(self, key) -> None:
    seed = int(np.asarray(key).ravel()[-1]) if key is not None else 0
    self.alpha = float(np.random.Generator(np.random.Philox(seed)).uniform())
