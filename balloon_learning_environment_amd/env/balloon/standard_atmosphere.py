"""env/balloon/standard_atmosphere.py:52-202 of the reference: `Atmosphere(key)` with `at_pressure` / `at_height`.

`at_pressure` -- the lookup the transition makes every stride -- runs the kernel's own device function
(`ble_probe_atmosphere_f32`); `at_height` (needed once per episode: the 50 000 ft pressure bound of the initial-condition
sampler, utils/sampling.py:86-117, which the reset kernel has inline) is the float64 device function
`ble_probe_atmosphere_at_height_f64` (ABI 5: the layer walk of the reference's transition tables; rounds 4-5 inverted the
float32 lookup by bracketing, 1e-7)."""
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
    pressure, temperature = _probes.atmosphere_at_height(self.alpha, h)      # raises AssertionError out of range, like :94-95
    return AtmosphericValues(units.Distance(meters=h), temperature, pressure, pressure / (DRY_AIR_SPECIFIC_GAS_CONSTANT * temperature))


class Atmosphere(AtmosphereOps):
  """`Atmosphere(key)`: a new lapse-rate mix per reset(key) (the reference draws alpha ~ U[0, 1) from a jax key; here from a
  NumPy Philox stream keyed the same way -- the JAX streams themselves are not reproduced)."""

  def __init__(self, key):
    self.reset(key)

  def reset(self, key) -> None:
    seed = int(np.asarray(key).ravel()[-1]) if key is not None else 0
    self.alpha = float(np.random.Generator(np.random.Philox(seed)).uniform())
