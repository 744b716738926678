"""env/balloon/acs.py:24-68 of the reference on the transition's device function (`ble_probe_acs_f32`)."""
from balloon_learning_environment_amd.env.balloon import _probes
from balloon_learning_environment_amd.utils import units


def get_most_efficient_power(pressure_ratio: float) -> units.Power:                  # :44-58
  return units.Power(watts=_probes.acs(pressure_ratio)[0])


def get_fan_efficiency(pressure_ratio: float, power: units.Power) -> float:          # :61-64
  """The fan-efficiency table along the operating curve the transition flies (power = get_most_efficient_power(ratio)) --
  the only place the simulator reads it (balloon.py:500-510)."""
  w, eff, _ = _probes.acs(pressure_ratio)
  if abs(power.watts - w) > 1e-3:
    raise NotImplementedError('the device function evaluates the efficiency at the most efficient power of the ratio')
  return eff


def get_mass_flow(power: units.Power, efficiency: float) -> float:                   # :67-68
  return efficiency * power.watts / 3600.0
