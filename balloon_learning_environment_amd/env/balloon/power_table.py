"""power_table.lookup (env/balloon/power_table.py:21-38) by the device function the transition and the observation use
(`ble_power_table_f32`, csrc/ble_physics.h::power_table_lookup), on one element."""
from balloon_learning_environment_amd.env.balloon import _probes


def lookup(pressure_ratio: float, state_of_charge: float) -> float:
  """Watts; AssertionError outside pressure_ratio in [0.99, 5] like the reference (:27)."""
  return int(_probes.power_table(pressure_ratio, state_of_charge))
