"""power_table.lookup (env/balloon/power_table.py:21-38), host scalar: same 8x4 table the
device kernel (csrc/ble_physics.h::power_table_lookup) evaluates, as NumPy arrays."""
import numpy as np

_RATIO_EDGES = np.array([1.08, 1.11, 1.14, 1.17, 1.2, 1.23, 1.26])
_SOC_EDGES = np.array([[0.3, 0.4, 0.5], [0.3, 0.4, 0.7], [0.3, 0.4, 0.6], [0.3, 0.4, 0.5], [0.3, 0.4, 0.5],
                       [0.4, 0.5, np.inf], [0.5, 0.6, np.inf], [0.5, 0.6, np.inf]])
_WATTS = np.array([[0, 150, 175, 200], [0, 200, 200, 225], [0, 225, 225, 250], [0, 200, 225, 250],
                   [0, 225, 250, 275], [0, 275, 300, 300], [0, 300, 325, 325], [0, 325, 350, 350]])


def lookup(pressure_ratio: float, state_of_charge: float) -> float:
  assert pressure_ratio >= 0.99 and pressure_ratio <= 5
  row = int(np.searchsorted(_RATIO_EDGES, pressure_ratio, side='right'))
  col = int(np.searchsorted(_SOC_EDGES[row], state_of_charge, side='right'))
  return int(_WATTS[row, col])
