"""get_pressure_range (env/balloon/pressure_range_builder.py:31-275): the pressures a balloon
can float at with a safe superpressure.  TEST TOOLING (NumPy twin; the product's search runs inside ble_observe_f32); the cold-start solver is
reset_host.stable_params (stable_init.py:40-129), evaluated for all candidate levels at once."""
import dataclasses
import math

import numpy as np

import reset_host

_BUFFER = 250.0            # envelope_safety.BUFFER


@dataclasses.dataclass
class AccessiblePressureRange:
  min_pressure: float
  max_pressure: float


def _x_crossing(x1, y1, x2, y2, y_star):     # _compute_x_crossing :43-70
  if y_star < min(y1, y2) or y_star > max(y1, y2):
    raise ValueError('y_star must be in [y1, y2].')
  if x1 >= x2:
    raise ValueError('x2 must be greater than x1.')
  if y1 == y2:
    raise ValueError('y1 may not be equal to y2.')
  return abs((y_star - y1) / (y2 - y1)) * (x2 - x1) + x1


def _safe_pressure(p1, sp1, p2, sp2, min_sp, max_sp):   # _compute_safe_pressure :73-108
  if p1 >= p2:
    raise ValueError('pressure2 must be greater than pressure1.')
  if sp1 == sp2:
    raise ValueError('sp1 and sp2 may not be equal.')
  if (sp1 < min_sp and sp2 >= min_sp) or (sp1 >= min_sp and sp2 < min_sp):
    return _x_crossing(p1, sp1, p2, sp2, min_sp)
  if (sp1 > max_sp and sp2 <= max_sp) or (sp1 <= max_sp and sp2 > max_sp):
    return _x_crossing(p1, sp1, p2, sp2, max_sp)
  raise ValueError('Unable to find valid superpressure crossing for input params.')


def get_pressure_range(balloon_state, atmosphere) -> AccessiblePressureRange:
  """balloon_state: env.balloon.balloon.BalloonState; atmosphere: simulator_data.Atmosphere (alpha)."""
  b = balloon_state
  min_sp, max_sp = _BUFFER, b.envelope_max_superpressure - _BUFFER
  assert max_sp > 0.0
  atm = reset_host.AtmosphereTables(np.array([atmosphere.alpha]))
  search_max = float(atm.at_height(reset_host.MIN_ALTITUDE_M)[0][0])
  levels = np.linspace(1000.0, search_max, 20)
  atm20 = reset_host.AtmosphereTables(np.full(20, atmosphere.alpha))
  _, t_col = atm20.at_pressure(levels)
  total_empty_mass = b.payload_mass + b.envelope_mass + b.mols_lift_gas * reset_host.HE_MOLAR_MASS
  max_alt_p_over_t = total_empty_mass * reset_host.UNIVERSAL_GAS_CONSTANT / (reset_host.DRY_AIR_MOLAR_MASS * b.envelope_volume_base)
  p_over_t = levels / t_col
  assert np.all(np.diff(p_over_t) > 0)
  # scipy interp1d(kind='linear', fill_value='extrapolate')
  i = int(np.clip(np.searchsorted(p_over_t, max_alt_p_over_t), 1, 19))
  slope = (levels[i] - levels[i - 1]) / (p_over_t[i] - p_over_t[i - 1])
  min_pressure = float(slope * (max_alt_p_over_t - p_over_t[i - 1]) + levels[i - 1])
  max_pressure = search_max

  c = b.center_latlng          # (BalloonState.latlng on the host: the package's property is a device probe)
  lat_a, lng_a = reset_host.latlng_from_offset(np.array([math.radians(c.lat_deg)]), np.array([math.radians(c.lng_deg)]),
                                               np.array([b.x.m]), np.array([b.y.m]))
  lat, lng = float(lat_a[0]), float(lng_a[0])
  now = int(b.date_time.timestamp())

  def superpressures(ps):
    ps = np.asarray(ps, np.float64)
    n = ps.size
    out = reset_host.stable_params(ps, np.full(n, lat), np.full(n, lng), np.full(n, now, np.int64),
                                   np.full(n, b.upwelling_infrared), reset_host.AtmosphereTables(np.full(n, atmosphere.alpha)))
    return out['superpressure']

  sp_levels = superpressures(levels)

  def search(significant, direction):     # _search_for_safe_pressure :111-182
    sp = float(superpressures([significant])[0])
    if min_sp <= sp <= max_sp:
      return significant
    last = (significant, sp)
    order = range(19, -1, -1) if direction == 'min' else range(20)
    for k in order:
      pressure = float(levels[k])
      if (direction == 'min' and pressure > significant) or (direction == 'max' and pressure < significant):
        continue
      sp = float(sp_levels[k])
      if sp > max_sp or sp < min_sp:
        last = (pressure, sp)
        continue
      if direction == 'min':
        return _safe_pressure(pressure, sp, last[0], last[1], min_sp, max_sp)
      return _safe_pressure(last[0], last[1], pressure, sp, min_sp, max_sp)
    raise ValueError('Unable to find safe pressure for balloon.')

  return AccessiblePressureRange(min_pressure=float(search(min_pressure, 'max')),
                                 max_pressure=float(search(max_pressure, 'min')))
