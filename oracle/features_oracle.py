"""CPU oracle for the observation path: the reference's
PerciatelliFeatureConstructor (env/features.py:269-581), WindGP (env/wind_gp.py:33-241) and
get_pressure_range (env/balloon/pressure_range_builder.py:31-275) restated in NumPy float64 on
top of the C oracle's primitives (solar calculator, sunrise search, cold-start solver, wind
interpolation, power table).

TEST INFRASTRUCTURE ONLY (see oracle.py).  PINNED: tests/test_oracle_golden.py checks it against
the reference's own 1099-vectors (tests/golden/f11_features.npz, f12_features_long.npz).
"""
import math

import numpy as np
import scipy.linalg

import oracle

N_LEVELS = 181
P_MIN, P_MAX = 5000.0, 14000.0                     # utils/constants.py:37-38
LEVELS = np.linspace(P_MIN, P_MAX, N_LEVELS)       # features.py:288-289
LENGTH_SCALE = np.array([357000.0, 357000.0, 326.0, 34560.0])      # wind_gp.py:33-35
SIGMA2, NOISE2 = 3.6 ** 2, 0.05                    # wind_gp.py:36-37
HORIZON_S = 6 * 3600                               # wind_gp.py:63
MIN_ALTITUDE_M = 15240.0                           # altitude_safety.py:33 (50 000 ft)
BUFFER_PA = 250.0                                  # envelope_safety.py BUFFER
MAX_SUPERPRESSURE = 2380.0
HE_MOLAR_MASS, AIR_MOLAR_MASS, R_GAS = 0.004002602, 0.028964922481160, 8.3144621
TOL_M = 1e-5                                       # features.py:52


def simple_static_wind_column(x, y, pressures, elapsed_s):
  """SimpleStaticWindField.get_forecast (reference env/wind_field.py:149-184) at an array of pressures: four sheets blowing E / N / W / S
  by pressure band."""
  p = np.asarray(pressures, np.float64)
  band = (p >= 8000.0).astype(int) + (p >= 10000.0) + (p >= 12000.0)
  return np.array([10.0, 0.0, -10.0, 0.0])[band], np.array([0.0, 10.0, 0.0, -10.0])[band]


def gp_kernel(a, b):                               # 3.6^2 * Matern(nu=0.5): s^2 exp(-|d / ls|)
  d = (a[:, None, :] - b[None, :, :]) / LENGTH_SCALE
  return SIGMA2 * np.exp(-np.sqrt((d * d).sum(-1)))


class FeatureOracle:
  """One environment.  `observe(row, err_uv)`: row is a dict of float64 state values in the
  units of ble_state_f32 (plus start_unix, time_elapsed_s), err_uv = measured - forecast."""

  def __init__(self, field, alpha, vehicle=None, forecast_column=None):
    """forecast_column: None (the forecast is the grid `field`) or a function (x m, y m, pressures Pa, elapsed s) -> (u, v) arrays: a
    forecast that is not a grid, asked for its column like the reference does (features.py:499-503 -> WindField.get_forecast_column).
    vehicle: None or a dict of the BalloonState vehicle fields that differ from the reference's defaults (oracle.VEHICLE_DEFAULTS):
    the features read battery_soc and excess_energy (balloon.py:223-238), get_pressure_range the envelope, masses and lift gas."""
    self.field, self.alpha = field, float(alpha)
    self.forecast_column = forecast_column
    self.vehicle = dict(vehicle or {})
    self.veh = dict(oracle.VEHICLE_DEFAULTS); self.veh.update(self.vehicle)
    self.locs, self.errs = [], []
    self.row = None

  def observe(self, row, err_uv):
    self.row = row
    self.locs.append([row['x'], row['y'], row['pressure'], float(row['time_elapsed_s'])])
    self.errs.append([float(err_uv[0]), float(err_uv[1])])

  # ---- wind_gp.py:136-241
  def query_column(self):
    r = self.row
    q = np.zeros((N_LEVELS, 4))
    q[:, 0], q[:, 1], q[:, 2], q[:, 3] = r['x'], r['y'], LEVELS, float(r['time_elapsed_s'])
    x, y = np.array(self.locs), np.array(self.errs)
    keep = np.abs(x[:, 3] - q[0, 3]) < HORIZON_S
    x, y = x[keep], y[keep]
    k = gp_kernel(x, x) + NOISE2 * np.eye(len(x))
    chol = scipy.linalg.cholesky(k, lower=True)
    k_star = gp_kernel(q, x)
    mean = k_star @ scipy.linalg.cho_solve((chol, True), y)
    v = scipy.linalg.solve_triangular(chol, k_star.T, lower=True)
    var = np.maximum(SIGMA2 - (v * v).sum(0), 0.0)
    if self.forecast_column is not None:
      fu, fv = self.forecast_column(r['x'], r['y'], LEVELS, int(r['time_elapsed_s']))
    else:
      fu, fv = oracle.wind_forecast(self.field, q[:, 0], q[:, 1], LEVELS, np.full(N_LEVELS, int(r['time_elapsed_s']), np.int64))
    mean[:, 0] += fu; mean[:, 1] += fv
    return mean, var / SIGMA2

  # ---- pressure_range_builder.py:203-275
  def pressure_range(self):
    r = self.row
    p_floor = float(oracle.at_height(self.alpha, MIN_ALTITUDE_M)[0][0])
    levels = np.linspace(1000.0, p_floor, 20)
    t_col = oracle.at_pressure(self.alpha, levels)[1]
    p_over_t = levels / t_col
    v = self.veh                                     # pressure_range_builder.py:236-245
    target = (v['payload_mass'] + v['envelope_mass'] + v['mols_lift_gas'] * HE_MOLAR_MASS) * R_GAS / (AIR_MOLAR_MASS * v['envelope_volume_base'])
    i = int(np.clip(np.searchsorted(p_over_t, target), 1, 19))           # interp1d linear, extrapolating
    ceiling = (levels[i] - levels[i - 1]) / (p_over_t[i] - p_over_t[i - 1]) * (target - p_over_t[i - 1]) + levels[i - 1]
    now = int(r['start_unix']) + int(r['time_elapsed_s'])

    def superpressure(ps):
      ps = np.atleast_1d(np.asarray(ps, np.float64)); n = ps.size
      out, _ = oracle.stable_init(ps, np.full(n, r['center_lat_deg']), np.full(n, r['center_lng_deg']), np.full(n, r['x']),
                                  np.full(n, r['y']), np.full(n, now, np.int64), np.full(n, r['upwelling_infrared']),
                                  np.full(n, self.alpha), vehicle=self.vehicle)
      return out['superpressure']

    sp_levels = superpressure(levels)
    lo, hi = BUFFER_PA, v['envelope_max_superpressure'] - BUFFER_PA          # :224-228

    def crossing(p1, s1, p2, s2):                                        # :73-108 (+ :43-70)
      if (s1 < lo) != (s2 < lo):
        target_sp = lo
      elif (s1 > hi) != (s2 > hi):
        target_sp = hi
      else:
        raise ValueError('no superpressure crossing')
      return abs((target_sp - s1) / (s2 - s1)) * (p2 - p1) + p1

    def search(significant, upward):                                     # :111-182
      sp = float(superpressure(significant)[0])
      if lo <= sp <= hi:
        return significant
      last = (significant, sp)
      for k in (range(20) if upward else range(19, -1, -1)):
        p = float(levels[k])
        if (upward and p < significant) or (not upward and p > significant):
          continue
        if not lo <= sp_levels[k] <= hi:
          last = (p, float(sp_levels[k]))
          continue
        return crossing(last[0], last[1], p, float(sp_levels[k])) if upward else crossing(p, float(sp_levels[k]), last[0], last[1])
      raise ValueError('no safe pressure')

    return search(float(ceiling), True), search(p_floor, False)

  # ---- features.py:301-581
  def features(self):
    r = self.row
    out = np.zeros(3 * (2 * N_LEVELS - 1) + 16, np.float32)
    now = int(r['start_unix']) + int(r['time_elapsed_s'])
    lat, lng = oracle.latlng_from_offset(math.radians(r['center_lat_deg']), math.radians(r['center_lng_deg']), r['x'], r['y'])
    el = float(oracle.solar_calculator(lat, lng, now)[0][0])
    soc = r['battery_charge'] / self.veh['battery_capacity_wh']
    out[0] = np.clip((r['pressure'] - P_MIN) / (P_MAX - P_MIN), 0.0, 1.0)
    out[1] = soc
    out[2] = np.clip((el + 90.0) / 180.0, 0.0, 1.0)
    sunrise, sunset = (int(v[0]) for v in oracle.next_sunrise_sunset(lat, lng, now))
    if sunset < sunrise:
      cycle = math.pi * (now - (sunrise - 86400)) / (sunset - (sunrise - 86400))
    else:
      cycle = math.pi + math.pi * (now - (sunset - 86400)) / (sunrise - (sunset - 86400))
    out[3], out[4] = math.sin(cycle), math.cos(cycle)
    heading = math.atan2(-r['x'] / 1000.0, -r['y'] / 1000.0)
    out[5], out[6] = math.sin(heading), math.cos(heading)
    dist_m = math.sqrt(r['x'] ** 2 + r['y'] ** 2)
    out[7] = (dist_m / 1000.0) / (dist_m / 1000.0 + 250.0)
    out[8], out[9], out[10] = (float(int(r['last_command']) == c) for c in (2, 1, 0))
    paused = bool(r['power_paused']) or int(r['env_fsm']) != 0 or int(r['alt_fsm']) != 0
    out[11], out[12] = float(paused), float(not paused)
    out[13] = float(float(oracle.solar_power(el, r['pressure'])[0][0]) > self.veh['daytime_power_load_w'] and soc > 0.99)
    ratio = (r['pressure'] + max(r['superpressure'], 0.0)) / r['pressure']
    out[14] = np.clip((float(oracle.power_table(ratio, soc)[0][0]) - 100.0) / 200.0, 0.0, 1.0)
    out[15] = ratio

    mean, dev = self.query_column()
    level = int(round((min(max(r['pressure'], P_MIN), P_MAX) - P_MIN) / (LEVELS[1] - LEVELS[0])))
    pad_above = N_LEVELS - level - 1
    to_station = -np.array([r['x'], r['y']]) / (dist_m + TOL_M)
    p_lo, p_hi = self.pressure_range()
    speed = np.linalg.norm(mean, axis=1)
    unit = mean / (speed + TOL_M).reshape(-1, 1)
    if dist_m < TOL_M:
      angle = np.zeros(N_LEVELS)
    else:
      angle = np.where(speed < TOL_M, np.pi, np.arccos(np.clip(unit @ to_station, -1.0, 1.0)))
    col = np.empty((2 * N_LEVELS - 1, 3), np.float32)
    col[:] = (0.0, 1.0, 1.0)
    ok = (LEVELS >= p_lo) & (LEVELS <= p_hi)
    body = col[pad_above:pad_above + N_LEVELS]
    body[ok, 0], body[ok, 1], body[ok, 2] = dev[ok], angle[ok] / math.pi, (speed / (speed + 30.0))[ok]
    out[16:] = col.reshape(-1)
    return out
