"""TEST TOOLING, not product: batched episode reset on the host (NumPy, fp64, vectorised over environments) -- the state
sampler of the parity tests and profile scripts, and a second opinion on ble_reset_f32 (tests/test_reset_host.py).

Host-side logic of BalloonArena.reset (env/balloon_arena.py:161-182,228-268): sampling of
initial conditions (utils/sampling.py), the Newton cold start
(env/balloon/stable_init.py:40-157) and PowerSafetyLayer.__init__'s sunrise/sunset search
(env/balloon/power_safety.py:40-41 -> env/balloon/solar.py:239-483).  Reset is off the
per-step hot path (it runs once per episode); the transition itself runs only in
libble_hip.so.  Paths are relative to /root/reference/balloon_learning_environment/.

Random streams: the reference draws from JAX's threefry PRNG (absent here, parity
unpinned -- SURVEY.md 8c); this module uses numpy Philox streams, so seeds give different
(but identically distributed) initial conditions.
"""
from typing import Dict, Optional

import numpy as np

# utils/constants.py
GRAVITY = 9.80665
UNIVERSAL_GAS_CONSTANT = 8.3144621
DRY_AIR_MOLAR_MASS = 0.028964922481160
HE_MOLAR_MASS = 0.004002602
DRY_AIR_SPECIFIC_GAS_CONSTANT = UNIVERSAL_GAS_CONSTANT / DRY_AIR_MOLAR_MASS

_HEIGHTS = np.array([-610.0, 17000.0, 21000.0, 32000.0, 47000.0, 51000.0, 71000.0, 85000.0])
_LAPSE_LOW = np.array([-0.007, 0.006, 0.001, 0.0028, 0.0, -0.0028, -0.002])
_LAPSE_HIGH = np.array([-0.0058, 0.005, 0.001, 0.0028, 0.0, -0.0028, -0.002])

MIN_SOLAR_EL_DEG = -4.242
MIN_ALTITUDE_M = 50000.0 * 0.3048  # altitude_safety.MIN_ALTITUDE

UNIX_2011_01_01 = 1293840000  # units.datetime(2011, 1, 1)
UNIX_2014_12_31 = 1419984000  # units.datetime(2014, 12, 31)


# ----------------------------------------------------------------------- atmosphere
class AtmosphereTables:
  """standard_atmosphere.Atmosphere for a vector of alphas (:76-87,:156-183)."""

  def __init__(self, alpha: np.ndarray):
    alpha = np.asarray(alpha, np.float64).reshape(-1, 1)
    self.lapse = (1 - alpha) * _LAPSE_LOW + alpha * _LAPSE_HIGH           # (n, 7)
    n = alpha.shape[0]
    self.temp = np.empty((n, 8)); self.pres = np.empty((n, 8))
    self.temp[:, 0] = 300.0; self.pres[:, 0] = 108870.8213
    for i in range(7):
      dh = _HEIGHTS[i + 1] - _HEIGHTS[i]
      self.temp[:, i + 1] = self.temp[:, i] + self.lapse[:, i] * dh
      lap = self.lapse[:, i]
      iso = lap == 0.0
      safe = np.where(iso, 1.0, lap)
      lin = self.pres[:, i] * (self.temp[:, i + 1] / self.temp[:, i]) ** (-GRAVITY / (DRY_AIR_SPECIFIC_GAS_CONSTANT * safe))
      const = self.pres[:, i] * np.exp(-(GRAVITY * dh) / (DRY_AIR_SPECIFIC_GAS_CONSTANT * self.temp[:, i + 1]))
      self.pres[:, i + 1] = np.where(iso, const, lin)

  def at_pressure(self, pressure):
    """(:122-154) -> height [m], temperature [K]."""
    p = np.asarray(pressure, np.float64)
    assert np.all(p > self.pres[:, 7]) and np.all(p <= self.pres[:, 0])
    idx = np.clip((p[:, None] <= self.pres[:, 1:]).sum(1), 0, 6)          # first i with p > P[i+1]
    r = np.arange(p.size)
    lap, tb, pb, hb = self.lapse[r, idx], self.temp[r, idx], self.pres[r, idx], _HEIGHTS[idx]
    iso = lap == 0.0
    safe = np.where(iso, 1.0, lap)
    h_lin = ((p / pb) ** (-DRY_AIR_SPECIFIC_GAS_CONSTANT * safe / GRAVITY) - 1) * tb / safe + hb
    h_iso = (-DRY_AIR_SPECIFIC_GAS_CONSTANT * tb / GRAVITY) * np.log(p / pb) + hb
    h = np.where(iso, h_iso, h_lin)
    return h, tb + lap * (h - hb)

  def at_height(self, height):
    """(:89-120) -> pressure [Pa], temperature [K]."""
    h = np.broadcast_to(np.asarray(height, np.float64), (self.lapse.shape[0],)).copy()
    assert np.all(h >= _HEIGHTS[0]) and np.all(h < _HEIGHTS[7])
    idx = np.clip((h[:, None] >= _HEIGHTS[None, 1:]).sum(1), 0, 6)
    r = np.arange(h.size)
    lap, tb, pb, hb = self.lapse[r, idx], self.temp[r, idx], self.pres[r, idx], _HEIGHTS[idx]
    t = tb + lap * (h - hb)
    iso = lap == 0.0
    safe = np.where(iso, 1.0, lap)
    p_lin = pb * (t / tb) ** (-GRAVITY / (DRY_AIR_SPECIFIC_GAS_CONSTANT * safe))
    p_iso = pb * np.exp(-(GRAVITY * (h - hb)) / (DRY_AIR_SPECIFIC_GAS_CONSTANT * t))
    return np.where(iso, p_iso, p_lin), t


# ----------------------------------------------------------------------- solar
def latlng_from_offset(lat0_rad, lng0_rad, x_m, y_m):
  """utils/spherical_geometry.py:44-76 (+ s2 LatLng.normalized())."""
  heading = np.arctan2(x_m / 1000.0, y_m / 1000.0)
  angle = np.sqrt(x_m * x_m + y_m * y_m) / 6371000.0
  cos_a, sin_a = np.cos(angle), np.sin(angle)
  sin_from, cos_from = np.sin(lat0_rad), np.cos(lat0_rad)
  sin_lat = cos_a * sin_from + sin_a * cos_from * np.cos(heading)
  d_lng = np.arctan2(sin_a * cos_from * np.sin(heading), cos_a - sin_from * sin_lat)
  lat = np.clip(np.arcsin(np.clip(sin_lat, -1.0, 1.0)), -np.pi / 2, np.pi / 2)
  lng = np.remainder(lng0_rad + d_lng + np.pi, 2 * np.pi) - np.pi
  return lat, lng


def _civil(unix_s):
  days = np.floor_divide(unix_s, 86400)
  z = days + 719468
  era = np.floor_divide(z, 146097)
  doe = z - era * 146097
  yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
  y = yoe + era * 400
  doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
  mp = (5 * doy + 2) // 153
  d = doy - (153 * mp + 2) // 5 + 1
  m = np.where(mp < 10, mp + 3, mp - 9)
  return y + (m <= 2), m, d


def solar_calculator(lat_rad, lng_rad, unix_s):
  """solar.py:43-174 -> (elevation [deg], flux [W/m^2]); azimuth is not needed here."""
  unix_s = np.asarray(unix_s, np.int64)
  frac = np.remainder(unix_s, 86400) / 86400.0
  year, month, day = _civil(unix_s)
  year = year.astype(np.float64); month = month.astype(np.float64); day = day.astype(np.float64)
  jdn = (367.0 * year - np.floor(7.0 * (year + np.floor((month + 9.0) / 12.0)) / 4.0) -
         np.floor(3.0 * (np.floor((year + (month - 9.0) / 7.0) / 100.0) + 1.0) / 4.0) +
         np.floor(275.0 * month / 9.0) + day + 1721028.5)
  jc = ((jdn + frac) - 2451545.0) / 36525.0
  l0 = np.radians(280.46646 + jc * (36000.76983 + jc * 0.0003032))
  sin2l0, cos2l0, sin4l0 = np.sin(2.0 * l0), np.cos(2.0 * l0), np.sin(4.0 * l0)
  m0 = np.radians(357.52911 + jc * (35999.05029 - 0.0001537 * jc))
  sinm0, sin2m0, sin3m0 = np.sin(m0), np.sin(2.0 * m0), np.sin(3.0 * m0)
  mean_obl = np.radians(23.0 + (26.0 + ((21.448 - jc * (46.815 + jc * (0.00059 - jc * 0.001813)))) / 60.0) / 60.0)
  obl = mean_obl + np.radians(0.00256 * np.cos(np.radians(125.04 - 1934.136 * jc)))
  var_y = np.tan(obl / 2.0) ** 2
  ecc = 0.016708634 - jc * (0.000042037 + 0.0000001267 * jc)
  eot = 4.0 * (var_y * sin2l0 - 2.0 * ecc * sinm0 + 4.0 * ecc * var_y * sinm0 * cos2l0 -
               0.5 * var_y * var_y * sin4l0 - 1.25 * ecc * ecc * sin2m0)
  ha = np.radians(np.fmod(1440.0 * frac + np.degrees(eot) + 4.0 * np.degrees(lng_rad), 1440.0)) / 4.0
  ha = np.where(ha < 0, ha + np.pi, ha - np.pi)
  eoc = np.radians(sinm0 * (1.914602 - jc * (0.004817 + 0.000014 * jc)) + sin2m0 * (0.019993 - 0.000101 * jc) +
                   sin3m0 * 0.000289)
  app = l0 + eoc - np.radians(0.00569 - 0.00478 * np.sin(np.radians(125.04 - 1934.136 * jc)))
  decl = np.arcsin(np.sin(obl) * np.sin(app))
  zen = np.arccos(np.clip(np.sin(lat_rad) * np.sin(decl) + np.cos(lat_rad) * np.cos(decl) * np.cos(ha), -1.0, 1.0))
  el = 90.0 - np.degrees(zen)
  with np.errstate(divide='ignore', invalid='ignore'):
    tan_el = np.tan(np.radians(el))
    r_mid = 58.1 / tan_el - 0.07 / tan_el ** 3 + 0.000086 / tan_el ** 5
    r_low = 1735.0 + el * (-518.2 + el * (103.4 + el * (-12.79 + el * 0.711)))
    r_neg = -20.772 / tan_el
  refr = np.where(el > 85.0, 0.0, np.where(el > 5.0, r_mid, np.where(el > -0.575, r_low, r_neg)))
  flux = 1366.0 * (1 + 0.5 * (((1 + ecc) / (1 - ecc)) ** 2 - 1) * np.cos(m0))
  return el + refr / 3600.0, flux


def solar_atmospheric_attenuation(el_deg, pressure):
  """solar.py:177-209."""
  t = 614.0 * np.sin(np.radians(el_deg))
  airmass = 0.34764 * (pressure / 101325.0) * (np.sqrt(1229.0 + t * t) - t)
  att = 0.5 * (np.exp(-0.65 * airmass) + np.exp(-0.95 * airmass))
  return np.where(el_deg < MIN_SOLAR_EL_DEG, 0.0, att)


# ----------------------------------------------------------------------- sunrise / sunset
def _find_elevation(lat, lng, min_t, max_t, mode, target=0.0, dt=180):
  """solar.py:295-372, vectorised.  mode: 'min' | 'max' | 'abs' (|el - target|)."""
  def objective(idx):
    el, _ = solar_calculator(lat, lng, min_t + dt * idx)
    return el if mode == 'min' else (-el if mode == 'max' else np.abs(el - target))
  low = np.zeros(lat.shape, np.int64)
  high = (max_t - min_t) // dt
  while True:
    active = high > low + 1
    if not active.any():
      break
    mid = low + (high - low) / 2.0
    go_down = objective(low) < objective(high)
    new_high = np.where(go_down, np.ceil(mid).astype(np.int64), high)
    new_low = np.where(go_down, low, np.floor(mid).astype(np.int64))
    high = np.where(active, new_high, high)
    low = np.where(active, new_low, low)
  idx = np.where(objective(low) < objective(high), low, high)
  return min_t + dt * idx


def next_sunrise_sunset(lat_rad, lng_rad, unix_s):
  """solar.get_next_sunrise_sunset (solar.py:432-483) for vectors; integer unix seconds."""
  lat = np.asarray(lat_rad, np.float64); lng = np.asarray(lng_rad, np.float64)
  t = np.asarray(unix_s, np.int64)
  assert np.all(np.abs(np.degrees(lat)) < 60.0), 'High latitudes not supported.'
  h12, h24 = 12 * 3600, 24 * 3600
  afternoon = solar_calculator(lat, lng, t + 1)[0] < solar_calculator(lat, lng, t)[0]    # :239-256
  noon = _find_elevation(lat, lng, np.where(afternoon, t + h12, t), np.where(afternoon, t + h24, t + h12), 'max')
  midnight = _find_elevation(lat, lng, np.where(afternoon, t, t + h12), np.where(afternoon, t + h12, t + h24), 'min')
  sunrise = _find_elevation(lat, lng, np.where(afternoon, midnight, midnight - h24), noon, 'abs', MIN_SOLAR_EL_DEG)
  sunset = _find_elevation(lat, lng, np.where(afternoon, noon - h24, noon), midnight, 'abs', MIN_SOLAR_EL_DEG)
  sunrise = np.where(sunrise < t, sunrise + h24, sunrise)
  sunset = np.where(sunset < t, sunset + h24, sunset)
  return sunrise, sunset


# ----------------------------------------------------------------------- thermal / stable init
_SB = 0.000000056704


def _absorptivity_ir(t):
  return 0.04587 + 0.000232 * (t - 210)


def _total_absorptivity(a, r=0.0291):
  f = a * (1.0 + (1.0 - a - r) / (1.0 - r))
  if np.any(f < 0.0) or np.any(f > 1.0):
    raise ValueError('total_absorptivity: Computed total absorptivity factor out of expected range [0, 1].')
  return f


def d_balloon_temperature_dt(volume, mass, t_int, t_amb, pressure, el_deg, flux, earth_flux):
  """thermal.py:175-230."""
  radius = (3 * volume / (4 * np.pi)) ** (1 / 3)
  area = 4 * np.pi * radius * radius
  att = solar_atmospheric_attenuation(el_deg, pressure)
  q_solar = flux * att * 0.25 * area * _total_absorptivity(0.01435)
  q_earth = earth_flux * 0.4605 * area * _total_absorptivity(_absorptivity_ir((earth_flux / _SB) ** 0.25))
  q_emit = _SB * t_int ** 4 * area * _total_absorptivity(_absorptivity_ir(t_int))
  visc = 1.458e-6 * (t_amb ** 1.5) / (t_amb + 110.4)
  cond = 0.0241 * ((t_amb / 273.15) ** 0.9)
  prandtl = 0.804 - 3.25e-4 * t_amb
  rho = pressure * DRY_AIR_MOLAR_MASS / (UNIVERSAL_GAS_CONSTANT * t_amb)
  grashof = (9.80665 * rho ** 2 * (2 * radius) ** 3 / (t_amb * visc ** 2)) * np.abs(t_amb - t_int)
  ra = prandtl * grashof
  nusselt = 2 + 0.457 * ra ** 0.25 + (1 + 2.69e-8 * ra) ** (1.0 / 12.0)
  q_conv = area * (nusselt * cond / (2 * radius)) * (t_amb - t_int)
  return (q_solar + q_earth + q_conv - q_emit) / (1500 * mass)


def superpressure_and_volume(mols_air, t_int, pressure, mols_lift_gas=6830.0, v0=1804.0, dv_dp=0.0199):
  """balloon.py:552-609."""
  vu = (mols_lift_gas + mols_air) * UNIVERSAL_GAS_CONSTANT * t_int / pressure
  b = -(v0 - dv_dp * pressure)
  c = -(dv_dp * vu * pressure)
  v_full = 0.5 * (-b + np.sqrt(b * b - 4 * c))
  sp_full = pressure * vu / v_full - pressure
  slack = vu <= v0
  return np.where(slack, vu, v_full), np.where(slack, 0.0, sp_full)


def stable_params(pressure, lat_rad, lng_rad, unix_s, upwelling_ir, atm: AtmosphereTables) -> Dict[str, np.ndarray]:
  """stable_init.calculate_stable_params_for_pressure (stable_init.py:40-129), vectorised."""
  _, t_amb = atm.at_pressure(pressure)
  mols_air = ((pressure * DRY_AIR_MOLAR_MASS * 1804.0 / (UNIVERSAL_GAS_CONSTANT * t_amb) - 68.5 - 92.5 -
               HE_MOLAR_MASS * 6830.0) / DRY_AIR_MOLAR_MASS)
  mols_air = np.clip(mols_air, 0.0, None)
  t_int = np.full(pressure.shape, 206.0)
  el, flux = solar_calculator(lat_rad, lng_rad, unix_s)
  done = np.zeros(pressure.shape, bool)
  delta = 0.01
  for _ in range(10):
    d1 = d_balloon_temperature_dt(1804.0, 68.5, t_int - delta / 2, t_amb, pressure, el, flux, upwelling_ir)
    d2 = d_balloon_temperature_dt(1804.0, 68.5, t_int + delta / 2, t_amb, pressure, el, flux, upwelling_ir)
    d2t = (d2 - d1) / delta
    mean = (d1 + d2) / 2.0
    upd = (~done) & (np.abs(d2t) > 0.0)
    t_int = np.where(upd, t_int - mean / np.where(d2t == 0.0, 1.0, d2t), t_int)
    done = done | (np.abs(mean) < 1e-5)
    if done.all():
      break
  vol, sp = superpressure_and_volume(mols_air, t_int, pressure)
  return dict(ambient_temperature=t_amb, internal_temperature=t_int, mols_air=mols_air, envelope_volume=vol,
              superpressure=sp)


# ----------------------------------------------------------------------- sampling + reset
def sample_upwelling_infrared(rng: np.random.Generator, n: int) -> np.ndarray:
  """utils/sampling.py:120-152 as written: 315 * sigmoid(N(2, 315)), rejected below 225."""
  out = np.empty(n)
  todo = np.arange(n)
  while todo.size:
    z = rng.standard_normal(todo.size)
    with np.errstate(over='ignore'):
      s = 315.0 / (1.0 + np.exp(-(2.0 + 315.0 * z)))
    ok = s >= 225.0
    out[todo[ok]] = s[ok]
    todo = todo[~ok]
  return out


def sample_initial_state(n: int, seed: int = 0, upwelling_ir: Optional[str] = 'reference') -> Dict[str, np.ndarray]:
  """BalloonArena.reset's draws (balloon_arena.py:161-182,228-268) for n environments.

  Returns host arrays for every field of ble_state_f32 (values already rounded to the
  dtypes of the ABI where that matters for consistency: the Newton start and the
  sunrise search use the float32-rounded inputs the kernel will see).
  """
  rng = np.random.Generator(np.random.Philox(seed))
  f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
  alpha = f32(rng.uniform(0.0, 1.0, n))                                             # standard_atmosphere.py:82
  start = rng.integers(UNIX_2011_01_01, UNIX_2014_12_31, n)                         # sampling.py:65-83
  radius = 200_000.0 * rng.beta(1.2, 2.0, n)                                        # balloon_arena.py:153-154,246-247
  theta = rng.uniform(0.0, 2.0 * np.pi, n)
  x, y = f32(np.cos(theta) * radius), f32(np.sin(theta) * radius)
  lat_deg = f32(rng.uniform(-10.0, 10.0, n)); lng_deg = f32(rng.uniform(-175.0, 175.0, n))   # sampling.py:37-62
  atm = AtmosphereTables(alpha)
  p_max, _ = atm.at_height(MIN_ALTITUDE_M)                                          # sampling.py:102-111
  pressure = f32(rng.uniform(6500.0, p_max))
  if upwelling_ir == 'reference':
    ir = f32(sample_upwelling_infrared(rng, n))                                     # sampling.py:120-152
  else:
    ir = f32(np.full(n, float(upwelling_ir)))
  lat, lng = latlng_from_offset(np.radians(lat_deg), np.radians(lng_deg), x, y)     # BalloonState.latlng
  st = stable_params(pressure, lat, lng, start, ir, atm)                            # stable_init.py:132-157
  sunrise, sunset = next_sunrise_sunset(lat, lng, start)                            # power_safety.py:40-41
  out = dict(x=x, y=y, pressure=pressure, center_lat_deg=lat_deg, center_lng_deg=lng_deg, upwelling_infrared=ir,
             alpha=alpha, start_unix=start.astype(np.int64), time_elapsed_s=np.zeros(n, np.int32),
             sunrise_h_rel=(sunrise + 1800 - start).astype(np.int32), sunset_rel=(sunset - start).astype(np.int32),
             battery_charge=np.full(n, 2905.6),                                    # balloon.py:195
             acs_power=np.zeros(n), acs_mass_flow=np.zeros(n), solar_charging=np.zeros(n), power_load=np.zeros(n),
             status=np.zeros(n, np.uint8), last_command=np.ones(n, np.uint8),       # STAY, balloon.py:197-198
             alt_fsm=np.zeros(n, np.uint8), env_fsm=np.zeros(n, np.uint8), power_paused=np.zeros(n, np.uint8))
  out.update(st)
  return out
