#!/usr/bin/env python3
"""Generates the golden vectors in tests/golden/*.npz from the REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference).  It installs the
container-type shims of oracle/ref_shims.py, imports the reference modules
unmodified, evaluates them in fp64 on seeded inputs and stores inputs + outputs.
The .npz files are data (inputs and expected outputs); no reference source is stored.

  python tests/golden/make_golden.py            # regenerate everything

Fixture ids follow SURVEY.md section 8(c): F1 atmosphere, F2 solar, F3 thermal,
F4 superpressure/volume, F5 ACS + power table, F6 safety-layer traces, F7 wind
interpolation, F8 simulate_step trajectories, F9 arena-style step with a grid
wind field, F10 reset path (stable_init, sunrise/sunset).
"""
import datetime as dt
import os
import sys

# One BLAS / OpenMP thread, whatever the caller's environment says: a threaded matrix product sums in an order that depends on the
# thread count, and the fixtures are held to their generator BYTE for byte (tests/test_golden_reproducible.py) -- F15's float64 MLP and
# the reference's own GP solves came out one ulp apart between 4 and 8 OpenBLAS threads.  Before NumPy is imported.
for _var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
  os.environ[_var] = '1'

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ref_shims  # noqa: E402

ref_shims.install()

from balloon_learning_environment.env import balloon_env  # noqa: E402
from balloon_learning_environment.env import simulator_data  # noqa: E402
from balloon_learning_environment.env import wind_field  # noqa: E402
from balloon_learning_environment.env.balloon import acs  # noqa: E402
from balloon_learning_environment.env.balloon import altitude_safety  # noqa: E402
from balloon_learning_environment.env.balloon import balloon  # noqa: E402
from balloon_learning_environment.env.balloon import control  # noqa: E402
from balloon_learning_environment.env.balloon import envelope_safety  # noqa: E402
from balloon_learning_environment.env.balloon import power_safety  # noqa: E402
from balloon_learning_environment.env.balloon import power_table  # noqa: E402
from balloon_learning_environment.env.balloon import solar  # noqa: E402
from balloon_learning_environment.env.balloon import stable_init  # noqa: E402
from balloon_learning_environment.env.balloon import thermal  # noqa: E402
from balloon_learning_environment.utils import spherical_geometry  # noqa: E402
from balloon_learning_environment.utils import units  # noqa: E402
import s2sphere as s2  # noqa: E402  (shim)

UNIX_2011 = int(units.datetime(2011, 1, 1).timestamp())
UNIX_2015 = int(units.datetime(2015, 1, 3).timestamp())


OUT_DIR = os.environ.get('BLE_GOLDEN_OUT') or HERE   # tests/test_golden_reproducible.py regenerates into a scratch directory


def save(name, **arrays):
  path = os.path.join(OUT_DIR, name + '.npz')
  np.savez_compressed(path, **arrays)
  print(f'{name}: {os.path.getsize(path)} bytes')


# ----------------------------------------------------------------------------- F1
def f1_atmosphere():
  alphas = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 0.3137])
  pressures = np.concatenate([np.linspace(3000.0, 20000.0, 64),
                              np.array([0.5, 3.0, 50.0, 70.0, 100.0, 800.0, 108000.0])])
  heights = np.concatenate([np.linspace(-600.0, 84000.0, 48), np.array([15240.0, 15392.4, 15544.8])])
  out = dict(alphas=alphas, pressures=pressures, heights=heights)
  h_of_p = np.empty((alphas.size, pressures.size)); t_of_p = np.empty_like(h_of_p); rho_of_p = np.empty_like(h_of_p)
  p_of_h = np.empty((alphas.size, heights.size)); t_of_h = np.empty_like(p_of_h)
  lapse = np.empty((alphas.size, 7)); ttr = np.empty((alphas.size, 8)); ptr = np.empty((alphas.size, 8))
  for i, a in enumerate(alphas):
    atm = ref_shims.make_atmosphere(float(a))
    lapse[i] = atm._lapse_rates; ttr[i] = atm._temperature_transitions; ptr[i] = atm._pressure_transitions
    for j, p in enumerate(pressures):
      v = atm.at_pressure(float(p))
      h_of_p[i, j], t_of_p[i, j], rho_of_p[i, j] = v.height.meters, v.temperature, v.density
    for j, h in enumerate(heights):
      v = atm.at_height(units.Distance(meters=float(h)))
      p_of_h[i, j], t_of_h[i, j] = v.pressure, v.temperature
  save('f1_atmosphere', h_of_p=h_of_p, t_of_p=t_of_p, rho_of_p=rho_of_p, p_of_h=p_of_h, t_of_h=t_of_h,
       lapse=lapse, temperature_transitions=ttr, pressure_transitions=ptr, **out)


# ----------------------------------------------------------------------------- F2
def f2_solar():
  rng = np.random.default_rng(2)
  n = 320
  lat = rng.uniform(-60, 60, n); lng = rng.uniform(-180, 180, n)
  lat[:64] = rng.uniform(-15, 15, 64)
  t = rng.integers(UNIX_2011, UNIX_2015, n)
  lat_rad = np.radians(lat); lng_rad = np.radians(lng)
  el = np.empty(n); az = np.empty(n); flux = np.empty(n)
  for i in range(n):
    ll = s2.LatLng.from_radians(float(lat_rad[i]), float(lng_rad[i]))
    el[i], az[i], flux[i] = solar.solar_calculator(ll, units.datetime_from_timestamp(int(t[i])))
  # attenuation / power / shadow on an (el, p) set, including thresholds
  els = np.concatenate([np.linspace(-90, 90, 61), np.array([-4.242, -4.2421, -4.2419, 34.3, 34.5, 37.6, 37.8])])
  ps = np.array([0.0, 3000.0, 5000.0, 8000.0, 11000.0, 14000.0, 20000.0, 101325.0])
  E, P = np.meshgrid(els, ps, indexing='ij')
  att = np.vectorize(lambda e, p: float(solar.solar_atmospheric_attenuation(float(e), float(p))))(E, P)
  pw = np.vectorize(lambda e, p: float(solar.solar_power(float(e), float(p)).watts))(E, P)
  sh33 = np.array([solar.balloon_shadow(float(e), 3.3) for e in els])
  sh27 = np.array([solar.balloon_shadow(float(e), 2.7) for e in els])
  # lat/lng from offsets (spherical_geometry.calculate_latlng_from_offset + normalized())
  m = 256
  lat0 = np.radians(rng.uniform(-12, 12, m)); lng0 = np.radians(rng.uniform(-179, 179, m))
  x = rng.uniform(-6e5, 6e5, m); y = rng.uniform(-6e5, 6e5, m)
  x[:4] = [0.0, 0.0, 1000.0, -250.0]; y[:4] = [0.0, 5000.0, 0.0, 0.0]
  olat = np.empty(m); olng = np.empty(m)
  for i in range(m):
    ll = spherical_geometry.calculate_latlng_from_offset(
        s2.LatLng.from_radians(float(lat0[i]), float(lng0[i])), units.Distance(m=float(x[i])),
        units.Distance(m=float(y[i])))
    olat[i], olng[i] = ll.lat().radians, ll.lng().radians
  save('f2_solar', lat_rad=lat_rad, lng_rad=lng_rad, unix_s=t, el=el, az=az, flux=flux,
       att_el=E, att_p=P, attenuation=att, power=pw, shadow_el=els, shadow33=sh33, shadow27=sh27,
       off_lat0=lat0, off_lng0=lng0, off_x=x, off_y=y, off_lat=olat, off_lng=olng)


# ----------------------------------------------------------------------------- F3
def f3_thermal():
  rng = np.random.default_rng(3)
  n = 256
  v = rng.uniform(900, 1900, n); ti = rng.uniform(180, 260, n); ta = rng.uniform(180, 230, n)
  p = rng.uniform(4000, 15000, n); el = rng.uniform(-90, 90, n); fl = rng.uniform(1310, 1420, n)
  ir = rng.uniform(100, 400, n)
  ti[:8] = ta[:8]  # zero convection
  out = np.array([thermal.d_balloon_temperature_dt(float(v[i]), 68.5, float(ti[i]), float(ta[i]), float(p[i]),
                                                   float(el[i]), float(fl[i]), float(ir[i])) for i in range(n)])
  save('f3_thermal', volume=v, t_int=ti, t_amb=ta, pressure=p, el=el, flux=fl, ir=ir, dtdt=out)


# ----------------------------------------------------------------------------- F4
def f4_sp_volume():
  rng = np.random.default_rng(4)
  n = 256
  na = rng.uniform(0, 3500, n); ti = rng.uniform(180, 260, n); p = rng.uniform(4000, 15000, n)
  na[:32] = rng.uniform(0, 300, 32); p[:32] = rng.uniform(9000, 15000, 32)  # not fully inflated
  vol = np.empty(n); sp = np.empty(n)
  for i in range(n):
    vol[i], sp[i] = balloon.calculate_superpressure_and_volume(6830.0, float(na[i]), float(ti[i]), float(p[i]),
                                                               1804, 0.0199)
  save('f4_sp_volume', mols_air=na, t_int=ti, pressure=p, volume=vol, superpressure=sp)


# ----------------------------------------------------------------------------- F5
def f5_acs_power_table():
  rng = np.random.default_rng(5)
  pr = np.concatenate([np.linspace(0.9, 1.6, 71), rng.uniform(1.0, 1.4, 57),
                       np.linspace(1.05, 1.35, 13)])
  power = np.array([float(acs.get_most_efficient_power(float(r)).watts) for r in pr])
  eff = np.array([acs.get_fan_efficiency(float(r), units.Power(watts=float(w))) for r, w in zip(pr, power)])
  mdot = np.array([acs.get_mass_flow(units.Power(watts=float(w)), float(e)) for w, e in zip(power, eff)])
  # free (pr, power) efficiency grid, including outside the table
  pr2 = rng.uniform(1.0, 1.45, 128); pw2 = rng.uniform(50, 450, 128)
  pr2[:13] = np.linspace(1.05, 1.35, 13); pw2[:13] = 200.0
  eff2 = np.array([acs.get_fan_efficiency(float(r), units.Power(watts=float(w))) for r, w in zip(pr2, pw2)])
  # power table
  prt = np.concatenate([rng.uniform(0.99, 1.4, 200), np.array([0.99, 1.08, 1.11, 1.14, 1.17, 1.2, 1.23, 1.26, 5.0])])
  soc = np.concatenate([rng.uniform(0, 1, 200), np.array([0.3, 0.4, 0.5, 0.6, 0.7, 0.0, 1.0, 0.45, 0.55])])
  watts = np.array([power_table.lookup(float(a), float(b)) for a, b in zip(prt, soc)], dtype=np.float64)
  save('f5_acs_power_table', pr=pr, power=power, eff=eff, mass_flow=mdot, pr2=pr2, power2=pw2, eff2=eff2,
       pt_pr=prt, pt_soc=soc, pt_watts=watts)


# ----------------------------------------------------------------------------- F6
_ALT = {altitude_safety._AltitudeState.NOMINAL: 0, altitude_safety._AltitudeState.LOW: 1,
        altitude_safety._AltitudeState.VERY_LOW: 2}
_ENV = {envelope_safety._SuperpressureState.NOMINAL: 0, envelope_safety._SuperpressureState.LOW_CRITICAL: 1,
        envelope_safety._SuperpressureState.LOW: 2, envelope_safety._SuperpressureState.HIGH: 3,
        envelope_safety._SuperpressureState.HIGH_CRITICAL: 4}


def f6_safety():
  rng = np.random.default_rng(6)
  # envelope: random walk of superpressure through every band, all three actions
  n = 600
  sp = np.clip(np.cumsum(rng.normal(0, 60, n)) + 200, -50, None)
  sp[200:400] = 2000 + np.cumsum(rng.normal(0, 40, 200))
  sp[400:] = rng.uniform(0, 2500, 200)
  act = rng.integers(0, 3, n).astype(np.uint8)
  layer = envelope_safety.EnvelopeSafetyLayer(2380)
  ea = np.empty(n, np.uint8); ef = np.empty(n, np.uint8)
  for i in range(n):
    ea[i] = int(layer.get_action(control.AltitudeControlCommand(int(act[i])), float(sp[i])))
    ef[i] = _ENV[layer._state_machine.state]
  # altitude: pressure random walk around the 50 000 ft band for two alphas
  out = dict(env_sp=sp, env_action=act, env_out_action=ea, env_out_fsm=ef)
  for tag, alpha in (('a', 0.2), ('b', 0.85)):
    atm = ref_shims.make_atmosphere(alpha)
    p50 = atm.at_height(altitude_safety.MIN_ALTITUDE).pressure
    pr = p50 + np.cumsum(rng.normal(0, 60, n)) - 300
    pr[300:] = rng.uniform(p50 - 1500, p50 + 500, n - 300)
    a2 = rng.integers(0, 3, n).astype(np.uint8)
    lay = altitude_safety.AltitudeSafetyLayer()
    oa = np.empty(n, np.uint8); of = np.empty(n, np.uint8)
    for i in range(n):
      oa[i] = int(lay.get_action(control.AltitudeControlCommand(int(a2[i])), atm, float(pr[i])))
      of[i] = _ALT[lay._state_machine.state]
    out.update({f'alt_{tag}_alpha': np.float64(alpha), f'alt_{tag}_p': pr, f'alt_{tag}_action': a2,
                f'alt_{tag}_out_action': oa, f'alt_{tag}_out_fsm': of})
  # power: several layers stepped through 3 days at 3 min with a battery trace
  cases = []
  for k, (lat, lng, start) in enumerate([(0.0, 0.0, units.datetime(2021, 8, 26, 0, 43)),
                                         (7.5, -120.0, units.datetime(2012, 3, 3, 17, 20, 11)),
                                         (-9.0, 100.0, units.datetime(2014, 11, 30, 6, 1, 1))]):
    ll = s2.LatLng.from_degrees(lat, lng)
    lay = power_safety.PowerSafetyLayer(ll, start)
    sr0 = int(lay._sunrise_with_hysteresis.timestamp()); ss0 = int(lay._sunset.timestamp())
    m = 1440
    now = int(start.timestamp()) + 180 * np.arange(m)
    batt = np.clip(400 + 350 * np.sin(np.arange(m) / 90.0 + k) + rng.normal(0, 30, m), 1.0, 3058.56)
    a3 = rng.integers(0, 3, m).astype(np.uint8)
    oa = np.empty(m, np.uint8); osr = np.empty(m, np.int64); oss = np.empty(m, np.int64); op = np.empty(m, np.uint8)
    for i in range(m):
      oa[i] = int(lay.get_action(control.AltitudeControlCommand(int(a3[i])),
                                 units.datetime_from_timestamp(int(now[i])), units.Power(watts=183.7),
                                 units.Energy(watt_hours=float(batt[i])), units.Energy(watt_hours=3058.56)))
      osr[i] = int(lay._sunrise_with_hysteresis.timestamp()); oss[i] = int(lay._sunset.timestamp())
      op[i] = int(lay.navigation_is_paused)
    out.update({f'pow_{k}_lat': np.float64(lat), f'pow_{k}_lng': np.float64(lng),
                f'pow_{k}_sunrise_h0': np.int64(sr0), f'pow_{k}_sunset0': np.int64(ss0), f'pow_{k}_now': now,
                f'pow_{k}_batt': batt, f'pow_{k}_action': a3, f'pow_{k}_out_action': oa,
                f'pow_{k}_out_sunrise_h': osr, f'pow_{k}_out_sunset': oss, f'pow_{k}_out_paused': op})
  save('f6_safety', **out)


# ----------------------------------------------------------------------------- F7
def make_field(seed=0, scale=5.0):
  return (np.random.default_rng(seed).standard_normal((21, 21, 10, 9, 2)) * scale).astype(np.float32)


def f7_wind():
  rng = np.random.default_rng(7)
  field = make_field(0)
  wf = ref_shims.make_grid_wind_field(field)
  n = 1024
  x = rng.uniform(-6.5e5, 6.5e5, n); y = rng.uniform(-6.5e5, 6.5e5, n)
  p = rng.uniform(3500, 16000, n); t = rng.integers(0, 200 * 3600, n)
  # exact nodes, edges, boomerang times
  x[:8] = [-500000, 500000, 0, 50000, -450000, 123456.7, 499999.9, -500000.1]
  y[:8] = [500000, -500000, 0, -50000, 450000, -77700.0, 0.0, 12.5]
  p[:8] = [5000, 14000, 9000, 8123.4, 13999.9, 5000.1, 4000, 15000]
  t[:12] = np.array([0, 6, 46, 48, 49, 50, 96, 142, 146, 47.999, 95.5, 192]) * 3600
  u = np.empty(n); v = np.empty(n)
  for i in range(n):
    w = wf.get_forecast(units.Distance(m=float(x[i])), units.Distance(m=float(y[i])), float(p[i]),
                        dt.timedelta(seconds=int(t[i])))
    u[i], v[i] = w.u.mps, w.v.mps
  save('f7_wind', field_seed=np.int64(0), field_scale=np.float64(5.0), field=field, x=x, y=y, pressure=p,
       elapsed_s=t.astype(np.int64), u=u, v=v)


# ----------------------------------------------------------------------------- F8 / F9
STATE_FLOATS = ('x', 'y', 'pressure', 'ambient_temperature', 'internal_temperature', 'envelope_volume',
                'superpressure', 'mols_air', 'battery_charge', 'acs_power', 'acs_mass_flow',
                'solar_charging', 'power_load')
_STATUS = {balloon.BalloonStatus.OK: 0, balloon.BalloonStatus.OUT_OF_POWER: 1, balloon.BalloonStatus.BURST: 2,
           balloon.BalloonStatus.ZEROPRESSURE: 3}


def snapshot(st, start_unix):
  d = dict(x=st.x.meters, y=st.y.meters, pressure=st.pressure, ambient_temperature=st.ambient_temperature,
           internal_temperature=st.internal_temperature, envelope_volume=st.envelope_volume,
           superpressure=st.superpressure, mols_air=st.mols_air, battery_charge=st.battery_charge.watt_hours,
           acs_power=st.acs_power.watts, acs_mass_flow=st.acs_mass_flow, solar_charging=st.solar_charging.watts,
           power_load=st.power_load.watts)
  d = {k: float(v) for k, v in d.items()}
  d.update(time_elapsed_s=int(st.time_elapsed.total_seconds()), status=_STATUS[st.status],
           last_command=int(st.last_command), alt_fsm=_ALT[st.altitude_safety_layer._state_machine.state],
           env_fsm=_ENV[st.envelope_safety_layer._state_machine.state],
           power_paused=int(st.power_safety_layer.navigation_is_paused),
           sunrise_h=int(st.power_safety_layer._sunrise_with_hysteresis.timestamp()),
           sunset=int(st.power_safety_layer._sunset.timestamp()))
  assert int(st.date_time.timestamp()) == start_unix + d['time_elapsed_s']
  return d


SNAP_KEYS = STATE_FLOATS + ('time_elapsed_s', 'status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused',
                            'sunrise_h', 'sunset')


def scenario_list(rng):
  """(lat, lng, start datetime, pressure, x, y, ir, alpha, tweak, script) tuples."""
  sc = []
  starts = [units.datetime(2013, 3, 25, 9, 25, 32), units.datetime(2011, 7, 1, 22, 0, 5),
            units.datetime(2014, 12, 30, 3, 59, 59), units.datetime(2012, 2, 29, 15, 10, 0),
            units.datetime(2013, 9, 21, 17, 45, 0), units.datetime(2013, 9, 22, 5, 20, 0)]
  scripts = ['down', 'stay', 'up', 'cycle', 'random']
  k = 0
  for lat in (-10.0, 0.0, 9.5):
    for start in starts[:4]:
      sc.append(dict(lat=lat, lng=float(rng.uniform(-175, 175)), start=start,
                     pressure=float(rng.uniform(6500, 11000)), x=float(rng.uniform(-2e5, 2e5)),
                     y=float(rng.uniform(-2e5, 2e5)), ir=float(rng.uniform(225, 330)),
                     alpha=float(rng.uniform(0, 1)), tweak=None, script=scripts[k % 5]))
      k += 1
  # around sunset / sunrise on the equator (is_day flips inside the trajectory)
  for start in starts[4:]:
    for script in ('down', 'random'):
      sc.append(dict(lat=0.0, lng=0.0, start=start, pressure=8000.0, x=1000.0, y=-2000.0, ir=250.0, alpha=0.5,
                     tweak=None, script=script))
  # envelope-safety bands: push mols_air to land superpressure near thresholds
  for target_sp, script in ((120.0, 'down'), (230.0, 'down'), (280.0, 'cycle'), (2090.0, 'down'),
                            (2150.0, 'down'), (2250.0, 'stay'), (2370.0, 'stay'), (20.0, 'down')):
    sc.append(dict(lat=3.0, lng=40.0, start=starts[0], pressure=8500.0, x=0.0, y=0.0, ir=260.0, alpha=0.4,
                   tweak=('sp', target_sp), script=script))
  # altitude safety: start low (near / below 50 000 ft)
  for dp_, script in ((-150.0, 'down'), (30.0, 'down'), (400.0, 'stay')):
    sc.append(dict(lat=-4.0, lng=-70.0, start=starts[1], pressure=None, x=5e4, y=5e4, ir=240.0, alpha=0.7,
                   tweak=('p50', dp_), script=script))
  # battery: nearly empty at night (pause + out of power), nearly empty by day
  for batt, start, script in ((90.0, starts[1], 'down'), (2.0, starts[1], 'down'), (0.05, starts[1], 'stay'),
                              (60.0, starts[0], 'down'), (3050.0, starts[0], 'down')):
    sc.append(dict(lat=1.0, lng=10.0, start=start, pressure=9000.0, x=-3e4, y=2e4, ir=255.0, alpha=0.5,
                   tweak=('batt', batt), script=script))
  # far from the station (reward decay) and beyond the wind grid
  sc.append(dict(lat=5.0, lng=170.0, start=starts[3], pressure=7000.0, x=4.9e5, y=-5.2e5, ir=300.0, alpha=0.1,
                 tweak=None, script='random'))
  # partially inflated / zero pressure and burst
  sc.append(dict(lat=0.0, lng=0.0, start=starts[0], pressure=12500.0, x=0.0, y=0.0, ir=230.0, alpha=0.5,
                 tweak=('mols_air', 0.0), script='stay'))
  sc.append(dict(lat=0.0, lng=0.0, start=starts[0], pressure=7000.0, x=0.0, y=0.0, ir=320.0, alpha=0.5,
                 tweak=('sp', 2379.0), script='down'))
  sc.append(dict(lat=0.0, lng=0.0, start=starts[0], pressure=7000.0, x=0.0, y=0.0, ir=320.0, alpha=0.5,
                 tweak=('sp', 2500.0), script='stay'))
  return sc


def build_state(s, atm):
  pressure = s['pressure']
  if s['tweak'] is not None and s['tweak'][0] == 'p50':
    pressure = atm.at_height(altitude_safety.MIN_ALTITUDE).pressure + s['tweak'][1]
  st = balloon.BalloonState(center_latlng=s2.LatLng.from_degrees(s['lat'], s['lng']), date_time=s['start'],
                            x=units.Distance(m=s['x']), y=units.Distance(m=s['y']), pressure=float(pressure),
                            upwelling_infrared=s['ir'])
  stable_init.cold_start_to_stable_params(st, atm)
  if s['tweak'] is not None:
    kind, val = s['tweak']
    if kind == 'sp':
      # choose mols_air so that calculate_superpressure_and_volume gives ~val
      lo, hi = 0.0, 20000.0
      for _ in range(200):
        mid = 0.5 * (lo + hi)
        _, sp = balloon.calculate_superpressure_and_volume(6830.0, mid, st.internal_temperature, st.pressure,
                                                           1804, 0.0199)
        if sp < val:
          lo = mid
        else:
          hi = mid
      st.mols_air = 0.5 * (lo + hi)
      st.envelope_volume, st.superpressure = balloon.calculate_superpressure_and_volume(
          6830.0, st.mols_air, st.internal_temperature, st.pressure, 1804, 0.0199)
    elif kind == 'batt':
      st.battery_charge = units.Energy(watt_hours=val)
    elif kind == 'mols_air':
      st.mols_air = val
      st.envelope_volume, st.superpressure = balloon.calculate_superpressure_and_volume(
          6830.0, st.mols_air, st.internal_temperature, st.pressure, 1804, 0.0199)
  return st


def action_for(script, i, rng):
  return {'down': 0, 'stay': 1, 'up': 2, 'cycle': i % 3, 'random': int(rng.integers(0, 3))}[script]


def f8_trajectories(n_steps=40):
  rng = np.random.default_rng(8)
  scen = scenario_list(rng)
  ns = len(scen)
  cols = {k: np.zeros((ns, n_steps + 1)) for k in STATE_FLOATS}
  for k in ('time_elapsed_s', 'sunrise_h', 'sunset'):
    cols[k] = np.zeros((ns, n_steps + 1), np.int64)
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused'):
    cols[k] = np.zeros((ns, n_steps + 1), np.uint8)
  actions = np.zeros((ns, n_steps), np.uint8); wind = np.zeros((ns, n_steps, 2)); reward = np.zeros((ns, n_steps))
  valid = np.zeros((ns, n_steps), np.uint8)  # 1 where the reference actually stepped (status OK on entry)
  consts = {k: np.zeros(ns) for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha')}
  start_unix = np.zeros(ns, np.int64)
  for j, s in enumerate(scen):
    atm = ref_shims.make_atmosphere(s['alpha'])
    st = build_state(s, atm)
    b = balloon.Balloon(st)
    su = int(s['start'].timestamp()); start_unix[j] = su
    consts['center_lat_deg'][j] = s['lat']; consts['center_lng_deg'][j] = s['lng']
    consts['upwelling_infrared'][j] = s['ir']; consts['alpha'][j] = s['alpha']
    snap = snapshot(b.state, su)
    for k in SNAP_KEYS:
      cols[k][j, 0] = snap[k]
    for i in range(n_steps):
      a = action_for(s['script'], i, rng); actions[j, i] = a
      u, v = rng.normal(0, 8.0, 2); wind[j, i] = (u, v)
      if b.state.status == balloon.BalloonStatus.OK:
        valid[j, i] = 1
        b.simulate_step(wind_field.WindVector(units.Velocity(mps=float(u)), units.Velocity(mps=float(v))), atm,
                        control.AltitudeControlCommand(a), dt.timedelta(minutes=3))
        reward[j, i] = balloon_env.perciatelli_reward_function(
            simulator_data.SimulatorState(b.state, None, atm))
      snap = snapshot(b.state, su)
      for k in SNAP_KEYS:
        cols[k][j, i + 1] = snap[k]
  save('f8_trajectories', actions=actions, wind_uv=wind, reward=reward, valid=valid, start_unix=start_unix,
       **consts, **cols)


def f9_arena(n_env=4, n_steps=60):
  """BalloonArena.step-shaped loop: wind from GridBasedWindField.get_forecast at the pre-step state."""
  rng = np.random.default_rng(9)
  field = make_field(0)
  wf = ref_shims.make_grid_wind_field(field)
  cols = {k: np.zeros((n_env, n_steps + 1)) for k in STATE_FLOATS}
  for k in ('time_elapsed_s', 'sunrise_h', 'sunset'):
    cols[k] = np.zeros((n_env, n_steps + 1), np.int64)
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused'):
    cols[k] = np.zeros((n_env, n_steps + 1), np.uint8)
  actions = rng.integers(0, 3, (n_env, n_steps)).astype(np.uint8)
  wind = np.zeros((n_env, n_steps, 2)); reward = np.zeros((n_env, n_steps))
  consts = {k: np.zeros(n_env) for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha')}
  start_unix = np.zeros(n_env, np.int64)
  starts = [units.datetime(2013, 3, 25, 9, 25, 32), units.datetime(2011, 7, 1, 22, 0, 5),
            units.datetime(2014, 12, 30, 3, 59, 59), units.datetime(2012, 2, 29, 15, 10, 0)]
  for j in range(n_env):
    s = dict(lat=float(rng.uniform(-10, 10)), lng=float(rng.uniform(-175, 175)), start=starts[j % 4],
             pressure=float(rng.uniform(6500, 11000)), x=float(rng.uniform(-2e5, 2e5)),
             y=float(rng.uniform(-2e5, 2e5)), ir=float(rng.uniform(225, 330)), alpha=float(rng.uniform(0, 1)),
             tweak=None)
    atm = ref_shims.make_atmosphere(s['alpha'])
    b = balloon.Balloon(build_state(s, atm))
    su = int(s['start'].timestamp()); start_unix[j] = su
    consts['center_lat_deg'][j] = s['lat']; consts['center_lng_deg'][j] = s['lng']
    consts['upwelling_infrared'][j] = s['ir']; consts['alpha'][j] = s['alpha']
    snap = snapshot(b.state, su)
    for k in SNAP_KEYS:
      cols[k][j, 0] = snap[k]
    for i in range(n_steps):
      w = wf.get_forecast(b.state.x, b.state.y, b.state.pressure, b.state.time_elapsed)
      wind[j, i] = (w.u.mps, w.v.mps)
      b.simulate_step(w, atm, control.AltitudeControlCommand(int(actions[j, i])), dt.timedelta(minutes=3))
      reward[j, i] = balloon_env.perciatelli_reward_function(simulator_data.SimulatorState(b.state, None, atm))
      snap = snapshot(b.state, su)
      for k in SNAP_KEYS:
        cols[k][j, i + 1] = snap[k]
  save('f9_arena', field_seed=np.int64(0), field_scale=np.float64(5.0), actions=actions, wind_uv=wind,
       reward=reward, start_unix=start_unix, **consts, **cols)


# ----------------------------------------------------------------------------- F10
def f10_reset():
  rng = np.random.default_rng(10)
  n = 48
  lat = rng.uniform(-10, 10, n); lng = rng.uniform(-175, 175, n)
  t = rng.integers(UNIX_2011, int(units.datetime(2014, 12, 31).timestamp()), n)
  p = rng.uniform(6500, 11400, n); x = rng.uniform(-2e5, 2e5, n); y = rng.uniform(-2e5, 2e5, n)
  ir = rng.uniform(225, 330, n); alpha = rng.uniform(0, 1, n)
  keys = ('ambient_temperature', 'internal_temperature', 'mols_air', 'envelope_volume', 'superpressure')
  out = {k: np.empty(n) for k in keys}
  sunrise = np.empty(n, np.int64); sunset = np.empty(n, np.int64); blat = np.empty(n); blng = np.empty(n)
  for i in range(n):
    atm = ref_shims.make_atmosphere(float(alpha[i]))
    st = balloon.BalloonState(center_latlng=s2.LatLng.from_degrees(float(lat[i]), float(lng[i])),
                              date_time=units.datetime_from_timestamp(int(t[i])), x=units.Distance(m=float(x[i])),
                              y=units.Distance(m=float(y[i])), pressure=float(p[i]),
                              upwelling_infrared=float(ir[i]))
    # PowerSafetyLayer.__init__ ran in __post_init__ at the balloon's own latlng:
    sunrise[i] = int(st.power_safety_layer._sunrise.timestamp()); sunset[i] = int(st.power_safety_layer._sunset.timestamp())
    blat[i], blng[i] = st.latlng.lat().radians, st.latlng.lng().radians
    stable_init.cold_start_to_stable_params(st, atm)
    for k in keys:
      out[k][i] = getattr(st, k)
  save('f10_reset', center_lat_deg=lat, center_lng_deg=lng, unix_s=t, pressure=p, x=x, y=y,
       upwelling_infrared=ir, alpha=alpha, balloon_lat_rad=blat, balloon_lng_rad=blng, sunrise=sunrise,
       sunset=sunset, **out)


# ----------------------------------------------------------------------------- F11
def f11_features(n_env=3, n_steps=45, name='f11_features', rng_seed=11, keep_last=None):
  """PerciatelliFeatureConstructor (env/features.py:270-581) driven like BalloonArena: observe()
  after every simulate_step, get_features().  The 'measured' wind is forecast + a smooth
  pseudo-noise so that the WindGP (env/wind_gp.py) has non-zero errors to model."""
  from balloon_learning_environment.env import features
  rng = np.random.default_rng(rng_seed)
  field = make_field(0)
  feats = np.zeros((n_env, n_steps + 1, 1099), np.float32)
  cols = {k: np.zeros((n_env, n_steps + 1)) for k in STATE_FLOATS}
  for k in ('time_elapsed_s', 'sunrise_h', 'sunset'):
    cols[k] = np.zeros((n_env, n_steps + 1), np.int64)
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused'):
    cols[k] = np.zeros((n_env, n_steps + 1), np.uint8)
  wind_meas = np.zeros((n_env, n_steps + 1, 2))
  consts = {k: np.zeros(n_env) for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha')}
  start_unix = np.zeros(n_env, np.int64)
  actions = rng.integers(0, 3, (n_env, n_steps)).astype(np.uint8)
  starts = [units.datetime(2013, 3, 25, 9, 25, 32), units.datetime(2011, 7, 1, 22, 0, 5), units.datetime(2013, 9, 21, 17, 45, 0)]
  for j in range(n_env):
    s = dict(lat=float(rng.uniform(-10, 10)), lng=float(rng.uniform(-175, 175)), start=starts[j % 3],
             pressure=float(rng.uniform(7000, 10500)), x=float(rng.uniform(-1.5e5, 1.5e5)),
             y=float(rng.uniform(-1.5e5, 1.5e5)), ir=float(rng.uniform(230, 320)), alpha=float(rng.uniform(0, 1)), tweak=None)
    atm = ref_shims.make_atmosphere(s['alpha'])
    wf = ref_shims.make_grid_wind_field(field)
    b = balloon.Balloon(build_state(s, atm))
    su = int(s['start'].timestamp()); start_unix[j] = su
    consts['center_lat_deg'][j] = s['lat']; consts['center_lng_deg'][j] = s['lng']
    consts['upwelling_infrared'][j] = s['ir']; consts['alpha'][j] = s['alpha']
    fc = features.PerciatelliFeatureConstructor(wf, atm)

    def measure(i):
      w = wf.get_forecast(b.state.x, b.state.y, b.state.pressure, b.state.time_elapsed)
      nu, nv = 1.5 * np.sin(0.3 * i + j), -1.0 * np.cos(0.17 * i) + 0.2 * j
      wind_meas[j, i] = (w.u.mps + nu, w.v.mps + nv)
      return w, wind_field.WindVector(units.Velocity(mps=float(wind_meas[j, i, 0])), units.Velocity(mps=float(wind_meas[j, i, 1])))

    for i in range(n_steps + 1):
      if i > 0:
        w, _ = measure(i - 1)
        b.simulate_step(w, atm, control.AltitudeControlCommand(int(actions[j, i - 1])), dt.timedelta(minutes=3))
      _, meas = measure(i)
      fc.observe(simulator_data.SimulatorObservation(balloon_observation=b.state, wind_at_balloon=meas))
      feats[j, i] = fc.get_features()
      snap = snapshot(b.state, su)
      for k in SNAP_KEYS:
        cols[k][j, i] = snap[k]
  if keep_last is not None:       # long runs: keep the feature vectors of the last steps only
    feats = feats[:, -keep_last:]
  save(name, field_seed=np.int64(0), field_scale=np.float64(5.0), features=feats, wind_measured=wind_meas,
       actions=actions, start_unix=start_unix, **consts, **cols)


def f12_features_long():
  """As F11 but 135 steps (> the WindGP's 6 h / 120-observation horizon, wind_gp.py:179-185):
  pins the dropping of old observations.  Only the last 16 feature vectors are stored."""
  f11_features(n_env=1, n_steps=135, name='f12_features_long', rng_seed=12, keep_last=16)


# ----------------------------------------------------------------------------- F13
def f13_station_seeker_episode(n_steps=960):
  """BASELINE.json configs[0] in closed loop: the reference's StationSeekerAgent
  (agents/station_seeker_agent.py:72-86) picks every action from the reference's
  PerciatelliFeatureConstructor output; the balloon flies BalloonArena.step's loop
  (env/balloon_arena.py:184-202) for micro_eval's 960 steps (eval/suites.py:43): ground-truth
  wind (forecast + noise) at the PRE-step state -> simulate_step -> observe with the ground truth at
  the POST-step state.  The noise is a smooth function of the step index (opensimplex is absent).
  Stored: every action, every state, the 1099-vector of every step (float32, as the reference
  returns it), the agent's chosen level."""
  from balloon_learning_environment.agents import station_seeker_agent
  from balloon_learning_environment.env import features
  rng = np.random.default_rng(13)
  field = make_field(0)
  wf = ref_shims.make_grid_wind_field(field)
  s = dict(lat=float(rng.uniform(-10, 10)), lng=float(rng.uniform(-175, 175)), start=units.datetime(2013, 3, 25, 9, 25, 32),
           pressure=float(rng.uniform(7000, 10500)), x=float(rng.uniform(-1.0e5, 1.0e5)), y=float(rng.uniform(-1.0e5, 1.0e5)),
           ir=float(rng.uniform(230, 320)), alpha=float(rng.uniform(0, 1)), tweak=None)
  atm = ref_shims.make_atmosphere(s['alpha'])
  b = balloon.Balloon(build_state(s, atm))
  su = int(s['start'].timestamp())
  fc = features.PerciatelliFeatureConstructor(wf, atm)
  agent = station_seeker_agent.StationSeekerAgent(3, (1099,))
  feats = np.zeros((n_steps + 1, 1099), np.float32)
  cols = {k: np.zeros((1, n_steps + 1)) for k in STATE_FLOATS}
  for k in ('time_elapsed_s', 'sunrise_h', 'sunset'):
    cols[k] = np.zeros((1, n_steps + 1), np.int64)
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused'):
    cols[k] = np.zeros((1, n_steps + 1), np.uint8)
  wind_truth = np.zeros((1, n_steps + 1, 2)); noise = np.zeros((1, n_steps + 1, 2))
  actions = np.zeros((1, n_steps), np.uint8); levels = np.zeros(n_steps + 1, np.int32); reward = np.zeros((1, n_steps))
  valid = np.zeros((1, n_steps), np.uint8)

  def truth(i):
    w = wf.get_forecast(b.state.x, b.state.y, b.state.pressure, b.state.time_elapsed)
    noise[0, i] = (1.5 * np.sin(0.3 * i), -1.0 * np.cos(0.17 * i) + 0.2)
    wind_truth[0, i] = (w.u.mps + noise[0, i, 0], w.v.mps + noise[0, i, 1])
    return wind_field.WindVector(units.Velocity(mps=float(wind_truth[0, i, 0])), units.Velocity(mps=float(wind_truth[0, i, 1])))

  def observe(i):
    fc.observe(simulator_data.SimulatorObservation(balloon_observation=b.state, wind_at_balloon=truth(i)))
    feats[i] = fc.get_features()
    snap = snapshot(b.state, su)
    for k in SNAP_KEYS:
      cols[k][0, i] = snap[k]

  observe(0)
  last = n_steps
  for i in range(n_steps):
    named = features.NamedPerciatelliFeatures(feats[i])
    levels[i], _ = agent.find_best_pressure_level(named)
    a = agent.pick_action(feats[i]); actions[0, i] = a
    if b.state.status != balloon.BalloonStatus.OK:
      last = i
      break
    valid[0, i] = 1
    w = wind_field.WindVector(units.Velocity(mps=float(wind_truth[0, i, 0])), units.Velocity(mps=float(wind_truth[0, i, 1])))
    b.simulate_step(w, atm, control.AltitudeControlCommand(int(a)), dt.timedelta(minutes=3))
    reward[0, i] = balloon_env.perciatelli_reward_function(simulator_data.SimulatorState(b.state, None, atm))
    observe(i + 1)
  print(f'f13: flew {last} steps, final status {int(b.state.status.value) if hasattr(b.state.status, "value") else b.state.status}, '
        f'actions {np.bincount(actions[0, :last], minlength=3)}, mean reward {reward[0, :last].mean():.3f}')
  consts = dict(center_lat_deg=np.array([s['lat']]), center_lng_deg=np.array([s['lng']]),
                upwelling_infrared=np.array([s['ir']]), alpha=np.array([s['alpha']]))
  save('f13_station_seeker', field_seed=np.int64(0), field_scale=np.float64(5.0), features=feats[None], wind_measured=wind_truth,
       noise_uv=noise, actions=actions, levels=levels, reward=reward, valid=valid, n_flown=np.int64(last),
       start_unix=np.array([su], np.int64), **consts, **cols)


# ----------------------------------------------------------------------------- F14
def f14_wind_noise(n_points=192):
  """The reference's wind-noise COMPOSITION (env/simplex_wind_noise.py:82-211, env/wind_field.py:113-145,187-218) around a
  stand-in primitive: `opensimplex.OpenSimplex` is oracle/ref_shims.py::_StandInSimplex (the kernel's simplex4, NOT
  opensimplex 0.3), the jax.random draws of NoisyWindHarmonic.reset are recorded.  Stored: the seeds / offsets the
  reference's own reset() put into its harmonics, and get_wind_noise / get_ground_truth at seeded points.  Pins the
  harmonic tables, spacings, offsets, NOISE_MAGNITUDE and the variance adjustment; the primitive stays unpinned."""
  from balloon_learning_environment.env import simplex_wind_noise
  import noise_oracle                    # (the stand-in primitive's home: its version goes into the fixture)
  rng = np.random.default_rng(14)
  episodes = 3
  x = rng.uniform(-4e5, 4e5, (episodes, n_points)); y = rng.uniform(-4e5, 4e5, (episodes, n_points))
  p = rng.uniform(5000.0, 14000.0, (episodes, n_points)); t = rng.integers(0, 60 * 3600, (episodes, n_points))
  x[:, 0] = 0.0; y[:, 0] = 0.0; t[:, 0] = 0
  seeds = np.zeros((episodes, 2, 5), np.int64); offsets = np.zeros((episodes, 2, 5, 4)); noise = np.zeros((episodes, n_points, 2))
  truth = np.zeros((episodes, n_points, 2)); forecast = np.zeros((episodes, n_points, 2))
  field = make_field()
  for e in range(episodes):
    wf = ref_shims.make_grid_wind_field(field)
    wf._noise_model = wind_field.SimplexWindNoise()
    wf._noise_model.reset_wind_noise(np.array([14, e], np.uint32), None)
    for c, comp in enumerate((wf._noise_model.noise_u, wf._noise_model.noise_v)):
      for h, harm in enumerate(comp._harmonics):
        seeds[e, c, h] = harm._simplex_generator.seed
        offsets[e, c, h] = (harm._offsets.x, harm._offsets.y, harm._offsets.pressure, harm._offsets.time)
    for i in range(n_points):
      args = (units.Distance(meters=float(x[e, i])), units.Distance(meters=float(y[e, i])), float(p[e, i]), dt.timedelta(seconds=int(t[e, i])))
      w = wf._noise_model.get_wind_noise(*args)
      noise[e, i] = (w.u.meters_per_second, w.v.meters_per_second)
      g = wf.get_ground_truth(*args); f = wf.get_forecast(*args)
      truth[e, i] = (g.u.meters_per_second, g.v.meters_per_second); forecast[e, i] = (f.u.meters_per_second, f.v.meters_per_second)
  assert seeds.max() < 1634753849 and np.abs(offsets).max() <= 1.0
  save('f14_wind_noise', x=x, y=y, pressure=p, elapsed_s=t.astype(np.int64), seeds=seeds, offsets=offsets, noise=noise,
       ground_truth=truth, forecast=forecast, field_seed=np.int64(0), field_scale=np.float64(5.0),
       noise_magnitude=np.float64(simplex_wind_noise.NOISE_MAGNITUDE), noise_primitive_version=np.int64(noise_oracle.PRIMITIVE_VERSION))


# ----------------------------------------------------------------------------- F15
def f15_decoder(n_samples=2):
  """The reference's VAE decoder (generative/vae.py:140-186) and GenerativeWindFieldSampler.sample_field
  (env/generative_wind_field.py:49-62), imported unmodified, on synthetic weights (tests/helpers.py::
  hashed_decoder_params(seed 15); the trained ones are absent) with the flax / jax stand-ins of oracle/ref_shims.py.  Pins
  the layer order and activations, the (7, 7, 90) reshape, the roll / slice order of the central differences, u = d psi /
  d axis0, v = -d psi / d axis1, the (21, 21, 10, 9) reshape and the stack axis; the resize operator is the assumption."""
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import helpers
  from balloon_learning_environment.env import generative_wind_field as gwf
  from balloon_learning_environment.generative import vae
  params = helpers.hashed_decoder_params(seed=15)
  variables = {'params': {f'Dense_{k}': {'kernel': w, 'bias': b} for k, (w, b) in enumerate(params)}}
  sampler = gwf.GenerativeWindFieldSampler.__new__(gwf.GenerativeWindFieldSampler)
  sampler.params = variables
  import jax
  latents = np.zeros((n_samples, 64), np.float32); fields = np.zeros((n_samples, 21, 21, 10, 9, 2)); flow = np.zeros((n_samples, 4410))
  for i in range(n_samples):
    key = np.array([15, i], np.uint32)
    latents[i] = jax.random.normal(key, shape=(64,))
    fields[i] = sampler.sample_field(key, None)
    z = np.asarray(latents[i], np.float64)                       # the MLP output alone (vae.py:145-148), for a per-stage check
    for k, (w, b) in enumerate(params):
      z = z @ np.asarray(w, np.float64) + np.asarray(b, np.float64)
      if k < 3:
        z = np.maximum(z, 0.0)
    flow[i] = z
  assert fields.shape[1:] == vae.FieldShape().grid_shape() and np.isfinite(fields).all()
  print(f'f15: |wind| max {np.abs(fields).max():.2f} m/s, rms {np.sqrt((fields ** 2).mean()):.2f}')
  save('f15_decoder', param_seed=np.int64(15), latents=latents, flow=flow, fields=fields.astype(np.float32))


# ----------------------------------------------------------------------------- F16
# BalloonState's flight-vehicle constants are dataclass FIELDS (balloon.py:156-173,183) and power_safety_layer_enabled a
# per-state switch (:200,305): vehicles other than the default one, flown by the reference's own Balloon.simulate_step.
F16_VEHICLE_FIELDS = ('envelope_volume_base', 'envelope_volume_dv_pressure', 'envelope_mass', 'envelope_max_superpressure',
                      'envelope_cod', 'payload_mass', 'nighttime_power_load_w', 'daytime_power_load_w',
                      'acs_valve_hole_diameter_m', 'battery_capacity_wh', 'mols_lift_gas', 'power_safety_layer_enabled')
F16_VEHICLES = (
    # a larger envelope with more lift gas and a weaker skin
    dict(envelope_volume_base=2100.0, envelope_volume_dv_pressure=0.025, envelope_mass=75.0, envelope_max_superpressure=2000.0,
         payload_mass=100.0, mols_lift_gas=7800.0),
    # a smaller, draggier one with a stronger skin
    dict(envelope_volume_base=1500.0, envelope_volume_dv_pressure=0.015, envelope_mass=60.0, envelope_max_superpressure=2600.0,
         envelope_cod=0.3, payload_mass=80.0, mols_lift_gas=5700.0),
    # the default envelope with another power system and a wider valve
    dict(nighttime_power_load_w=250.0, daytime_power_load_w=150.0, battery_capacity_wh=2000.0, acs_valve_hole_diameter_m=0.05),
    # the default vehicle with the power safety layer switched off
    dict(power_safety_layer_enabled=0),
    # everything at once
    dict(envelope_volume_base=1950.0, envelope_volume_dv_pressure=0.0215, envelope_mass=71.25, envelope_max_superpressure=2200.0,
         envelope_cod=0.22, payload_mass=88.0, nighttime_power_load_w=160.5, daytime_power_load_w=131.0,
         acs_valve_hole_diameter_m=0.035, battery_capacity_wh=3500.0, mols_lift_gas=7300.0, power_safety_layer_enabled=0),
)


def f16_vehicle_kwargs(v):
  """A dict of F16_VEHICLE_FIELDS -> the keyword arguments of the reference's BalloonState."""
  kw = {}
  for k, val in v.items():
    if k == 'nighttime_power_load_w':
      kw['nighttime_power_load'] = units.Power(watts=val)
    elif k == 'daytime_power_load_w':
      kw['daytime_power_load'] = units.Power(watts=val)
    elif k == 'acs_valve_hole_diameter_m':
      kw['acs_valve_hole_diameter'] = units.Distance(m=val)
    elif k == 'battery_capacity_wh':
      kw['battery_capacity'] = units.Energy(watt_hours=val)
    elif k == 'power_safety_layer_enabled':
      kw[k] = bool(val)
    else:
      kw[k] = val
  return kw


def f16_vehicles(n_steps=40):
  rng = np.random.default_rng(16)
  defaults = balloon.BalloonState(center_latlng=s2.LatLng.from_degrees(0.0, 0.0), date_time=units.datetime(2013, 3, 25, 9))
  default_row = [defaults.envelope_volume_base, defaults.envelope_volume_dv_pressure, defaults.envelope_mass,
                 defaults.envelope_max_superpressure, defaults.envelope_cod, defaults.payload_mass,
                 defaults.nighttime_power_load.watts, defaults.daytime_power_load.watts, defaults.acs_valve_hole_diameter.meters,
                 defaults.battery_capacity.watt_hours, defaults.mols_lift_gas, float(defaults.power_safety_layer_enabled)]
  vehicles = np.tile(np.array(default_row, np.float64), (len(F16_VEHICLES), 1))
  for i, v in enumerate(F16_VEHICLES):
    for k, val in v.items():
      vehicles[i, F16_VEHICLE_FIELDS.index(k)] = val
  day, night = units.datetime(2013, 3, 25, 9, 25, 32), units.datetime(2011, 7, 1, 22, 0, 5)
  dusk = units.datetime(2013, 9, 21, 17, 45, 0)
  scen = []
  for vi in range(len(F16_VEHICLES)):
    cap = vehicles[vi, F16_VEHICLE_FIELDS.index('battery_capacity_wh')]
    for start, script, batt in ((day, 'down', None), (night, 'cycle', None), (dusk, 'random', None), (day, 'up', None),
                                (night, 'down', 0.035 * cap), (night, 'down', 0.004 * cap), (day, 'down', 0.995 * cap)):
      scen.append(dict(vehicle=vi, lat=float(rng.uniform(-10, 10)), lng=float(rng.uniform(-175, 175)), start=start,
                       pressure=float(rng.uniform(7000, 10500)), x=float(rng.uniform(-1.5e5, 1.5e5)),
                       y=float(rng.uniform(-1.5e5, 1.5e5)), ir=float(rng.uniform(230, 320)), alpha=float(rng.uniform(0, 1)),
                       script=script, batt=batt))
  ns = len(scen)
  cols = {k: np.zeros((ns, n_steps + 1)) for k in STATE_FLOATS}
  for k in ('time_elapsed_s', 'sunrise_h', 'sunset'):
    cols[k] = np.zeros((ns, n_steps + 1), np.int64)
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused'):
    cols[k] = np.zeros((ns, n_steps + 1), np.uint8)
  actions = np.zeros((ns, n_steps), np.uint8); wind = np.zeros((ns, n_steps, 2)); reward = np.zeros((ns, n_steps))
  valid = np.zeros((ns, n_steps), np.uint8)
  consts = {k: np.zeros(ns) for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha')}
  start_unix = np.zeros(ns, np.int64); vehicle_index = np.zeros(ns, np.int64)
  cold = {k: np.zeros(ns) for k in ('ambient_temperature', 'internal_temperature', 'mols_air', 'envelope_volume', 'superpressure')}
  for j, s in enumerate(scen):
    atm = ref_shims.make_atmosphere(s['alpha'])
    st = balloon.BalloonState(center_latlng=s2.LatLng.from_degrees(s['lat'], s['lng']), date_time=s['start'],
                              x=units.Distance(m=s['x']), y=units.Distance(m=s['y']), pressure=s['pressure'],
                              upwelling_infrared=s['ir'], **f16_vehicle_kwargs(F16_VEHICLES[s['vehicle']]))
    stable_init.cold_start_to_stable_params(st, atm)                   # (the vehicle's own cold start: stable_init.py:132-157)
    for k in cold:
      cold[k][j] = getattr(st, k)
    assert 0.0 < st.superpressure < st.envelope_max_superpressure, (j, st.superpressure)
    if s['batt'] is not None:
      st.battery_charge = units.Energy(watt_hours=s['batt'])
    b = balloon.Balloon(st)
    su = int(s['start'].timestamp()); start_unix[j] = su; vehicle_index[j] = s['vehicle']
    consts['center_lat_deg'][j] = s['lat']; consts['center_lng_deg'][j] = s['lng']
    consts['upwelling_infrared'][j] = s['ir']; consts['alpha'][j] = s['alpha']
    snap = snapshot(b.state, su)
    for k in SNAP_KEYS:
      cols[k][j, 0] = snap[k]
    for i in range(n_steps):
      a = action_for(s['script'], i, rng); actions[j, i] = a
      u, v = rng.normal(0, 8.0, 2); wind[j, i] = (u, v)
      if b.state.status == balloon.BalloonStatus.OK:
        valid[j, i] = 1
        b.simulate_step(wind_field.WindVector(units.Velocity(mps=float(u)), units.Velocity(mps=float(v))), atm,
                        control.AltitudeControlCommand(a), dt.timedelta(minutes=3))
        reward[j, i] = balloon_env.perciatelli_reward_function(simulator_data.SimulatorState(b.state, None, atm))
      snap = snapshot(b.state, su)
      for k in SNAP_KEYS:
        cols[k][j, i + 1] = snap[k]
  # ---- the observation of such a vehicle: PerciatelliFeatureConstructor (features.py:301-581) reads battery_soc (capacity),
  # excess_energy (daytime load, capacity) and get_pressure_range (volume base, dV/dp, masses, lift gas, maximum superpressure)
  from balloon_learning_environment.env import features
  field = make_field(0)
  n_obs = 8
  nv = len(F16_VEHICLES)
  obs_feats = np.zeros((nv, n_obs + 1, 1099), np.float32)
  obs_cols = {k: np.zeros((nv, n_obs + 1)) for k in STATE_FLOATS}
  for k in ('time_elapsed_s', 'sunrise_h', 'sunset'):
    obs_cols[k] = np.zeros((nv, n_obs + 1), np.int64)
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused'):
    obs_cols[k] = np.zeros((nv, n_obs + 1), np.uint8)
  obs_wind = np.zeros((nv, n_obs + 1, 2)); obs_actions = rng.integers(0, 3, (nv, n_obs)).astype(np.uint8)
  obs_consts = {k: np.zeros(nv) for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha')}
  obs_start = np.zeros(nv, np.int64)
  for j in range(nv):
    so = dict(lat=float(rng.uniform(-10, 10)), lng=float(rng.uniform(-175, 175)), start=(day, night, dusk)[j % 3],
              pressure=float(rng.uniform(7500, 10000)), x=float(rng.uniform(-1.5e5, 1.5e5)), y=float(rng.uniform(-1.5e5, 1.5e5)),
              ir=float(rng.uniform(230, 320)), alpha=float(rng.uniform(0, 1)))
    atm = ref_shims.make_atmosphere(so['alpha'])
    wf = ref_shims.make_grid_wind_field(field)
    st = balloon.BalloonState(center_latlng=s2.LatLng.from_degrees(so['lat'], so['lng']), date_time=so['start'],
                              x=units.Distance(m=so['x']), y=units.Distance(m=so['y']), pressure=so['pressure'],
                              upwelling_infrared=so['ir'], **f16_vehicle_kwargs(F16_VEHICLES[j]))
    stable_init.cold_start_to_stable_params(st, atm)
    if j % 2 == 0:       # a nearly full battery by day: excess_energy depends on the vehicle's capacity and daytime load
      st.battery_charge = units.Energy(watt_hours=0.995 * st.battery_capacity.watt_hours)
    b = balloon.Balloon(st)
    su = int(so['start'].timestamp()); obs_start[j] = su
    obs_consts['center_lat_deg'][j] = so['lat']; obs_consts['center_lng_deg'][j] = so['lng']
    obs_consts['upwelling_infrared'][j] = so['ir']; obs_consts['alpha'][j] = so['alpha']
    fc = features.PerciatelliFeatureConstructor(wf, atm)
    for i in range(n_obs + 1):
      if i > 0:
        w = wf.get_forecast(b.state.x, b.state.y, b.state.pressure, b.state.time_elapsed)
        b.simulate_step(w, atm, control.AltitudeControlCommand(int(obs_actions[j, i - 1])), dt.timedelta(minutes=3))
      w = wf.get_forecast(b.state.x, b.state.y, b.state.pressure, b.state.time_elapsed)
      obs_wind[j, i] = (w.u.mps + 1.2 * np.sin(0.4 * i + j), w.v.mps - 0.8 * np.cos(0.23 * i) + 0.1 * j)
      fc.observe(simulator_data.SimulatorObservation(
          balloon_observation=b.state, wind_at_balloon=wind_field.WindVector(units.Velocity(mps=float(obs_wind[j, i, 0])),
                                                                            units.Velocity(mps=float(obs_wind[j, i, 1])))))
      obs_feats[j, i] = fc.get_features()
      snap = snapshot(b.state, su)
      for k in SNAP_KEYS:
        obs_cols[k][j, i] = snap[k]
  obs = dict(obs_features=obs_feats, obs_wind_measured=obs_wind, obs_actions=obs_actions, obs_start_unix=obs_start, field_seed=np.int64(0),
             field_scale=np.float64(5.0), **{'obs_' + k: v for k, v in obs_consts.items()}, **{'obs_' + k: v for k, v in obs_cols.items()})
  print(f'f16: {ns} trajectories over {len(F16_VEHICLES)} vehicles, {int(valid.sum())} reference steps, final status counts '
        f'{np.bincount(cols["status"][:, -1], minlength=4).tolist()}, paused at some step: {int((cols["power_paused"].max(axis=1) > 0).sum())}')
  save('f16_vehicles', vehicles=vehicles, vehicle_fields=np.array(F16_VEHICLE_FIELDS), vehicle_index=vehicle_index, actions=actions,
       wind_uv=wind, reward=reward, valid=valid, start_unix=start_unix, **{'cold_' + k: v for k, v in cold.items()}, **consts, **cols, **obs)


# ----------------------------------------------------------------------------- F17
def f17_static_wind_features(n_env=2, n_steps=14):
  """PerciatelliFeatureConstructor (env/features.py:270-581) over a forecast that is NOT a grid: the reference's unit-test field,
  SimpleStaticWindField (env/wind_field.py:149-184: four sheets blowing E / N / W / S by pressure band -- a step function of
  pressure).  The constructor asks the WindField for its column above the balloon (features.py:499-503); balloons fly in the
  forecast, the 'measured' wind adds a smooth pseudo-noise so that the WindGP has errors to model."""
  from balloon_learning_environment.env import features
  rng = np.random.default_rng(17)
  feats = np.zeros((n_env, n_steps + 1, 1099), np.float32)
  cols = {k: np.zeros((n_env, n_steps + 1)) for k in STATE_FLOATS}
  for k in ('time_elapsed_s', 'sunrise_h', 'sunset'):
    cols[k] = np.zeros((n_env, n_steps + 1), np.int64)
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused'):
    cols[k] = np.zeros((n_env, n_steps + 1), np.uint8)
  wind_meas = np.zeros((n_env, n_steps + 1, 2)); forecast_at = np.zeros((n_env, n_steps + 1, 2))
  consts = {k: np.zeros(n_env) for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha')}
  start_unix = np.zeros(n_env, np.int64)
  actions = rng.integers(0, 3, (n_env, n_steps)).astype(np.uint8)
  starts = [units.datetime(2013, 3, 25, 9, 25, 32), units.datetime(2011, 7, 1, 22, 0, 5)]
  wf = wind_field.SimpleStaticWindField.__new__(wind_field.SimpleStaticWindField)      # (no SimplexWindNoise: get_forecast only)
  for j in range(n_env):
    s = dict(lat=float(rng.uniform(-10, 10)), lng=float(rng.uniform(-175, 175)), start=starts[j % 2],
             pressure=(7900.0, 9950.0)[j % 2], x=float(rng.uniform(-1e5, 1e5)), y=float(rng.uniform(-1e5, 1e5)),
             ir=float(rng.uniform(230, 320)), alpha=float(rng.uniform(0, 1)), tweak=None)     # (next to a sheet boundary)
    atm = ref_shims.make_atmosphere(s['alpha'])
    b = balloon.Balloon(build_state(s, atm))
    su = int(s['start'].timestamp()); start_unix[j] = su
    consts['center_lat_deg'][j] = s['lat']; consts['center_lng_deg'][j] = s['lng']
    consts['upwelling_infrared'][j] = s['ir']; consts['alpha'][j] = s['alpha']
    fc = features.PerciatelliFeatureConstructor(wf, atm)
    for i in range(n_steps + 1):
      if i > 0:
        w = wf.get_forecast(b.state.x, b.state.y, b.state.pressure, b.state.time_elapsed)
        b.simulate_step(w, atm, control.AltitudeControlCommand(int(actions[j, i - 1])), dt.timedelta(minutes=3))
      w = wf.get_forecast(b.state.x, b.state.y, b.state.pressure, b.state.time_elapsed)
      forecast_at[j, i] = (w.u.mps, w.v.mps)
      wind_meas[j, i] = (w.u.mps + 1.1 * np.sin(0.35 * i + j), w.v.mps - 0.9 * np.cos(0.21 * i) + 0.15 * j)
      fc.observe(simulator_data.SimulatorObservation(
          balloon_observation=b.state, wind_at_balloon=wind_field.WindVector(units.Velocity(mps=float(wind_meas[j, i, 0])),
                                                                            units.Velocity(mps=float(wind_meas[j, i, 1])))))
      feats[j, i] = fc.get_features()
      snap = snapshot(b.state, su)
      for k in SNAP_KEYS:
        cols[k][j, i] = snap[k]
  bands = sorted({(float(a), float(c)) for a, c in forecast_at.reshape(-1, 2)})
  print(f'f17: forecast sheets visited {bands}')
  save('f17_static_wind_features', features=feats, wind_measured=wind_meas, forecast_at_balloon=forecast_at, actions=actions,
       start_unix=start_unix, **consts, **cols)


if __name__ == '__main__':
  which = sys.argv[1:] or ['all']
  if 'all' in which:
    f1_atmosphere(); f2_solar(); f3_thermal(); f4_sp_volume(); f5_acs_power_table(); f6_safety(); f7_wind()
    f8_trajectories(); f9_arena(); f10_reset(); f11_features(); f12_features_long()
  if 'all' in which or 'f13' in which:
    f13_station_seeker_episode()
  if 'all' in which or 'f14' in which:
    f14_wind_noise()
  if 'all' in which or 'f15' in which:
    f15_decoder()
  if 'all' in which or 'f16' in which:
    f16_vehicles()
  if 'all' in which or 'f17' in which:
    f17_static_wind_features()
