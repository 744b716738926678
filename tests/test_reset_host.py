"""Host reset path (NumPy) against the reference's golden vectors (F10, F1, F2) and the oracle.  CPU only."""
import time

import numpy as np

import oracle
import reset_host as rh
from helpers import golden, known_answers, unix


def test_atmosphere_tables_match_reference():
  d = golden('f1_atmosphere')
  atm = rh.AtmosphereTables(d['alphas'])
  np.testing.assert_allclose(atm.pres, d['pressure_transitions'], rtol=1e-13)
  np.testing.assert_allclose(atm.temp, d['temperature_transitions'], rtol=1e-13)
  for j in range(0, d['pressures'].size, 5):
    if d['pressures'][j] < 1.0:
      continue
    h, t = atm.at_pressure(np.full(d['alphas'].size, d['pressures'][j]))
    np.testing.assert_allclose(h, d['h_of_p'][:, j], rtol=1e-11)
    np.testing.assert_allclose(t, d['t_of_p'][:, j], rtol=1e-12)
  for j in range(d['heights'].size):
    p, t = atm.at_height(d['heights'][j])
    np.testing.assert_allclose(p, d['p_of_h'][:, j], rtol=1e-12)


def test_solar_calculator_matches_reference():
  d = golden('f2_solar')
  el, flux = rh.solar_calculator(d['lat_rad'], d['lng_rad'], d['unix_s'])
  np.testing.assert_allclose(el, d['el'], rtol=0, atol=1e-9)
  np.testing.assert_allclose(flux, d['flux'], rtol=1e-12)
  lat, lng = rh.latlng_from_offset(d['off_lat0'], d['off_lng0'], d['off_x'], d['off_y'])
  np.testing.assert_allclose(lat, d['off_lat'], atol=1e-14)
  np.testing.assert_allclose(lng, d['off_lng'], atol=1e-14)


def test_sunrise_sunset_known_answers_and_golden():
  for now, sr, ss in known_answers()['sunrise_sunset']['cases']:
    a, b = rh.next_sunrise_sunset(np.array([0.0]), np.array([0.0]), np.array([unix(now)]))
    assert a[0] == unix(sr) and b[0] == unix(ss)
  d = golden('f10_reset')
  sr, ss = rh.next_sunrise_sunset(d['balloon_lat_rad'], d['balloon_lng_rad'], d['unix_s'])
  np.testing.assert_array_equal(sr, d['sunrise']); np.testing.assert_array_equal(ss, d['sunset'])


def test_stable_init_matches_reference():
  d = golden('f10_reset')
  atm = rh.AtmosphereTables(d['alpha'])
  out = rh.stable_params(d['pressure'], d['balloon_lat_rad'], d['balloon_lng_rad'], d['unix_s'],
                         d['upwelling_infrared'], atm)
  for k, v in out.items():
    np.testing.assert_allclose(v, d[k], rtol=1e-9, atol=1e-9, err_msg=k)
  f3 = golden('f3_thermal')
  got = rh.d_balloon_temperature_dt(f3['volume'], 68.5, f3['t_int'], f3['t_amb'], f3['pressure'], f3['el'],
                                    f3['flux'], f3['ir'])
  np.testing.assert_allclose(got, f3['dtdt'], rtol=1e-11, atol=1e-16)


def test_sample_initial_state_is_consistent_with_oracle():
  n = 2048
  t0 = time.time()
  st = rh.sample_initial_state(n, seed=3)
  dt = time.time() - t0
  assert dt < 60
  # ranges of balloon_arena_test.py:56-87 / sampling
  r = np.hypot(st['x'], st['y'])
  assert r.max() <= 200_000.0 + 1 and (st['pressure'] >= 6500).all()
  atm = rh.AtmosphereTables(st['alpha'])
  assert (st['pressure'] <= atm.at_height(rh.MIN_ALTITUDE_M)[0] + 1e-2).all()
  assert (st['upwelling_infrared'] >= 225.0).all() and (np.abs(st['center_lat_deg']) <= 10).all()
  # the oracle's stable_init / sunrise search agree on the same inputs
  out, err = oracle.stable_init(st['pressure'], st['center_lat_deg'], st['center_lng_deg'], st['x'], st['y'],
                                st['start_unix'], st['upwelling_infrared'], st['alpha'])
  assert err == 0
  for k, v in out.items():
    np.testing.assert_allclose(st[k], v, rtol=1e-9, atol=1e-9, err_msg=k)
  lat, lng = oracle.latlng_from_offset(np.radians(st['center_lat_deg']), np.radians(st['center_lng_deg']), st['x'], st['y'])
  sr, ss = oracle.next_sunrise_sunset(lat, lng, st['start_unix'])
  np.testing.assert_array_equal(st['sunrise_h_rel'], sr + 1800 - st['start_unix'])
  np.testing.assert_array_equal(st['sunset_rel'], ss - st['start_unix'])
  # determinism
  st2 = rh.sample_initial_state(n, seed=3)
  for k in st:
    np.testing.assert_array_equal(st[k], st2[k])
