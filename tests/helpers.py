"""Shared test helpers (golden loading, state conversion, tolerances)."""
import datetime as dt
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
  return np.load(os.path.join(GOLDEN, name + '.npz'))


def known_answers():
  with open(os.path.join(GOLDEN, 'reference_known_answers.json')) as f:
    return json.load(f)


def unix(iso):
  return int(dt.datetime.fromisoformat(iso).replace(tzinfo=dt.timezone.utc).timestamp())


STATE_FLOATS = ('x', 'y', 'pressure', 'ambient_temperature', 'internal_temperature', 'envelope_volume',
                'superpressure', 'mols_air', 'battery_charge', 'acs_power', 'acs_mass_flow',
                'solar_charging', 'power_load')
STATE_INTS = ('time_elapsed_s', 'sunrise_h', 'sunset')
STATE_U8 = ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused')
CONSTS = ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha')

# Absolute floors that turn "1e-5 relative" into a usable bound for fields that pass
# through zero (position, ACS outputs, ...).  |a-b| <= rtol * max(|b|, floor).
FLOORS = dict(x=1000.0, y=1000.0, pressure=1.0, ambient_temperature=1.0, internal_temperature=1.0,
              envelope_volume=1.0, superpressure=100.0, mols_air=100.0, battery_charge=100.0,
              acs_power=10.0, acs_mass_flow=1e-3, solar_charging=10.0, power_load=10.0)


def rel_err(a, b, floor):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  return np.abs(a - b) / np.maximum(np.abs(b), floor)


def traj_state_at(d, step, rows=None):
  """Oracle-typed state dict from a trajectory fixture (f8/f9) at `step`."""
  import oracle
  n = d['x'].shape[0]
  rows = np.arange(n) if rows is None else rows
  st = oracle.new_state(len(rows))
  for k in STATE_FLOATS:
    st[k][:] = d[k][rows, step]
  for k in STATE_INTS:
    st[k][:] = d[k][rows, step]
  for k in STATE_U8:
    st[k][:] = d[k][rows, step]
  for k in CONSTS:
    st[k][:] = d[k][rows]
  st['start_unix'][:] = d['start_unix'][rows]
  return st


def feature_row(g, j, i):
  """Row dict (fields of ble_state_f32, float64) of env j at step i of a features fixture (F11/F12)."""
  row = {k: float(g[k][j, i]) for k in STATE_FLOATS}
  for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s'):
    row[k] = int(g[k][j, i])
  for k in ('center_lat_deg', 'center_lng_deg', 'upwelling_infrared', 'alpha'):
    row[k] = float(g[k][j])
  row['start_unix'] = int(g['start_unix'][j])
  row['sunrise_h_rel'] = int(g['sunrise_h'][j, i] - g['start_unix'][j])
  row['sunset_rel'] = int(g['sunset'][j, i] - g['start_unix'][j])
  return row


def fixture_field(g):
  import numpy as np
  return (np.random.default_rng(int(g['field_seed'])).standard_normal((21, 21, 10, 9, 2)) * float(g['field_scale'])).astype(np.float32)


def wide_domain_states(n, seed):
  """ABI-typed initial states far outside the flight envelope the samplers draw (tests of the transition's robustness):
  pressures 1 200 .. 40 000 Pa (above the 21 km window and below the troposphere's top), stations up to 85 deg of latitude
  (polar day and night),
  balloons up to 850 km from the station (beyond the wind grid), up to 110 h into the episode (the boomeranged part of the
  forecast, later table segments of everything time-based), any temperatures / infrared / battery, safety layers in any
  state.  The envelope is consistent: a superpressure drawn in 20 .. 2 300 Pa fixes volume and air content."""
  import reset_host
  rng = np.random.default_rng(seed)
  init = reset_host.sample_initial_state(n, seed=seed)
  init['pressure'][:] = np.exp(rng.uniform(np.log(1200.0), np.log(40000.0), n))
  init['x'][:] = rng.uniform(-600e3, 600e3, n); init['y'][:] = rng.uniform(-600e3, 600e3, n)
  init['center_lat_deg'][:] = rng.uniform(-85, 85, n); init['center_lng_deg'][:] = rng.uniform(-180, 180, n)
  init['time_elapsed_s'][:] = rng.integers(0, 2200, n) * 180
  init['upwelling_infrared'][:] = rng.uniform(150, 400, n)
  init['battery_charge'][:] = rng.uniform(5, 3058, n)
  init['internal_temperature'][:] = rng.uniform(180, 300, n); init['ambient_temperature'][:] = rng.uniform(180, 280, n)
  sp = rng.uniform(20, 2300, n); vol = 1804.0 + 0.0199 * sp
  p = init['pressure'].astype(np.float64); t_int = init['internal_temperature'].astype(np.float64)
  init['mols_air'][:] = np.maximum((p + sp) * vol / (8.3144621 * t_int) - 6830.0, 0.0)
  init['envelope_volume'][:] = vol; init['superpressure'][:] = sp
  init['alt_fsm'][:] = rng.integers(0, 3, n); init['env_fsm'][:] = rng.integers(0, 5, n); init['power_paused'][:] = rng.integers(0, 2, n)
  return init


def hashed_decoder_params(seed=0, hidden=1000, output_gain=30.0):
  """Synthetic weights of the wind-field VAE decoder (generative/vae.py:140-148: 64 -> hidden x 3 -> 4410, ReLU) for
  fixture F15: [(kernel [in, out] float32, bias [out] float32)] x 4.  A pure integer hash of (layer, row, column, seed)
  -> uniform(-1, 1) scaled like LeCun initialisation, NON-zero biases -- no dependence on any library's random streams,
  so the generator (build container) and the test (GPU box) always rebuild the same numbers."""
  dims = [64, hidden, hidden, hidden, 7 * 7 * 90]
  out = []
  for layer, (a, b) in enumerate(zip(dims, dims[1:])):
    def unit(rows, cols, salt):
      with np.errstate(over='ignore'):
        h = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(0x9E3779B97F4A7C15) +
             np.arange(cols, dtype=np.uint64)[None, :] * np.uint64(0xC2B2AE3D27D4EB4F) +
             np.uint64((seed * 1000003 + layer * 7919 + salt) & 0xFFFFFFFF))
        h ^= h >> np.uint64(29); h *= np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
        h *= np.uint64(0x94D049BB133111EB); h ^= h >> np.uint64(29)
      return (h >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0      # uniform [-1, 1)
    gain = output_gain if layer == 3 else np.sqrt(2.0)
    kernel = (unit(a, b, 1) * np.sqrt(3.0 / a) * gain).astype(np.float32)
    bias = (unit(1, b, 2)[0] * (0.5 if layer == 3 else 0.05)).astype(np.float32)
    out.append((kernel, bias))
  return out


def noise_cache_from_draws(seeds, offsets, n, seed, episode=0):
  """The `harmonic_cache` of ble_wind_noise_f32 / ble_noise_gen ([53][n] 32-bit words) holding GIVEN generator seeds
  [2][5] and offsets [2][5][4] for all n environments -- as if they had been drawn for (seed, episode): rows 5 k .. 5 k + 4 =
  (seed, ox, oy, op, ot) of harmonic k = 5 comp + h, rows 50 .. 52 the key (episode + 1, seed lo, seed hi)."""
  c = np.zeros((53, n), np.uint32)
  for comp in range(2):
    for h in range(5):
      k = 5 * comp + h
      c[5 * k] = np.uint32(int(seeds[comp][h]) & 0xFFFFFFFF)
      c[5 * k + 1:5 * k + 5] = np.asarray(offsets[comp][h], np.float32).view(np.uint32)[:, None]
  c[50] = np.uint32(episode + 1); c[51] = np.uint32(seed & 0xFFFFFFFF); c[52] = np.uint32((seed >> 32) & 0xFFFFFFFF)
  return c


def fixture_vehicle(d, vi):
  """Vehicle `vi` of the F16 fixture as a dict of the fields that DIFFER from the reference's defaults (the keyword form of
  oracle.step(vehicle=...) and VecSimulator.set_vehicle(...)); {} for the default vehicle."""
  import oracle
  names = [str(k) for k in d['vehicle_fields']]
  out = {}
  for k, v in zip(names, d['vehicles'][vi]):
    v = int(v) if k == 'power_safety_layer_enabled' else float(v)
    if v != oracle.VEHICLE_DEFAULTS[k]:
      out[k] = v
  return out
