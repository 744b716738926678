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
