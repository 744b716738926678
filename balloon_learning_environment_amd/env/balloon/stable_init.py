"""env/balloon/stable_init.py:32-157 of the reference by the reset kernel's Newton cold start (`ble_reset_f32`, sample = 0)."""
import dataclasses

from balloon_learning_environment_amd.env.balloon import _probes
from balloon_learning_environment_amd.env.balloon import balloon


@dataclasses.dataclass
class StableParams:
  ambient_temperature: float
  internal_temperature: float
  mols_air: float
  envelope_volume: float
  superpressure: float


def calculate_stable_params_for_pressure(pressure: float, envelope_volume_base: float, envelope_volume_dv_pressure: float,
                                         envelope_mass: float, payload_mass: float, mols_lift_gas: float, latlng, date_time,
                                         upwelling_infrared: float, atmosphere) -> StableParams:
  """stable_init.py:40-129: ambient and internal temperature, mols of air, envelope volume and superpressure at which a vehicle of
  these constants floats at `pressure` -- the reset kernel's Newton cold start on one environment (`ble_reset_f32`, sample = 0,
  `ble_state_f32.vehicle`)."""
  out = _probes.reset_one(dict(x=0.0, y=0.0, pressure=pressure, center_lat_deg=latlng.lat().degrees, center_lng_deg=latlng.lng().degrees,
                               upwelling_infrared=upwelling_infrared, alpha=float(atmosphere.alpha), start_unix=int(date_time.timestamp())),
                          vehicle=dict(envelope_volume_base=envelope_volume_base, envelope_volume_dv_pressure=envelope_volume_dv_pressure,
                                       envelope_mass=envelope_mass, payload_mass=payload_mass, mols_lift_gas=mols_lift_gas))
  return StableParams(float(out['ambient_temperature']), float(out['internal_temperature']), float(out['mols_air']),
                      float(out['envelope_volume']), float(out['superpressure']))


def cold_start_to_stable_params(balloon_state: 'balloon.BalloonState', atmosphere) -> None:
  """Sets ambient / internal temperature, mols_air, envelope volume and superpressure of `balloon_state` to the values at
  which it floats at its pressure (stable_init.py:132-157), in place -- with the state's own vehicle constants."""
  row = balloon.row_from_state(balloon_state, float(atmosphere.alpha))
  out = _probes.reset_one({k: row[k] for k in ('x', 'y', 'pressure', 'center_lat_deg', 'center_lng_deg', 'upwelling_infrared',
                                               'alpha')} | {'start_unix': row['start_unix'] + row['time_elapsed_s']},
                          vehicle=balloon.vehicle_of(balloon_state))
  balloon_state.ambient_temperature = float(out['ambient_temperature'])
  balloon_state.internal_temperature = float(out['internal_temperature'])
  balloon_state.mols_air = float(out['mols_air'])
  balloon_state.envelope_volume = float(out['envelope_volume'])
  balloon_state.superpressure = float(out['superpressure'])
