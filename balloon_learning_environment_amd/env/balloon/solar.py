"""env/balloon/solar.py of the reference, function by function, on the transition's device functions.

`solar_calculator`, `solar_atmospheric_attenuation`, `solar_power` run the kernel's lane functions on one element
(`ble_probe_solar_f32`, `ble_probe_solar_power_f32`); `get_next_sunrise_sunset` runs the reset kernel's search
(`ble_reset_f32`, sample = 0) for a balloon parked at the site."""
import datetime as dt
import math
from typing import Tuple

from balloon_learning_environment_amd.env.balloon import _probes
from balloon_learning_environment_amd.utils import units

MIN_SOLAR_EL_DEG = -4.242          # solar.py:38
_BALLOON_SHADOW_MAX_R = 8.69275    # solar.py:214-216


def solar_calculator(latlng, time: dt.datetime) -> Tuple[float, float, float]:
  """(elevation deg, azimuth deg, flux W/m^2), solar.py:43-174.  The azimuth is not computed anywhere on the transition's
  path (its callers read the elevation and the flux): it comes back as NaN."""
  lat, lng = _lat_lng_deg(latlng)
  el, flux = _probes.solar(lat, lng, int(time.timestamp()))
  return el, float('nan'), flux


def solar_atmospheric_attenuation(el_deg: float, pressure_altitude_pa: float) -> float:
  """solar.py:177-209 (ValueError outside [0, 101 325] Pa like :194-197; 0 below MIN_SOLAR_EL_DEG like :199-200)."""
  if pressure_altitude_pa > 101325.0 or pressure_altitude_pa < 0.0:
    raise ValueError('solar_atmospheric_attenuation: Pressure altitude out of expected range [0, 101325] Pa.')
  return _probes.solar_power(el_deg, pressure_altitude_pa)[0]


def balloon_shadow(el_deg: float, panel_height_below_balloon_m: float) -> float:
  """solar.py:212-236: 0.4392 when the panels sit inside the envelope's shadow cone, else 1."""
  shadow_el = math.degrees(math.atan2(math.sqrt(panel_height_below_balloon_m * (10.41603 + panel_height_below_balloon_m)),
                                      _BALLOON_SHADOW_MAX_R))
  return 0.4392 if el_deg >= shadow_el else 1.0


def solar_power(el_deg: float, pressure_altitude_pa: float) -> units.Power:
  """solar.py:515-536."""
  return units.Power(watts=_probes.solar_power(el_deg, pressure_altitude_pa)[1])


def get_next_sunrise_sunset(latlng, time: dt.datetime) -> Tuple[dt.datetime, dt.datetime]:
  """solar.py:432-483, by the reset kernel's search for a balloon parked at `latlng` at `time`."""
  start = int(time.timestamp())
  lat, lng = _lat_lng_deg(latlng)
  row = dict(x=0.0, y=0.0, pressure=9000.0, center_lat_deg=lat, center_lng_deg=lng,
             upwelling_infrared=250.0, alpha=0.5, start_unix=start)
  out = _probes.reset_one(row)
  return (units.datetime_from_timestamp(start + int(out['sunrise_h_rel']) - 1800),     # the layer keeps sunrise + 30 min
          units.datetime_from_timestamp(start + int(out['sunset_rel'])))


def _lat_lng_deg(latlng):
  """(lat, lng) in degrees of this package's LatLng (`lat_deg`, `lng_deg`) or of an s2sphere-style one (`lat().degrees`)."""
  if hasattr(latlng, 'lat_deg'):
    return float(latlng.lat_deg), float(latlng.lng_deg)
  return float(latlng.lat().degrees), float(latlng.lng().degrees)
