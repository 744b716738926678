"""One-element calls into the device functions of the transition (`ble_probe_*`, include/ble_abi.h): what the per-function
mirror modules next to this file (standard_atmosphere, solar, thermal, acs, stable_init) are made of.  Each call runs
exactly the lane function `ble_step_f32` uses, on a batch of one, and reads the result back.  Needs a HIP device."""
import numpy as np
import torch

from balloon_learning_environment_amd import _lib
from balloon_learning_environment_amd import device as dev
from balloon_learning_environment_amd import vec_state


def _f32(*values):
  return [torch.tensor([float(v)], dtype=torch.float32, device='cuda') for v in values]


def _out(n=1, dtype=torch.float32):
  return torch.empty(n, dtype=dtype, device='cuda')


def _stream():
  return torch.cuda.current_stream().cuda_stream


def _flags():
  return torch.zeros(1, dtype=torch.int32, device='cuda')


def _raise(flags):
  code = int(flags.item())
  if code:
    vec_state.raise_for_flags(code)


def atmosphere(alpha, pressure):
  """Atmosphere.at_pressure -> (height m, temperature K); raises like the reference outside the model's range."""
  dev.require_gpu('cuda')
  a, p = _f32(alpha, pressure)
  h, t, flags = _out(), _out(), _flags()
  _lib.check(_lib.lib().ble_probe_atmosphere_f32(a.data_ptr(), p.data_ptr(), h.data_ptr(), t.data_ptr(), flags.data_ptr(), 1,
                                                 _stream()), 'ble_probe_atmosphere_f32')
  _raise(flags)
  return float(h.item()), float(t.item())


def atmosphere_at_height(alpha, height_m):
  """Atmosphere.at_height -> (pressure Pa, temperature K), float64 on the device (`ble_probe_atmosphere_at_height_f64`); raises like the
  reference outside the model's range."""
  dev.require_gpu('cuda')
  a, = _f32(alpha)
  h = torch.tensor([float(height_m)], dtype=torch.float64, device='cuda')
  out, flags = torch.empty(2, dtype=torch.float64, device='cuda'), _flags()
  _lib.check(_lib.lib().ble_probe_atmosphere_at_height_f64(a.data_ptr(), h.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(), flags.data_ptr(), 1,
                                                           _stream()), 'ble_probe_atmosphere_at_height_f64')
  _raise(flags)
  p, t = out.cpu().tolist()
  return p, t


def latlng(center_lat_deg, center_lng_deg, x_m, y_m):
  """BalloonState.latlng: (lat deg, lng deg) of the point (x, y) metres east / north of the centre."""
  dev.require_gpu('cuda')
  la, lo, x, y = _f32(center_lat_deg, center_lng_deg, x_m, y_m)
  out = torch.empty(2, dtype=torch.float64, device='cuda')
  _lib.check(_lib.lib().ble_probe_latlng_f64(la.data_ptr(), lo.data_ptr(), x.data_ptr(), y.data_ptr(), out[0:1].data_ptr(),
                                             out[1:2].data_ptr(), 1, _stream()), 'ble_probe_latlng_f64')
  lat, lng = out.cpu().tolist()
  return lat, lng


def atmosphere_column(alpha, pressures):
  """Atmosphere.at_pressure for a vector of pressures (no range check: out-of-range entries come back as they are) ->
  (heights m, temperatures K) as float64 NumPy arrays of the device's float32 results."""
  dev.require_gpu('cuda')
  p = torch.as_tensor(np.asarray(pressures, np.float32), device='cuda')
  a = torch.full_like(p, float(alpha))
  h, t, flags = torch.empty_like(p), torch.empty_like(p), _flags()
  _lib.check(_lib.lib().ble_probe_atmosphere_f32(a.data_ptr(), p.data_ptr(), h.data_ptr(), t.data_ptr(), flags.data_ptr(), p.numel(),
                                                 _stream()), 'ble_probe_atmosphere_f32')
  return h.cpu().numpy().astype(np.float64), t.cpu().numpy().astype(np.float64)


def power_table(pressure_ratio, state_of_charge):
  """power_table.lookup -> watts; raises like the reference outside [0.99, 5]."""
  dev.require_gpu('cuda')
  r, c = _f32(pressure_ratio, state_of_charge)
  w, flags = _out(), _flags()
  _lib.check(_lib.lib().ble_power_table_f32(r.data_ptr(), c.data_ptr(), w.data_ptr(), flags.data_ptr(), 1, _stream()), 'ble_power_table_f32')
  _raise(flags)
  return float(w.item())


def solar(lat_deg, lng_deg, unix_s, x_m=0.0, y_m=0.0):
  """solar_calculator at the point (x, y) metres east / north of (lat, lng) -> (refraction-corrected elevation deg, flux W/m^2)."""
  dev.require_gpu('cuda')
  la, lo, x, y = _f32(lat_deg, lng_deg, x_m, y_m)
  t = torch.tensor([int(unix_s)], dtype=torch.int64, device='cuda')
  el, flux = _out(), _out()
  _lib.check(_lib.lib().ble_probe_solar_f32(la.data_ptr(), lo.data_ptr(), x.data_ptr(), y.data_ptr(), t.data_ptr(), el.data_ptr(),
                                            flux.data_ptr(), 1, _stream()), 'ble_probe_solar_f32')
  return float(el.item()), float(flux.item())


def solar_power(el_deg, pressure):
  """-> (attenuation, panel power W)."""
  dev.require_gpu('cuda')
  e, p = _f32(el_deg, pressure)
  att, pw = _out(), _out()
  _lib.check(_lib.lib().ble_probe_solar_power_f32(e.data_ptr(), p.data_ptr(), att.data_ptr(), pw.data_ptr(), 1, _stream()),
             'ble_probe_solar_power_f32')
  return float(att.item()), float(pw.item())


def _vehicle_ptr(vehicle):
  """{vehicle field: value} (or None) -> (argument for a `const ble_vehicle*` parameter, the struct to keep alive)."""
  import ctypes
  from balloon_learning_environment_amd import _abi
  veh = _abi.vehicle_struct(**(vehicle or {}))
  return (None if veh is None else ctypes.byref(veh)), veh


def thermal(volume, t_int, t_amb, pressure, el_deg, flux, earth_flux, envelope_mass=68.5):
  """d_balloon_temperature_dt [K/s] for an envelope of `envelope_mass` kg (`ble_probe_thermal_vehicle_f32`)."""
  dev.require_gpu('cuda')
  args = _f32(volume, t_int, t_amb, pressure, el_deg, flux, earth_flux)
  out, flags = _out(), _flags()
  ptr, _keep = _vehicle_ptr({'envelope_mass': envelope_mass})
  _lib.check(_lib.lib().ble_probe_thermal_vehicle_f32(ptr, *[a.data_ptr() for a in args], out.data_ptr(), flags.data_ptr(), 1, _stream()),
             'ble_probe_thermal_vehicle_f32')
  _raise(flags)
  return float(out.item())


def acs(pressure_ratio):
  """-> (most efficient power W, fan efficiency at that power, mass flow kg/s)."""
  dev.require_gpu('cuda')
  pr, = _f32(pressure_ratio)
  w, eff, md = _out(), _out(), _out()
  _lib.check(_lib.lib().ble_probe_acs_f32(pr.data_ptr(), w.data_ptr(), eff.data_ptr(), md.data_ptr(), 1, _stream()), 'ble_probe_acs_f32')
  return float(w.item()), float(eff.item()), float(md.item())


def sp_volume(mols_air, t_int, pressure, vehicle=None):
  """calculate_superpressure_and_volume -> (envelope volume m^3, superpressure Pa) for the vehicle's lift gas / volume base / dV/dp."""
  dev.require_gpu('cuda')
  a = _f32(mols_air, t_int, pressure)
  v, sp = _out(), _out()
  ptr, _keep = _vehicle_ptr(vehicle)
  _lib.check(_lib.lib().ble_probe_sp_volume_vehicle_f32(ptr, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), v.data_ptr(), sp.data_ptr(), 1,
                                                        _stream()), 'ble_probe_sp_volume_vehicle_f32')
  return float(v.item()), float(sp.item())


def reset_one(row: dict, vehicle=None):
  """`ble_reset_f32(sample = 0)` on ONE environment whose position, pressure, centre, IR, alpha and start time are those of
  `row`: the Newton cold start (stable_init.py:132-157) of the vehicle `vehicle` ({field: value}; None = the reference's defaults)
  and PowerSafetyLayer's sunrise / sunset search (solar.py:432-483).  Returns the state row after the reset (python scalars)."""
  sim = vec_state.VecSimulator(1)
  sim.set_vehicle(**(vehicle or {}))
  sim.set_state({k: np.array([v]) for k, v in row.items()})
  sim.reset_device(seed=0, sample=False)
  sim.check_errors()
  return {k: t[0].item() for k, t in sim.state.items()}


def safety(layer, action, value, fsm, alpha=0.0, clocks=None, night_load_w=183.7, capacity_wh=3058.56, max_superpressure=None):
  """One call of one safety layer (`ble_probe_safety_f32`): layer 0 altitude (value = pressure), 1 envelope (value =
  superpressure; max_superpressure: the layer's own, None = 2 380 Pa), 2 power (value = battery Wh, clocks = (now, sunrise +
  30 min, sunset) in seconds from a common epoch).  -> (effective action, new fsm byte, clocks after the call or None)."""
  if layer == 1:
    alpha = max_superpressure
  dev.require_gpu('cuda')
  a = torch.tensor([int(action)], dtype=torch.uint8, device='cuda')
  v, al = _f32(value, 0.0 if alpha is None else alpha)
  al_ptr = None if (layer == 1 and alpha is None) else al.data_ptr()
  state = torch.tensor([int(fsm)], dtype=torch.uint8, device='cuda')
  ck = torch.tensor([list(clocks) if clocks is not None else [0, 0, 0]], dtype=torch.int32, device='cuda')
  eff, flags = _out(dtype=torch.uint8), _flags()
  _lib.check(_lib.lib().ble_probe_safety_f32(int(layer), a.data_ptr(), v.data_ptr(), al_ptr, ck.data_ptr(), float(night_load_w),
                                             float(capacity_wh), state.data_ptr(), eff.data_ptr(), flags.data_ptr(), 1, _stream()),
             'ble_probe_safety_f32')
  _raise(flags)
  return int(eff.item()), int(state.item()), (tuple(ck[0].tolist()) if clocks is not None else None)
