// ble_reset.h -- per-lane episode reset, fp64, one lane per environment.
//
// Device counterpart of BalloonArena.reset / _initialize_balloon
// (env/balloon_arena.py:161-182,228-268):
//   * draws of the initial conditions (utils/sampling.py:37-152) from a counter-based
//     Philox4x32-10 stream keyed by (seed, environment, episode) -- the reference draws from
//     JAX threefry (absent, parity unpinned), so only the DISTRIBUTIONS match;
//   * the Newton cold start (env/balloon/stable_init.py:40-157);
//   * PowerSafetyLayer.__init__'s sunrise / sunset search
//     (env/balloon/power_safety.py:40-41 -> env/balloon/solar.py:239-483).
// The deterministic part (cold start, sunrise search) is parity-tested against the oracle:
// the search compares elevations of neighbouring 3-minute samples near the solar
// extremum, where they differ by ~1e-5 deg, so the solar calculator here is full fp64
// (the mixed-precision one of the transition kernel is not good enough for it).
// Runs once per episode (960 agent steps), off the hot path.
#pragma once
#include "ble_physics.h"

namespace ble {

// d_asin, solar_elevation_f64 and latlng_f64 (the full fp64 solar calculator) live in ble_physics.h: the
// transition kernel uses them too, on the rare strides where a solar threshold is within its fp32 floor.

// ---------------------------------------------------------------- sunrise / sunset search
struct SunSite { double sin_lat, cos_lat, lng_deg; };
BLE_FN double site_elevation(const SunSite& g, int64_t t) {
  return solar_elevation_f64(g.sin_lat, g.cos_lat, g.lng_deg, t, nullptr);
}
// solar._find_solar_elevation_binary_search (solar.py:295-372); mode 0 min, 1 max, 2 |el - target|.
// `elev(t)` returns the elevation [deg] at unix time t: evaluated directly (reset kernel) or
// looked up in a table that a whole workgroup filled in parallel (observation kernel) -- every
// time the search touches is start + 180 s * integer.
template <class Elev>
BLE_FN int64_t find_solar_elevation(const Elev& elev, int64_t min_t, int64_t max_t, int mode, double target) {
  const int64_t dt = 180;
  int64_t low = 0, high = (max_t - min_t) / dt;
  auto obj = [&](int64_t idx) {
    const double el = elev(min_t + dt * idx);
    return mode == 0 ? el : (mode == 1 ? -el : fabs(el - target));
  };
  double ol = obj(low), oh = obj(high);
#pragma unroll 1
  while (high > low + 1) {
    const int64_t span = high - low;           // midpoint = low + span / 2.0
    if (ol < oh) { high = low + (span + 1) / 2; oh = obj(high); }   // ceil
    else { low = low + span / 2; ol = obj(low); }                   // floor
  }
  return min_t + dt * ((ol < oh) ? low : high);
}
// solar.get_next_sunrise_sunset (solar.py:432-483); `afternoon` = el(t + 1 s) < el(t) (:239-256)
template <class Elev>
BLE_FN void next_sunrise_sunset_from(const Elev& elev, bool afternoon, int64_t t, int64_t* sunrise, int64_t* sunset) {
  const int64_t h12 = 12 * 3600, h24 = 24 * 3600;
  const int64_t noon = afternoon ? find_solar_elevation(elev, t + h12, t + h24, 1, 0.0)
                                 : find_solar_elevation(elev, t, t + h12, 1, 0.0);
  const int64_t midnight = afternoon ? find_solar_elevation(elev, t, t + h12, 0, 0.0)
                                     : find_solar_elevation(elev, t + h12, t + h24, 0, 0.0);
  int64_t sr = find_solar_elevation(elev, afternoon ? midnight : midnight - h24, noon, 2, -4.242);
  int64_t ss = find_solar_elevation(elev, afternoon ? noon - h24 : noon, midnight, 2, -4.242);
  if (sr < t) sr += h24;
  if (ss < t) ss += h24;
  *sunrise = sr; *sunset = ss;
}
BLE_FN void next_sunrise_sunset(const SunSite& g, int64_t t, int64_t* sunrise, int64_t* sunset) {
  auto elev = [&](int64_t when) { return site_elevation(g, when); };
  next_sunrise_sunset_from(elev, site_elevation(g, t + 1) < site_elevation(g, t), t, sunrise, sunset);
}

// solar.solar_power (solar.py:515-536) in fp64; shadow thresholds are
// degrees(atan2(sqrt(h (10.41603 + h)), 8.69275)) for the panels 3.3 m and 2.7 m below the envelope.
BLE_FN double solar_attenuation_f64(double el_deg, double p);
BLE_FN double solar_power_f64(double el_deg, double p) {
  double s35, c35, s65, c65;
  sincos_f64((el_deg - 35.0) * (kPiD / 180.0), &s35, &c35);
  sincos_f64((el_deg - 65.0) * (kPiD / 180.0), &s65, &c65);
  const double shadow_a = el_deg >= 37.738149050524044 ? 0.4392 : 1.0;
  const double shadow_b = el_deg >= 34.39486500086289 ? 0.4392 : 1.0;
  return 210.0 * solar_attenuation_f64(el_deg, p) * (4 * c35 * shadow_a + 2 * c65 * shadow_b);
}

// ---------------------------------------------------------------- cold start (stable_init.py:40-129)
BLE_FN double solar_attenuation_f64(double el_deg, double p) {   // solar.py:177-209
  if (el_deg < -4.242) return 0.0;
  double s, c;
  sincos_f64(el_deg * (kPiD / 180.0), &s, &c);
  const double t = 614.0 * s;
  const double airmass = 0.34764 * (p / 101325.0) * (sqrt(1229.0 + t * t) - t);
  return 0.5 * (d_exp_fast(-0.65 * airmass) + d_exp_fast(-0.95 * airmass));
}
// d_balloon_temperature_dt for the cold start: the transition's pow-free fp64 model (ble_physics.h), with the
// twelfth root in fp64 as well.  yc = V^(-1/3).
BLE_FN double thermal_dtdt_f64(double volume, double yc, double t_int, double t_amb, double p, double att, double flux,
                               double q_earth_area, double thermal_scale = VehicleDefault::thermal_scale) {
  constexpr double kSolarAbs = 0.01435 * (1.0 + (1.0 - 0.01435 - 0.0291) / (1.0 - 0.0291));
  return 0.1 * thermal_increment_f64<true>(volume, yc, t_int, t_amb, p, flux * att * (0.25 * kSolarAbs), q_earth_area, stride_k_literal(),
                                           thermal_scale);
}
struct StableParams { double t_amb, t_int, mols_air, volume, sp; };
// calculate_stable_params_for_pressure (stable_init.py:40-129) for the vehicle `veh` (ble_physics.h::VehicleDefault / VehicleRt)
template <class V = VehicleDefault>
BLE_FN StableParams stable_params(double alpha, double p, double el_deg, double flux, double ir, uint32_t* flags, const V& veh = V()) {
  StableParams o;
  const AtmWindow w = atm_window(alpha, p, flags);
  double h;
  atm_at_pressure_f64(w, alpha, p, &h, &o.t_amb);
  double ma = ((p * kAirMolarMassD * veh.v0 / (kGasConstantD * o.t_amb) - veh.envelope_mass - veh.payload_mass - veh.he_mass) /
               kAirMolarMassD);
  o.mols_air = ma > 0.0 ? ma : 0.0;
  const double att = solar_attenuation_f64(el_deg, p);
  double ti = 206.0;
  const double delta = 0.01;
  uint32_t ignored = 0;                                     // total_absorptivity of the Earth term: checked by the transition
  const double q_earth = earth_heat_per_area_f64(ir, &ignored);
#pragma unroll 1
  for (int k = 0; k < 10; ++k) {
    const double d1 = thermal_dtdt_f64(veh.v0, veh.inv_cbrt_v0, ti - delta / 2, o.t_amb, p, att, flux, q_earth, veh.thermal_scale);
    const double d2 = thermal_dtdt_f64(veh.v0, veh.inv_cbrt_v0, ti + delta / 2, o.t_amb, p, att, flux, q_earth, veh.thermal_scale);
    const double d2t = (d2 - d1) / delta;
    const double mean = (d1 + d2) / 2.0;
    if (fabs(d2t) > 0.0) ti -= mean / d2t;
    if (fabs(mean) < 1e-5) break;
  }
  o.t_int = ti;
  superpressure_volume_f64(o.mols_air, ti, p, 1.0 / p, &o.volume, &o.sp, stride_k_literal(veh.dry_mass, veh.lift, veh.v0), veh.dvdp, veh.four_dvdp,
                           veh.inv_dvdp);
  return o;
}

// ---------------------------------------------------------------- Philox4x32-10 + distributions
struct Philox {
  uint32_t key0, key1, c0, c1, c2, c3;
  uint32_t out[4];
  int have;
};
BLE_FN void philox_round(uint32_t* c, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
BLE_FN void philox_refill(Philox& g) {
  uint32_t c[4] = {g.c0, g.c1, g.c2, g.c3};
  uint32_t k0 = g.key0, k1 = g.key1;
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  g.out[0] = c[0]; g.out[1] = c[1]; g.out[2] = c[2]; g.out[3] = c[3];
  g.have = 4;
  if (++g.c0 == 0) ++g.c1;
}
BLE_FN Philox philox_init(uint64_t seed, uint64_t env, uint32_t episode) {
  Philox g;
  g.key0 = (uint32_t)seed; g.key1 = (uint32_t)(seed >> 32);
  g.c0 = 0; g.c1 = episode; g.c2 = (uint32_t)env; g.c3 = (uint32_t)(env >> 32);
  g.have = 0;
  return g;
}
BLE_FN uint32_t philox_u32(Philox& g) {
  if (g.have == 0) philox_refill(g);
  // (selects, not g.out[--g.have]: a run-time index into the buffer put the whole generator in scratch memory -- 48 B per lane in
  // every kernel that draws inside a rolled loop: the noise kernel, the four-wave kernel with the noise generator)
  const int h = --g.have;
  return h == 3 ? g.out[3] : (h == 2 ? g.out[2] : (h == 1 ? g.out[1] : g.out[0]));
}
BLE_FN double philox_uniform(Philox& g) {   // [0, 1), 53 bits
  const uint64_t hi = philox_u32(g), lo = philox_u32(g);
  return (double)(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
BLE_FN double philox_normal(Philox& g) {    // Box-Muller
  const double u1 = 1.0 - philox_uniform(g), u2 = philox_uniform(g);
  double s, c;
  sincos_f64(2.0 * kPiD * u2, &s, &c);
  return sqrt(-2.0 * d_log_fast(u1)) * c;
}
BLE_FN double philox_gamma(Philox& g, double shape) {   // Marsaglia-Tsang, shape >= 1
  const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    const double x = philox_normal(g);
    const double t = 1.0 + c * x;
    if (t <= 0.0) continue;
    const double v = t * t * t;
    const double u = 1.0 - philox_uniform(g);
    if (d_log_fast(u) < 0.5 * x * x + d - d * v + d * d_log_fast(v)) return d * v;
  }
  return d;
}

}  // namespace ble
