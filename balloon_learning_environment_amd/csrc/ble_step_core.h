// ble_step_core.h -- one agent step of one environment, in registers.
//
// Balloon.simulate_step (env/balloon/balloon.py:263-328) with _simulate_step_internal
// (:356-549) inlined, preceded by the three safety layers (:304-313) and followed by
// perciatelli_reward_function (env/balloon_env.py:44-102).  Called by the HIP kernel in
// ble_kernels.hip with one lane per environment; also compiled on the host by tests/emul.
//
// Precision map.  The reference's vertical dynamics are an UNSTABLE map near float
// equilibrium: dh/dt = +-sqrt(|rho V - m| ...) has unbounded gain where rho V - m -> 0 and
// |gain| ~ 1.5-3 per 10 s substep in the quasi-steady regime, so a 1e-7 relative error in
// anything that feeds rho V - m (p, T_amb(p), V(n_air, T_int, p), the thermal and ACS increments)
// grows to O(1 Pa) within one agent step.  That chain -- p, T_amb, T_int, n_air, V, superpressure,
// the difference itself, dh/dt, 1/dH, the thermal increment (thermal_increment_f64) and the ACS mass
// flow (acs_down_poly / the valve formula) -- is carried in fp64 registers across the 18 substeps
// (inputs and outputs stay fp32).  fp64 too: the three solar nodes of a step, the safety layers'
// comparisons.  fp32: the wind blend, refraction, attenuation, panel power, battery, reward -- they feed
// nothing that is amplified.
#pragma once
#include "ble_physics.h"

namespace ble {

// The DEVICE pointers of ble_state_f32 (include/ble_abi.h: every member but the host-only `vehicle`), in its order: what the kernels take
// as an argument.  (ble_kernels.hip asserts the layout and copies the prefix.)
struct StateDev {
  float *x, *y, *pressure, *ambient_temperature, *internal_temperature, *envelope_volume, *superpressure, *mols_air, *battery_charge;
  float *acs_power, *acs_mass_flow, *solar_charging, *power_load;
  const float *center_lat_deg, *center_lng_deg, *upwelling_infrared, *alpha;
  const int64_t* start_unix;
  int32_t *time_elapsed_s, *sunrise_h_rel, *sunset_rel;
  uint8_t *status, *last_command, *alt_fsm, *env_fsm, *power_paused;
  double* episode_cache;
};
struct EnvRegs {
  float x, y, p, t_amb, t_int, vol, sp, n_air, batt;
  float acs_power, mdot, charge, load;
  int32_t t_elapsed, sunrise_h, sunset;
  uint8_t status, alt_fsm, env_fsm, paused;
};
struct EnvConst {
  float lat0_deg, lng0_deg, ir, alpha;
  int64_t start_unix;
};
// What depends on the per-episode constants only (two pows, a sincos, two square roots): the atmosphere's transition
// pressures, the station latitude's sine and cosine, the earth-IR heat per unit area.  Constant over an episode, so it is
// kept in HBM next to the state (`ble_state_f32.episode_cache`, [kEpisodeCacheRows][n] doubles, coalesced): ble_reset_f32
// fills it, and a transition whose cached entry was computed for other constants -- a caller edited alpha / the centre /
// the IR by hand, or never reset on the device -- recomputes and stores it itself.  The entry is keyed by the bit
// patterns of (alpha, centre latitude, upwelling IR); an all-zero (freshly allocated) row never matches.
struct EnvHoisted {
  AtmBase atm;
  double sin_lat0, cos_lat0;
  double q_earth;          // earth-IR heat per unit area (thermal.py:209-213): a function of the episode's IR alone
  uint32_t flags;          // its total_absorptivity range check
};
constexpr int kEpisodeCacheRows = 7;    // p1, p2, sin lat0, cos lat0, q_earth, key (alpha | lat0), key (IR | valid | flags)
BLE_FN EnvHoisted hoist_constants(const EnvConst& c) {
  EnvHoisted h;
  h.atm = atm_base((double)c.alpha);
  sincos_f64((double)c.lat0_deg * (kPiD / 180.0), &h.sin_lat0, &h.cos_lat0);
  h.flags = 0;
  h.q_earth = earth_heat_per_area_f64((double)c.ir, &h.flags);
  return h;
}
BLE_FN uint32_t float_bits(float v) { uint32_t u; __builtin_memcpy(&u, &v, 4); return u; }
BLE_FN void episode_cache_keys(const EnvConst& c, uint64_t* k1, uint64_t* k2) {
  *k1 = ((uint64_t)float_bits(c.alpha) << 32) | (uint64_t)float_bits(c.lat0_deg);
  *k2 = ((uint64_t)float_bits(c.ir) << 32) | 0x80000000ull;
}
BLE_FN void episode_cache_store(double* cache, int64_t n, int64_t i, const EnvConst& c, const EnvHoisted& h) {
  uint64_t k1, k2;
  episode_cache_keys(c, &k1, &k2);
  k2 |= (uint64_t)(h.flags & 0x7fffffffu);
  double d1, d2;
  __builtin_memcpy(&d1, &k1, 8); __builtin_memcpy(&d2, &k2, 8);
  cache[i] = h.atm.p1; cache[n + i] = h.atm.p2; cache[2 * n + i] = h.sin_lat0; cache[3 * n + i] = h.cos_lat0;
  cache[4 * n + i] = h.q_earth; cache[5 * n + i] = d1; cache[6 * n + i] = d2;
}
// the seven cached doubles of a lane, requested together with the state
struct EpisodeCacheRow { double v[kEpisodeCacheRows]; };
BLE_FN EpisodeCacheRow episode_cache_load(const double* cache, int64_t n, int64_t i) {
  EpisodeCacheRow r;
#pragma unroll
  for (int k = 0; k < kEpisodeCacheRows; ++k) r.v[k] = cache[k * n + i];
  return r;
}
BLE_FN bool episode_cache_hit(const EpisodeCacheRow& r, const EnvConst& c) {
  uint64_t k1, k2, g1, g2;
  episode_cache_keys(c, &k1, &k2);
  __builtin_memcpy(&g1, &r.v[5], 8); __builtin_memcpy(&g2, &r.v[6], 8);
  return g1 == k1 && (g2 & 0xffffffff80000000ull) == k2;
}
// the cheap, alpha-only part of AtmBase (lapse rates, layer-top temperatures) around the two cached transition pressures
BLE_FN EnvHoisted hoisted_from_cache(const EpisodeCacheRow& r, const EnvConst& c) {
  EnvHoisted h;
  atm_base_linear((double)c.alpha, &h.atm);
  h.atm.p1 = r.v[0]; h.atm.p2 = r.v[1];
  h.sin_lat0 = r.v[2]; h.cos_lat0 = r.v[3]; h.q_earth = r.v[4];
  uint64_t g2;
  __builtin_memcpy(&g2, &r.v[6], 8);
  h.flags = (uint32_t)(g2 & 0x7fffffffull);
  return h;
}

// ---------------------------------------------------------------------------------------------------------------
// The right-hand sides of one 10 s stride (balloon.py:356-549), one function per group of state variables.  Every one of
// them reads the OLD state only (balloon.py:322-325 commits after all of them), so they are independent inside a stride:
// agent_step below evaluates them one after the other on one lane; ble_step_split.h evaluates them on four wavefronts.
// Both call these very functions -- the two kernels agree bit for bit (tests/test_gpu_parity.py).

// V^(-1/3): 1/drag = 4 V^(-2/3) of the vertical dynamics and the radius of the thermal model; fp32 seed + one Newton step
// y <- y (4 - V y^3) / 3
BLE_FN double inv_cbrt_volume(double vol) {
  double yc = (double)f_exp2((-1.0f / 3.0f) * f_log2((float)vol));              // fp32 seed
  return yc * d_fma(-vol * yc, yc * yc, 4.0) * (1.0 / 3.0);
}
// the layer of the atmosphere window that holds p, carried from stride to stride
struct LayerCursor {
  int lay;                 // -1, 0, +1 relative to the window's centre layer
  double lapse_cur;        // its lapse rate
  double kl_cur;           // (-R_d / g) x the lapse rate: the exponent of T(p) inside the layer
  double cur_hi, cur_lo;   // the transition pressures that bound it (+-inf beyond the window)
};
// step 2: buoyancy -> dh/dt -> dp (balloon.py:412-445), fp64 throughout: near float equilibrium
// d(dp)/d(rho V - m) ~ 1/sqrt|rho V - m| is unbounded, so an fp32-sized error in the increment itself is amplified past the
// parity bar within a few substeps.  rho V - m = (p V M/R - m T) / T ; the common 1/T cancels in (rho V - m) / rho
BLE_FN double stride_pressure(const AtmWindow& win, const LayerCursor& lc, double p, double rp, double vol, double n_air,
                              double t_amb, double t_at_p, double yc, const StrideK& K, double drag_arg = VehicleDefault::drag_arg) {
  const double mass = d_fma(kAirMolarMassD, n_air, K.dry_mass);
  const double num = d_fma(p * vol, K.m_over_r, -mass * t_amb);
  const double dir = num >= 0.0 ? 1.0 : -1.0;
  // dh/dt = dir sqrt(|2 (rho V - m) g / (rho drag)|) = dir sqrt(2 g |num| (R/M) (1/p) V^(-2/3) / cod); drag_arg = 2 g (R/M) / cod (cod = 0.25: 8 g R/M)
  const double arg = drag_arg * __builtin_fabs(num) * rp * (yc * yc);
  const double dh_dt = d_sqrt_rs(d_max(arg, 1e-30));                      // arg == 0 (exact equilibrium): 1e-15 m/s, p unchanged
  const double inv_dh = atm_inv_delta_height_f64(win, lc.lay, lc.lapse_cur, lc.kl_cur, lc.cur_hi, lc.cur_lo, p, rp, dir, t_at_p, K);
  return d_fma(inv_dh * dh_dt, K.ten, p);                                 // dir * dir == 1
}
// step 3: internal temperature (balloon.py:451-467)
BLE_FN double stride_internal_temperature(double vol, double yc, double t_int, double t_amb, double p, float flux, float att,
                                          double q_earth, const StrideK& K, double thermal_scale = VehicleDefault::thermal_scale) {
  return t_int + thermal_increment_f64(vol, yc, t_int, t_amb, p, (double)((flux * att) * (0.25f * kSolarAbsorptivityTotal)), q_earth, K, thermal_scale);
}
// step 5: ACS (balloon.py:487-519); both branches evaluated, selected per lane.  fp64: the mass flow changes rho V - m by
// ~1e-2 kg per stride, an fp32 rounding of it (~1e-9 kg) is amplified like the thermal increment's.
BLE_FN void stride_acs(const double* acs_poly, int eff, double sp, double p, double rp, double t_int, float* acs_w, double* mdot_d,
                       const StrideK& K, double valve_k = VehicleDefault::valve_k) {
  // -0.62 A sqrt(2 sp rho_gas), rho_gas = (sp + p) M / (R T_int):  sqrt(a / T) = a rsqrt(a T)
  const double a2 = d_max((2.0 * (kAirMolarMassD / kGasConstantD)) * (sp * (sp + p)), 1e-30);
  const double mdot_up = (valve_k * a2) * d_rsqrt(a2 * t_int);                        // valve_k = -0.62 A;  sp == 0: -1e-17 kg/s
  const double prm1 = d_max(sp * rp, 0.0);                // pressure_ratio - 1 = max(sp, 0) / p (balloon.py:247-250; rp > 0)
  double w_down, mdot_down;
  acs_down_poly(acs_poly, prm1, &w_down, &mdot_down, K);
  *acs_w = eff == kDown ? (float)w_down : 0.0f;
  *mdot_d = eff == kUp ? mdot_up : (eff == kDown ? mdot_down : 0.0);
}
BLE_FN double stride_mols_air(double n_air, double mdot_d) {
  return d_max(d_fma(mdot_d, 10.0 / kAirMolarMassD, n_air), 0.0);
}
// step 6: power (balloon.py:524-542)
BLE_FN void stride_power_from_factor(bool is_day, float panel_factor, float att, float acs_w, float* charge, float* load, float* batt,
                                     float day_load = VehicleDefault::day_load, float night_load = VehicleDefault::night_load,
                                     float capacity = VehicleDefault::capacity) {
  *charge = is_day ? solar_power_from_factor(panel_factor, att) : 0.0f;
  *load = (is_day ? day_load : night_load) + acs_w;
  *batt = f_clamp(f_fma(*charge - *load, kStride / 3600.0f, *batt), 0.0f, capacity);
}
BLE_FN void stride_power(const SunState& sun, float att, float acs_w, float* charge, float* load, float* batt,
                         float day_load = VehicleDefault::day_load, float night_load = VehicleDefault::night_load,
                         float capacity = VehicleDefault::capacity) {
  stride_power_from_factor(sun.day, solar_panel_factor(sun), att, acs_w, charge, load, batt, day_load, night_load, capacity);
}
// T(p_new) for the next stride: advance inside the layer; if a transition was crossed (cold branch) re-anchor at it first --
// no transcendental either way (see AtmWindow).  Updates the cursor.
BLE_FN double stride_ambient_advance(const AtmWindow& win, LayerCursor& lc, double p, double rp, double t_at_p, double p_new,
                                     const StrideK& K) {
  const double kInf = (double)__builtin_huge_valf();
  double anchor_p = p, anchor_rp = rp, anchor_t = t_at_p;
  const bool crossed = p_new > lc.cur_hi || !(p_new > lc.cur_lo);
  if (__builtin_expect(wave_any(crossed), 0)) if (crossed) {
    BLE_STEP_EVENT(1);
    const int lay_new = atm_window_layer(win, p_new);
    const bool low_pair = (lc.lay + lay_new) < 0;            // crossing pb (else pt)
    anchor_p = low_pair ? win.pb : win.pt;
    anchor_rp = low_pair ? win.r_pb : win.r_pt;
    anchor_t = low_pair ? win.tb : win.tt;
    lc.lay = lay_new;
    lc.lapse_cur = pick3(lay_new, win.lapse_m1, win.lapse_0, win.lapse_p1);
    lc.kl_cur = (-kAirSpecificGasD / 9.80665) * lc.lapse_cur;
    lc.cur_hi = lay_new < 0 ? kInf : (lay_new == 0 ? win.pb : win.pt);
    lc.cur_lo = lay_new < 0 ? win.pb : (lay_new == 0 ? win.pt : -kInf);
  }
  return atm_temperature_advance(anchor_t, anchor_p, anchor_rp, p_new, lc.kl_cur, K);
}

// Solar geometry of one agent step: 1 - sin(el_uncorrected) at stride indices 0, n/2, n in fp64 (sun_one_minus_sin_f64), then a
// quadratic in the stride index evaluated in fp32 inside the loop.  The time-only half (hour-angle base and declination at the
// three nodes) and the site half are separate so that ble_step_split.h can evaluate them on different wavefronts.
struct SolarNodes {
  double sb0, cb0, sb1, cb1, sb2, cb2;     // (sin, cos) of the hour-angle base b = 360 frac_day + eot/4 + lng0 at the three nodes
  double sd0, cd0, hsd, hcd;               // declination (sin, cos) at node 0 and its change per half step
};
BLE_FN SolarNodes solar_nodes_time(const Ephemeris& e0, int64_t t0, float lng0_deg, float step_s) {
  SolarNodes n;
  // hour-angle base B = 360 frac_day + eot/4 + lng0  [deg]  (solar.py:113-116)
  double sod;
  if (__builtin_expect(t0 >= 0 && t0 < 4294967296LL, 1)) sod = (double)((uint32_t)t0 % 86400u);
  else { int64_t m = t0 % 86400; sod = (double)(m < 0 ? m + 86400 : m); }
  const double b0 = sod * (1.0 / 240.0) + 0.25 * e0.eot_min + (double)lng0_deg;
  const double half_db = 0.5 * ((double)step_s * (1.0 / 240.0) + 0.25 * (double)(e0.eot_min_rate * step_s));  // deg
  double sb0, cb0;
  sincos_f64(b0 * (kPiD / 180.0), &sb0, &cb0);
  // rotate by the half-step angle (0.375 deg): Taylor
  const double hr = half_db * (kPiD / 180.0), h2 = hr * hr;
  const double sh = hr * d_fma(h2, d_fma(h2, d_fma(h2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
  const double ch = d_fma(h2, d_fma(h2, d_fma(h2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
  const double sb1 = sb0 * ch + cb0 * sh, cb1 = cb0 * ch - sb0 * sh;
  const double sb2 = sb1 * ch + cb1 * sh, cb2 = cb1 * ch - sb1 * sh;
  n.sb0 = sb0; n.cb0 = cb0; n.sb1 = sb1; n.cb1 = cb1; n.sb2 = sb2; n.cb2 = cb2;
  // (sin, cos) of the declination as an exactly normalised fp64 pair: with the fp32 pair
  // (|sd^2 + cd^2 - 1| ~ 6e-8) the error of 1 - sin(el) near the zenith (el > 89.8 deg, 1 - sin(el)
  // < 1e-6) was ~10 % and cos(el) -- hence the panel power -- was off by 2e-5
  const double sd0 = (double)e0.sin_decl, cd0 = d_sqrt_fast(d_fma(-sd0, sd0, 1.0));
  const double hsd = 0.5 * (double)(e0.sin_decl_rate * step_s), hcd = -(sd0 * d_rcp(cd0)) * hsd;
  n.sd0 = sd0; n.cd0 = cd0; n.hsd = hsd; n.hcd = hcd;
  return n;
}
// 1 - sin(el_uncorrected) at node j (0: start of the step, 1: middle, 2: end); the balloon moves with the step's constant wind
template <int j>
BLE_FN double solar_node(const SolarNodes& n, double sl0, double cl0, float x_m, float y_m, float u, float v, int substeps) {
  const double x0 = (double)x_m, y0 = (double)y_m;
  const double dx = (double)u * (5.0 * (double)substeps), dy = (double)v * (5.0 * (double)substeps);  // half step
  if (j == 0) return sun_one_minus_sin_f64(sl0, cl0, x0, y0, n.sb0, n.cb0, n.sd0, n.cd0);
  if (j == 1) return sun_one_minus_sin_f64(sl0, cl0, x0 + dx, y0 + dy, n.sb1, n.cb1, n.sd0 + n.hsd, n.cd0 + n.hcd);
  return sun_one_minus_sin_f64(sl0, cl0, x0 + 2.0 * dx, y0 + 2.0 * dy, n.sb2, n.cb2, n.sd0 + 2.0 * n.hsd, n.cd0 + 2.0 * n.hcd);
}
// the quadratic through the three nodes, in the stride index, and the first node's distances to the solar thresholds
struct SunQuadratic { float c0, c1, c2; SunThresholds thr; };
BLE_FN SunQuadratic solar_node_coefs(double f0, double f1, double f2, int substeps) {
  // divided differences over the half step m = substeps / 2 strides: / (2 m) and / (2 m^2) through one reciprocal (the float32
  // coefficients absorb its 2e-15; two true fp64 divisions were ~60 instructions per agent step)
  const double inv_2m = d_rcp((double)substeps);
  SunQuadratic sq;
  sq.c0 = (float)f0;
  sq.c1 = (float)((-f2 + 4.0 * f1 - 3.0 * f0) * inv_2m);
  sq.c2 = (float)((f2 - 2.0 * f1 + f0) * (2.0 * (inv_2m * inv_2m)));
  sq.thr = sun_thresholds(f0, substeps);
  return sq;
}
BLE_FN SunQuadratic solar_nodes_site(const SolarNodes& n, double sl0, double cl0, float x_m, float y_m, float u, float v, int substeps) {
  const double f0 = solar_node<0>(n, sl0, cl0, x_m, y_m, u, v, substeps);
  const double f1 = solar_node<1>(n, sl0, cl0, x_m, y_m, u, v, substeps);
  const double f2 = solar_node<2>(n, sl0, cl0, x_m, y_m, u, v, substeps);
  return solar_node_coefs(f0, f1, f2, substeps);
}
// Sun at stride kk of a step: the quadratic through the three fp64 nodes, fp32; the reference's own fp64 chain on the (rare)
// strides where a solar threshold is within the fp32 floor.  (x_start, y_start, t_start: position and time at the START of the step.)
BLE_FN SunState sun_at_stride(int kk, const SunQuadratic& sq, const EnvConst& c, float u, float v, float x_start,
                              float y_start, int32_t t_start) {
  const float fkk = (float)kk;
  bool near;
  const float q = fkk * f_fma(fkk, sq.c2, sq.c1);        // the increment over the first node: 0 at stride 0
  SunState r = sun_fast(sq.c0 + q, q, sq.thr, &near);
  if (__builtin_expect(wave_any(near), 0)) if (near) {
    BLE_STEP_EVENT(0);
    kk = i_opaque(kk);                   // (otherwise the common path carries 10 kk and (double)kk as induction variables for this block)
    const double dk = 10.0 * (double)kk;
    r = sun_exact((double)c.lat0_deg, (double)c.lng0_deg, d_fma(dk, (double)u, (double)x_start), d_fma(dk, (double)v, (double)y_start),
                  c.start_unix + (int64_t)(t_start + 10 * kk));
  }
  return r;
}
// BalloonState.excess_energy's battery test (balloon.py:231-238): battery_charge / battery_capacity > 0.99 in the reference's
// float64, on the float32 charge the ABI carries.  A correctly rounded division is monotone in its numerator, so the predicate
// is a threshold on the float32 itself: 3027.9746 (0x1.7a7f3p+11) is the smallest float32 b with (double)b / 3058.56 > 0.99
// (its predecessor gives 0.98999999; tests/test_kernel_numerics_host.py) -- the same decision without the fp64 division.
// Another capacity (a run-time vehicle): the reference's division.
BLE_FN bool battery_above_99_percent(float batt, double capacity_wh = VehicleDefault::capacity_d) {
  return capacity_wh == 3058.56 ? batt >= 3027.9746f : (double)batt / capacity_wh > 0.99;
}
// perciatelli_reward_function (env/balloon_env.py:44-102) on the post-step state; `sun` = the sun at the end of the step
// (only read when the raw action was DOWN: last_command is the RAW action, balloon.py:286)
template <typename SunFn>
BLE_FN float step_reward(int action, float x, float y, float p, float batt, float acs_power, SunFn sun_end,
                         float day_load = VehicleDefault::day_load, double capacity_wh = VehicleDefault::capacity_d) {
  float r = reward_distance(x, y);
  if (action == kDown) {
    const SunState sun = sun_end();
    const float pw = solar_power(sun, solar_attenuation(sun.sin_el, p, sun.day));
    const bool excess = (pw > day_load) && battery_above_99_percent(batt, capacity_wh);   // balloon.py:231-238
    if (!excess) {
      const float scale = f_clamp((acs_power - 100.0f) * (1.0f / 200.0f), 0.0f, 1.0f);
      r *= f_fma(-0.3f, scale, 0.95f);
    }
  }
  return r;
}

constexpr int kTermSaveRows = 14, kTermSaveStride = 64;   // agent_step's parking area: 13 state / output floats + (status | strides << 8), one column per lane
// Returns the effective action (after the safety layers).  `reward` gets the post-step
// reward; `s` is advanced in place.  Precondition: s.status == kOk.
// The wind is handed over as the 16 gathered grid corners + weights (+ additive noise): the
// blend happens after the per-step constants so that the gather's latency is covered.
// `veh`: the flight vehicle (VehicleDefault: compile-time constants -- the code and the bits of every round before ABI 5 --, or VehicleRt).
template <class V = VehicleDefault>
BLE_FN int agent_step(EnvRegs& s, const EnvConst& c, const EnvHoisted& hc, int action, const WindCorners& corners, const WindQuery& wq,
                      float noise_u, float noise_v, int substeps, const double* acs_poly, const StrideK& K, float* term_save,
                      float* reward, uint32_t* flags, const V& veh = V()) {
  BLE_STEP_TICK(0);
  // ---- atmosphere at the pre-step pressure, fp64 (altitude layer + start of T(p) chain)
  const float p0_in = s.p;
  double p = (double)s.p;
  const AtmWindow win = atm_window_from(hc.atm, (double)c.alpha, p, flags);
  double altitude, t_at_p;
  atm_at_pressure_f64(win, (double)c.alpha, p, &altitude, &t_at_p);
  LayerCursor lc;                                // p is in the window's centre layer by construction
  lc.lay = 0; lc.lapse_cur = win.lapse_0; lc.kl_cur = (-kAirSpecificGasD / 9.80665) * win.lapse_0; lc.cur_hi = win.pb; lc.cur_lo = win.pt;

  // ---- safety layers, once per agent step, on the pre-step state (balloon.py:304-313)
  int eff = action;
  if (veh.power_layer)                           // BalloonState.power_safety_layer_enabled (balloon.py:305): a disabled layer's clocks do not move
    eff = power_safety(action, s.t_elapsed, s.batt, &s.sunrise_h, &s.sunset, &s.paused, veh.night_load_d, veh.capacity_d);
  eff = envelope_safety(eff, s.sp, &s.env_fsm, veh.max_sp);
  eff = altitude_safety(eff, altitude, &s.alt_fsm);
  BLE_STEP_TICK(1);

  // ---- per-step constants
  const int64_t t0 = c.start_unix + (int64_t)s.t_elapsed;
  const Ephemeris e0 = ephemeris(t0);
  const float step_s = (float)(10 * substeps);
  const float fl0 = e0.flux, dfl = e0.flux_rate * 10.0f;
  BLE_STEP_TICK(2);
  float u, v;
  wind_blend_corners(corners, wq, &u, &v);         // wind at the PRE-step position/time
  u += noise_u; v += noise_v;                      // WindField.get_ground_truth = forecast + noise
  BLE_STEP_TICK(3);
  // Solar geometry: 1 - sin(el_uncorrected) at substep indices 0, n/2, n in fp64, then a
  // quadratic in k evaluated in fp32 inside the loop (see sun_one_minus_sin_f64).
  const SolarNodes nodes = solar_nodes_time(e0, t0, c.lng0_deg, step_s);
  const SunQuadratic sq = solar_nodes_site(nodes, hc.sin_lat0, hc.cos_lat0, s.x, s.y, u, v, substeps);
  BLE_STEP_TICK(4);
  // (position and time at the START of the step, by value: the reward below calls this after s has been advanced)
  const float x_start = s.x, y_start = s.y;
  const int32_t t_start = s.t_elapsed;
  auto sun_at = [&](int kk) -> SunState {
    return sun_at_stride(kk, sq, c, u, v, x_start, y_start, t_start);
  };
  const double q_earth = hc.q_earth;
  *flags |= hc.flags;
  // total_absorptivity's range check (thermal.py:142-145) on the balloon's own temperature: the
  // factor leaves [0, 1] only for T_int < 12.3 K; T_int moves < 1 K per step, so checking the
  // step's first and last value is checking every stride
  *flags |= (s.t_int < 12.3f) ? kFlagAbsorptivity : 0u;

  // ---- fp64 carried chain
  double t_amb = (double)s.t_amb, t_int = (double)s.t_int, n_air = (double)s.n_air, vol = (double)s.vol,
         sp = (double)s.sp;
  float x = s.x, y = s.y, batt = s.batt;
  float acs_w = s.acs_power, mdot = s.mdot, charge = s.charge, load = s.load;
  int status = kOk;

  // The stride loop is wave-uniform: every lane runs all `substeps` strides and the loop index is a scalar.  A lane whose episode ends
  // inside the step (balloon.py:327-328 breaks there; about one stride in 300 of a wave) parks the state it ended with in LDS
  // (term_save: kTermSaveRows floats per lane, row stride kTermSaveStride) on a rare path, keeps computing on a state nobody reads, and
  // takes the parked values back after the loop -- a divergent `break` cost the common stride 20 instructions of exec-mask bookkeeping.
  float* const term_word = term_save + (kTermSaveRows - 1) * kTermSaveStride;   // status | strides run << 8; 0 while the episode runs
  *term_word = 0.0f;
  auto stride = [&](const int k) __attribute__((always_inline)) {
    const float pf = (float)p;
    const double rp = d_rcp(p);
    // ---- sun position at (x, y, date_time) of the OLD state (balloon.py:451-452)
    const float fk = (float)k;
    const SunState sun = sun_at(k);
    const float flux = f_fma(fk, dfl, fl0);

    // ---- step 2: buoyancy -> dh/dt -> dp (balloon.py:412-445)
    const double yc = inv_cbrt_volume(vol);
    const double p_new = stride_pressure(win, lc, p, rp, vol, n_air, t_amb, t_at_p, yc, K, veh.drag_arg);

    // ---- step 3: temperatures (balloon.py:451-467)
    const float att = solar_attenuation(sun.sin_el, pf, sun.day);
    const double t_int_new = stride_internal_temperature(vol, yc, t_int, t_amb, p, flux, att, q_earth, K, veh.thermal_scale);

    // ---- step 4: superpressure and volume (balloon.py:470-482)
    double vol_new, sp_new;
    superpressure_volume_f64(n_air, t_int, p, rp, &vol_new, &sp_new, K, veh.dvdp, veh.four_dvdp, veh.inv_dvdp);
    // balloon.py:479-482: burst above envelope_max_superpressure (2 380 Pa), zero pressure at <= 0 (the status code is formed after the loop)
    bool terminal = !(sp_new <= veh.max_sp) || sp_new <= 0.0;

    // ---- step 5: ACS (balloon.py:487-519)
    double mdot_d;
    stride_acs(acs_poly, eff, sp, p, rp, t_int, &acs_w, &mdot_d, K, veh.valve_k);
    mdot = (float)mdot_d;
    const double n_air_new = stride_mols_air(n_air, mdot_d);

    // ---- step 6: power (balloon.py:524-542)
    stride_power(sun, att, acs_w, &charge, &load, &batt, veh.day_load, veh.night_load, veh.capacity);
    terminal = terminal || batt <= 0.0f;          // balloon.py:541-542

    // ---- commit (balloon.py:322-325): every RHS above used the old state
    x = f_fma(u, kStride, x);            // step 1 (balloon.py:394-395)
    y = f_fma(v, kStride, y);
    t_amb = t_at_p;                      // ambient_temperature' = T(p_old)  (balloon.py:457)
    t_at_p = stride_ambient_advance(win, lc, p, rp, t_at_p, p_new, K);
    p = p_new; t_int = t_int_new; vol = vol_new; sp = sp_new; n_air = n_air_new;
    if (__builtin_expect(wave_any(terminal), 0)) if (terminal && __builtin_bit_cast(int, *term_word) == 0) {          // balloon.py:327-328
      // status of the stride that ended the step (later checks override earlier ones, like the reference's assignments).  The burst
      // test is `!(sp <= max)`: a non-finite superpressure ends the episode too (kBurst + kFlagNonFinite), so that the lane is
      // frozen for the remaining steps of a fused launch instead of stepping on NaN state.
      int st = kOk;
      if (!(sp <= veh.max_sp)) st = kBurst;
      if (sp <= 0.0) st = kZeroPressure;
      if (batt <= 0.0f) st = kOutOfPower;
      const float parked[kTermSaveRows - 1] = {x, y, (float)p, (float)t_amb, (float)t_int, (float)vol, (float)sp, (float)n_air, batt,
                                               acs_w, mdot, charge, load};
#pragma unroll
      for (int j = 0; j < kTermSaveRows - 1; ++j) term_save[j * kTermSaveStride] = parked[j];
      const int strides = i_opaque(k + 1);          // (otherwise the common path carries (k + 1) << 8 as an induction variable for this block)
      *term_word = __builtin_bit_cast(float, st | (strides << 8));
    }
  };
  // two strides per iteration: the loop-carried values alternate between two sets of registers instead of being copied
  int ks = 0;
#pragma unroll 1
  for (; ks + 1 < substeps; ks += 2) { stride(ks); stride(ks + 1); }
  if (ks < substeps) stride(ks);
  BLE_STEP_TICK(5);
  s.x = x; s.y = y; s.p = (float)p; s.t_amb = (float)t_amb; s.t_int = (float)t_int; s.vol = (float)vol;
  s.sp = (float)sp; s.n_air = (float)n_air; s.batt = batt;
  s.acs_power = acs_w; s.mdot = mdot; s.charge = charge; s.load = load;
  int k = substeps;                      // strides this lane ran (>= 1: the host entry point checks substeps >= 1)
  const int word = __builtin_bit_cast(int, *term_word);
  const bool done = word != 0;
  if (__builtin_expect(wave_any(done), 0)) if (done) {
    s.x = term_save[0]; s.y = term_save[kTermSaveStride]; s.p = term_save[2 * kTermSaveStride]; s.t_amb = term_save[3 * kTermSaveStride];
    s.t_int = term_save[4 * kTermSaveStride]; s.vol = term_save[5 * kTermSaveStride]; s.sp = term_save[6 * kTermSaveStride];
    s.n_air = term_save[7 * kTermSaveStride]; s.batt = term_save[8 * kTermSaveStride]; s.acs_power = term_save[9 * kTermSaveStride];
    s.mdot = term_save[10 * kTermSaveStride]; s.charge = term_save[11 * kTermSaveStride]; s.load = term_save[12 * kTermSaveStride];
    status = word & 0xff; k = word >> 8;
  }
  s.t_elapsed += 10 * k;
  s.status = (uint8_t)status;
  *flags |= (s.t_int < 12.3f) ? kFlagAbsorptivity : 0u;

  // solar_atmospheric_attenuation's range check (solar.py:194-197); p moves < 3 kPa per step
  *flags |= (s.p > 101325.0f || s.p < 0.0f || p0_in > 101325.0f || p0_in < 0.0f) ? kFlagSolarRange : 0u;

  // ---- reward (env/balloon_env.py:44-102), on the post-step state
  *reward = step_reward(action, s.x, s.y, s.p, s.batt, s.acs_power, [&]() { return sun_at(k); }, veh.day_load, veh.capacity_d);
  BLE_STEP_TICK(6);
  return eff;
}

}  // namespace ble
