// Host build of the kernel's per-lane functions (ble_physics.h / ble_step_core.h) for
// numerics triage on machines without a GPU.  TEST TOOLING: never loaded by the package;
// the product path is the HIP library only.  libm stands in for the v_exp/v_log/v_rcp/
// v_sqrt hardware approximations, so this shows the algorithmic fp32 error, not the
// last-ulp behaviour of the device.
#include "../../balloon_learning_environment_amd/csrc/ble_step_core.h"
#include "../../balloon_learning_environment_amd/csrc/ble_noise.h"
#include "../../balloon_learning_environment_amd/csrc/ble_decode.h"
#include "../../include/ble_abi.h"

using namespace ble;

extern "C" int emul_step_f32(const ble_state_f32* st, const uint8_t* action, const float* wind_grid,
                             const float* wind_uv, float* reward, uint8_t* terminal, uint8_t* effective_action,
                             uint32_t* err_flags, int64_t n, int substeps) {
  uint32_t flags_all = 0;
  const double* acs_poly = kAcsPoly.c;       // the compile-time table (the device copies it into LDS)
  for (int64_t i = 0; i < n; ++i) {
    if (st->status[i] != kOk) { reward[i] = 0.0f; terminal[i] = 1; if (effective_action) effective_action[i] = action[i]; continue; }
    EnvRegs s;
    s.x = st->x[i]; s.y = st->y[i]; s.p = st->pressure[i]; s.t_amb = st->ambient_temperature[i];
    s.t_int = st->internal_temperature[i]; s.vol = st->envelope_volume[i]; s.sp = st->superpressure[i];
    s.n_air = st->mols_air[i]; s.batt = st->battery_charge[i]; s.acs_power = st->acs_power[i];
    s.mdot = st->acs_mass_flow[i]; s.charge = st->solar_charging[i]; s.load = st->power_load[i];
    s.t_elapsed = st->time_elapsed_s[i]; s.sunrise_h = st->sunrise_h_rel[i]; s.sunset = st->sunset_rel[i];
    s.status = st->status[i]; s.alt_fsm = st->alt_fsm[i]; s.env_fsm = st->env_fsm[i]; s.paused = st->power_paused[i];
    EnvConst c{st->center_lat_deg[i], st->center_lng_deg[i], st->upwelling_infrared[i], st->alpha[i], st->start_unix[i]};
    WindQuery wq = wind_query(s.x, s.y, s.p, s.t_elapsed);
    WindCorners wc;
    float nu = 0.0f, nv = 0.0f;
    if (wind_uv) {  // fixed wind: zero grid + the wind as the additive term
      for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) wc.c[a][b] = 0.0f;
      nu = wind_uv[2 * i]; nv = wind_uv[2 * i + 1];
    } else {
      wind_gather(wind_grid, wq, &wc);
    }
    uint32_t flags = 0; float r;
    float term_save[kTermSaveRows * kTermSaveStride];       // (the kernel's LDS parking area of a lane whose episode ends inside the step)
    int eff = agent_step(s, c, hoist_constants(c), action[i], wc, wq, nu, nv, substeps, acs_poly, stride_k_literal(), term_save, &r, &flags);
    flags_all |= flags;
    st->x[i] = s.x; st->y[i] = s.y; st->pressure[i] = s.p; st->ambient_temperature[i] = s.t_amb;
    st->internal_temperature[i] = s.t_int; st->envelope_volume[i] = s.vol; st->superpressure[i] = s.sp;
    st->mols_air[i] = s.n_air; st->battery_charge[i] = s.batt; st->acs_power[i] = s.acs_power;
    st->acs_mass_flow[i] = s.mdot; st->solar_charging[i] = s.charge; st->power_load[i] = s.load;
    st->time_elapsed_s[i] = s.t_elapsed; st->sunrise_h_rel[i] = s.sunrise_h; st->sunset_rel[i] = s.sunset;
    st->status[i] = s.status; st->alt_fsm[i] = s.alt_fsm; st->env_fsm[i] = s.env_fsm; st->power_paused[i] = s.paused;
    st->last_command[i] = action[i];
    reward[i] = r; terminal[i] = s.status != kOk;
    if (effective_action) effective_action[i] = (uint8_t)eff;
  }
  if (err_flags) *err_flags |= flags_all;
  return 0;
}

extern "C" void emul_solar(int64_t n, const float* lat0, const float* lng0, const float* x, const float* y,
                           const int64_t* t, float* el_deg, float* flux) {
  for (int64_t i = 0; i < n; ++i) {
    Ephemeris e = ephemeris(t[i]);
    int64_t sod = t[i] % 86400; if (sod < 0) sod += 86400;
    double b = (double)sod / 240.0 + 0.25 * e.eot_min + (double)lng0[i];
    double sl, cl; sincos_f64((double)lat0[i] * (kPiD / 180.0), &sl, &cl);
    double sb, cb; sincos_f64(b * (kPiD / 180.0), &sb, &cb);
    double oms = sun_one_minus_sin_f64(sl, cl, x[i], y[i], sb, cb, (double)e.sin_decl, (double)e.cos_decl);
    SunSC sun = sun_refract(sun_from_one_minus_sin((float)oms));
    el_deg[i] = atan2f(sun.sin_el, sun.cos_el) * kRadToDeg;
    flux[i] = e.flux;
  }
}

extern "C" void emul_solar_power(int64_t n, const float* el_deg, const float* p, float* att, float* power) {
  for (int64_t i = 0; i < n; ++i) {
    double s, c; sincos_f64((double)el_deg[i] * (kPiD / 180.0), &s, &c);
    uint32_t fl = 0;
    const double el = (double)el_deg[i];
    SunState sun;
    sun.sin_el = (float)s; sun.cos_el = (float)c;
    sun.day = !(el < -4.242); sun.sh33 = el >= 37.738149050524044; sun.sh27 = el >= 34.39486500086289;
    att[i] = solar_attenuation(sun.sin_el, p[i], sun.day);
    power[i] = solar_power(sun, att[i]);
  }
}

extern "C" void emul_ephemeris(int64_t t, double* out4) {
  Ephemeris e = ephemeris(t);
  out4[0] = e.eot_min; out4[1] = e.sin_decl; out4[2] = e.cos_decl; out4[3] = e.flux;
  out4[4] = e.eot_min_rate; out4[5] = e.sin_decl_rate; out4[6] = e.cos_decl_rate; out4[7] = e.flux_rate;
}

// ---- reset path (ble_reset.h) on the host ----
#include "../../balloon_learning_environment_amd/csrc/ble_reset.h"
extern "C" void emul_reset_derive(int64_t n, const float* alpha, const float* x, const float* y, const float* p,
                                  const float* lat0, const float* lng0, const float* ir, const int64_t* start,
                                  double* t_amb, double* t_int, double* mols_air, double* volume, double* sp,
                                  int64_t* sunrise, int64_t* sunset, double* el_out) {
  for (int64_t i = 0; i < n; ++i) {
    SunSite site;
    latlng_f64((double)lat0[i], (double)lng0[i], (double)x[i], (double)y[i], &site.sin_lat, &site.cos_lat, &site.lng_deg);
    double flux;
    const double el = solar_elevation_f64(site.sin_lat, site.cos_lat, site.lng_deg, start[i], &flux);
    uint32_t flags = 0;
    const StableParams s = stable_params((double)alpha[i], (double)p[i], el, flux, (double)ir[i], &flags);
    t_amb[i] = s.t_amb; t_int[i] = s.t_int; mols_air[i] = s.mols_air; volume[i] = s.volume; sp[i] = s.sp;
    next_sunrise_sunset(site, start[i], &sunrise[i], &sunset[i]);
    el_out[i] = el;
  }
}
extern "C" void emul_philox(uint64_t seed, uint64_t env, uint32_t episode, int64_t n, double* uniform, double* normal,
                            double* gamma12) {
  Philox g = philox_init(seed, env, episode);
  for (int64_t i = 0; i < n; ++i) uniform[i] = philox_uniform(g);
  for (int64_t i = 0; i < n; ++i) normal[i] = philox_normal(g);
  for (int64_t i = 0; i < n; ++i) gamma12[i] = philox_gamma(g, 1.2);
}
extern "C" double emul_asin(double x) { return d_asin(x); }


// ---- wind noise (csrc/ble_noise.h) and the decoder tail (csrc/ble_decode.h), host build
static const float* host_grad_lut() {      // the kernels keep this table in LDS (grad_lut_fill)
  alignas(16) static float lut[kGradLutFloats];
  static bool filled = false;
  if (!filled) { grad_lut_fill(lut, 0, 1); filled = true; }
  return lut;
}
extern "C" void emul_simplex4(int64_t n, const float* x, const float* y, const float* z, const float* w, uint32_t seed,
                              float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = simplex4(x[i], y[i], z[i], w[i], seed, host_grad_lut());
}
extern "C" void emul_wind_noise(int64_t n, const float* x_m, const float* y_m, const float* pressure,
                                const int32_t* elapsed_s, uint64_t seed, const uint32_t* episode, float* noise_uv) {
  for (int64_t i = 0; i < n; ++i)
    wind_noise(x_m[i], y_m[i], pressure[i], elapsed_s[i], seed, (uint64_t)i, episode ? episode[i] : 0u, host_grad_lut(), &noise_uv[2 * i],
               &noise_uv[2 * i + 1]);
}
// the kernel's cached form: `cache` [kNoiseCacheRows][n] words holds the harmonics' seeds / offsets (fixture F14 writes recorded ones)
extern "C" void emul_wind_noise_cached(int64_t n, const float* x_m, const float* y_m, const float* pressure,
                                       const int32_t* elapsed_s, uint64_t seed, const uint32_t* episode, uint32_t* cache,
                                       float* noise_uv) {
  for (int64_t i = 0; i < n; ++i)
    wind_noise_cached(x_m[i], y_m[i], pressure[i], elapsed_s[i], seed, (uint64_t)i, (uint64_t)i, episode ? episode[i] : 0u, cache, n,
                      host_grad_lut(), &noise_uv[2 * i], &noise_uv[2 * i + 1]);
}
extern "C" void emul_decode_flow(int64_t n, const float* flow, float* grid) {
  int tap0[23]; float w1[23];
  for (int a = 0; a < 23; ++a) resize_tap(a, &tap0[a], &w1[a]);
  for (int64_t e = 0; e < n; ++e)
    for (int idx = 0; idx < 21 * 21 * 90; ++idx) {
      const int f = idx % 90, ij = idx / 90;
      float u, v;
      decode_flow_point(flow + e * 4410 + f, ij / 21, ij % 21, tap0, w1, &u, &v);
      grid[(e * 39690 + idx) * 2] = u; grid[(e * 39690 + idx) * 2 + 1] = v;
    }
}

// f_div_const / d_div_const against the true divisions over the ranges the transition uses them on (every float32 position from
// 2^-100 m to 600 km and zero, every whole second of the forecast's 48 h, every whole second of a night): the number of disagreeing inputs
extern "C" long long emul_check_constant_divisions() {
  long long bad = 0;
  const float top = 6.0e5f;
  uint32_t top_bits; __builtin_memcpy(&top_bits, &top, 4);
  const float low = 0x1p-100f;             // (below it x / 1000 is subnormal and the corrected quotient may sit one subnormal step off)
  uint32_t low_bits; __builtin_memcpy(&low_bits, &low, 4);
  bad += f_div_const(0.0f, 1000.0f, 1.0f / 1000.0f) != 0.0f;
  for (uint32_t b = low_bits; b <= top_bits; ++b) {
    float x; __builtin_memcpy(&x, &b, 4);
    const float q = f_div_const(x, 1000.0f, 1.0f / 1000.0f), t = x / 1000.0f;
    const float qn = f_div_const(-x, 1000.0f, 1.0f / 1000.0f);
    bad += (q != t) + (qn != -t);
  }
  // the wind noise's coordinate / spacing: every mantissa (two binades: the corrected quotient of normal numbers does not depend on the
  // exponent) against each of the forty spacings
  for (int k = 0; k < 10; ++k) {
    const Harmonic hp = kHarmonics.h[k]; const HarmonicRcp hr = kHarmonicRcp.h[k];
    const float b[4] = {hp.x_spacing, hp.y_spacing, hp.p_spacing, hp.t_spacing}, rb[4] = {hr.x, hr.y, hr.p, hr.t};
    for (int a = 0; a < 4; ++a) {
      bad += rb[a] != 1.0f / b[a];
      for (uint32_t m = 0x3f800000u; m < 0x40800000u; ++m) {
        float x; __builtin_memcpy(&x, &m, 4);
        bad += f_div_const(x, b[a], rb[a]) != x / b[a];
      }
    }
  }
  for (int32_t s = 0; s < 48 * 3600; ++s) bad += f_div_const((float)s, 3600.0f, 1.0f / 3600.0f) != (float)s / 3600.0f;
  for (int32_t s = -2 * 86400; s <= 2 * 86400; ++s) bad += d_div_const((double)s, 3600.0, 1.0 / 3600.0) != (double)s / 3600.0;
  return bad;
}
