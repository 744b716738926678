// ble_step_split.h -- the transition for SMALL batches: one environment on four wavefronts.
//
// ble_step_kernel (one lane per environment, one wave per workgroup) fills the chip at 65 536 environments = 1 024 waves =
// one per SIMD; a shard of 4 096 or 8 192 environments (BASELINE configs[1], one GPU's share of configs[3]) is 64 or 128
// waves on 1 024 SIMDs, and a lone wave issues one instruction per ~4.4 cycles whatever it is: the step takes the same
// 18.8 us however few environments there are.  Every right-hand side of a 10 s stride reads the OLD state only
// (balloon.py:322-325 commits afterwards), so the groups of state variables are independent inside a stride.  Here a
// workgroup is 4 waves = the 4 SIMDs of a CU, lane l of EVERY wave is environment 64 b + l, and each wave advances one group:
//
//   wave 0  vertical dynamics: p, T(p), ambient temperature, position           (stride_pressure, stride_ambient_advance)
//   wave 1  thermal model: internal temperature                                 (stride_internal_temperature)
//   wave 2  sun, power, battery, position again, reward                         (sun_fast one stride ahead, stride_power)
//   wave 3  envelope + ACS: volume, superpressure, mols of air                  (superpressure_volume_f64, stride_acs)
//
// After every stride the waves publish what they own in LDS (double-buffered by stride parity), meet at ONE workgroup
// barrier and read what they need: 6 doubles, 2 floats and 2 status words per lane.  The per-step part (atmosphere window +
// altitude layer | ephemeris for the flux | ephemeris + solar nodes | power + envelope layers) is spread the same way; the
// three safety layers publish their action MAPS (each layer is a function of the action alone once its state machine
// has moved) and every wave composes them.  Between agent steps every exchanged value is rounded to float32, exactly
// where ble_step_kernel stores its state as float32.
//
// Same lane functions as agent_step (ble_step_core.h), same expressions around them: the results are bit for bit those of
// ble_step_kernel (tests/test_gpu_parity.py::test_split_kernel_equals_one_lane_kernel).  Selected by the host entry
// points for n <= BLE_SPLIT_MAX_ENVS (4 waves x n / 64 workgroups still fit one wave per SIMD).
#pragma once
#include "ble_step_core.h"

namespace ble {

constexpr int kSplitWaves = 4;
constexpr int kSplitLanes = 64;

// LDS of one workgroup
struct SplitShared {
  double acs_poly[kAcsPolyDoubles];
  // stride exchange, [parity][lane]
  double p[2][kSplitLanes], t_amb[2][kSplitLanes];                         // wave 0
  double t_int[2][kSplitLanes];                                             // wave 1
  double vol[2][kSplitLanes], n_air[2][kSplitLanes], sp[2][kSplitLanes];    // wave 3
  float sin_el[2][kSplitLanes], batt[2][kSplitLanes];                       // wave 2: sun of the NEXT stride, battery
  uint32_t code2[2][kSplitLanes];                                           // wave 2: bit 0 battery empty, bit 1 next stride is day
  uint32_t code3[2][kSplitLanes];                                           // wave 3: 0 ok, kBurst, kZeroPressure
  // step exchange (written in the per-step part, read after its barrier; rewritten a step later, many barriers on)
  uint32_t map_alt[kSplitLanes], map_pow_env[kSplitLanes];                  // action maps of the safety layers (2 bits per input action)
  float sin_el0[kSplitLanes]; uint32_t day0[kSplitLanes];                   // wave 2 -> wave 1: the sun of stride 0
};

// a safety layer as a map action -> action, 2 bits per input action
BLE_FN uint32_t action_map(int r0, int r1, int r2) { return (uint32_t)r0 | ((uint32_t)r1 << 2) | ((uint32_t)r2 << 4); }
BLE_FN int action_apply(uint32_t map, int action) { return (int)((map >> (2 * action)) & 3u); }
// An action byte outside 0 .. 2 flies like STAY and is handed on as given (include/ble_abi.h): every layer returns UP when
// its state forces UP and the action itself otherwise -- i.e. what it answers to STAY decides.
BLE_FN int action_apply_any(uint32_t map_alt, uint32_t map_pow_env, int action) {
  const int known = action_apply(map_alt, action_apply(map_pow_env, action <= kUp ? action : kStay));
  return action <= kUp ? known : (known == kUp ? kUp : action);
}

struct SplitArgs {
  ble_state_f32 st;
  const uint8_t* action;
  const float* wind_grid;
  int64_t grid_env_stride;
  const float* noise_uv;
  float* reward;
  uint8_t* terminal;
  uint8_t* effective_action;
  uint32_t* err_flags;
  unsigned long long* active_count;
  int64_t n;
  int substeps, n_steps;
};

// One workgroup: 256 threads = 4 waves x 64 lanes; returns this thread's error flags.
BLE_FN uint32_t split_agent_steps(const SplitArgs& a, SplitShared& sh) {
  const int lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int64_t i = (int64_t)blockIdx.x * kSplitLanes + lane;
  const int64_t n = a.n;
  const bool in_range = i < n;
  const int substeps = a.substeps;
  uint32_t flags = 0;
  EnvRegs s = {};
  EnvConst c = {};
  EpisodeCacheRow cached = {};
  bool live = false;
  const ble_state_f32& st = a.st;
  if (in_range) {
    // every wave loads the whole state (one round trip): each needs most of it for its part of the per-step work
    s.status = st.status[i];
    s.x = st.x[i]; s.y = st.y[i]; s.p = st.pressure[i]; s.t_amb = st.ambient_temperature[i];
    s.t_int = st.internal_temperature[i]; s.vol = st.envelope_volume[i]; s.sp = st.superpressure[i];
    s.n_air = st.mols_air[i]; s.batt = st.battery_charge[i];
    s.t_elapsed = st.time_elapsed_s[i]; s.sunrise_h = st.sunrise_h_rel[i]; s.sunset = st.sunset_rel[i];
    s.alt_fsm = st.alt_fsm[i]; s.env_fsm = st.env_fsm[i]; s.paused = st.power_paused[i];
    c.lat0_deg = st.center_lat_deg[i]; c.lng0_deg = st.center_lng_deg[i];
    c.ir = st.upwelling_infrared[i]; c.alpha = st.alpha[i]; c.start_unix = st.start_unix[i];
    if (st.episode_cache != nullptr) cached = episode_cache_load(st.episode_cache, n, i);
    live = s.status == kOk;
  }
  for (int t = (int)threadIdx.x; t < kAcsPolyDoubles; t += kSplitWaves * kSplitLanes) sh.acs_poly[t] = kAcsPoly.c[t];
  __syncthreads();
  const bool was_live = live;
  int last_act = 0;
  EnvHoisted hc = {};
  if (live) {
    if (st.episode_cache != nullptr && episode_cache_hit(cached, c)) {
      hc = hoisted_from_cache(cached, c);
    } else {
      hc = hoist_constants(c);
      if (wave == 0 && st.episode_cache != nullptr) episode_cache_store(st.episode_cache, n, i, c, hc);
    }
  }
  int xk = 0;                        // exchange counter: stride parity across steps
  float acs_w = 0.0f, mdot = 0.0f, charge = 0.0f, load = 0.0f;

#pragma unroll 1
  for (int step = 0; step < a.n_steps; ++step) {
    const int64_t o = (int64_t)step * n + i;
    int act = 0;
    if (live) { act = a.action[o]; last_act = act; }

    // ================================================================ per-step part, one role per wave
    // carried across the strides (each wave uses its own subset)
    double p = (double)s.p, t_amb = (double)s.t_amb, t_int = (double)s.t_int, n_air = (double)s.n_air, vol = (double)s.vol,
           sp = (double)s.sp;
    float x = s.x, y = s.y, batt = s.batt;
    const float p0_in = s.p;
    const float x_start = s.x, y_start = s.y;
    const int32_t t_start = s.t_elapsed;
    AtmWindow win = {};
    LayerCursor lc = {};
    double t_at_p = 0.0;
    float u = 0.0f, v = 0.0f, fl0 = 0.0f, dfl = 0.0f, oms_c0 = 0.0f, oms_c1 = 0.0f, oms_c2 = 0.0f;
    SunState sun_next = {};           // wave 2: the sun of the stride about to run
    float sun_sin = 0.0f; bool sun_day = false;      // wave 1: (sin el, day) of the stride about to run
    const double q_earth = hc.q_earth;
    uint32_t my_map = action_map(0, 1, 2);

    if (live) {
      if (wave == 0 || wave == 2) {
        // wind at the PRE-step position/time (balloon_arena.py:194,270-275); waves 0 and 2 both need it (position; solar nodes)
        const WindQuery wq = wind_query(s.x, s.y, s.p, s.t_elapsed);
        WindCorners corners;
        wind_gather(a.wind_grid + i * a.grid_env_stride, wq, &corners);
        float nu = 0.0f, nv = 0.0f;
        if (a.noise_uv) { nu = a.noise_uv[2 * i]; nv = a.noise_uv[2 * i + 1]; }
        if (wave == 0) {
          win = atm_window_from(hc.atm, (double)c.alpha, p, &flags);
          double altitude;
          atm_at_pressure_f64(win, (double)c.alpha, p, &altitude, &t_at_p);
          lc.lay = 0; lc.lapse_cur = win.lapse_0; lc.cur_hi = win.pb; lc.cur_lo = win.pt;
          uint8_t f0 = s.alt_fsm, f1 = s.alt_fsm, f2 = s.alt_fsm;
          const int r0 = altitude_safety(kDown, altitude, &f0), r1 = altitude_safety(kStay, altitude, &f1),
                    r2 = altitude_safety(kUp, altitude, &f2);
          s.alt_fsm = f0;                                   // (the state machine moves independently of the action)
          my_map = action_map(r0, r1, r2);
        } else {
          const int64_t t0 = c.start_unix + (int64_t)s.t_elapsed;
          const Ephemeris e0 = ephemeris(t0);
          const float step_s = (float)(10 * substeps);
          {
            double sod;
            if (__builtin_expect(t0 >= 0 && t0 < 4294967296LL, 1)) sod = (double)((uint32_t)t0 % 86400u);
            else { int64_t m = t0 % 86400; sod = (double)(m < 0 ? m + 86400 : m); }
            // (the wind must be blended before the nodes: they sit at x0 + k u)
            wind_blend_corners(corners, wq, &u, &v);
            u += nu; v += nv;
            const double b0 = sod * (1.0 / 240.0) + 0.25 * e0.eot_min + (double)c.lng0_deg;
            const double half_db = 0.5 * ((double)step_s * (1.0 / 240.0) + 0.25 * (double)(e0.eot_min_rate * step_s));  // deg
            double sb0, cb0;
            sincos_f64(b0 * (kPiD / 180.0), &sb0, &cb0);
            const double hr = half_db * (kPiD / 180.0), h2 = hr * hr;
            const double shh = hr * d_fma(h2, d_fma(h2, d_fma(h2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
            const double ch = d_fma(h2, d_fma(h2, d_fma(h2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
            const double sb1 = sb0 * ch + cb0 * shh, cb1 = cb0 * ch - sb0 * shh;
            const double sb2 = sb1 * ch + cb1 * shh, cb2 = cb1 * ch - sb1 * shh;
            const double sl0 = hc.sin_lat0, cl0 = hc.cos_lat0;
            const double x0 = (double)s.x, y0 = (double)s.y;
            const double dx = (double)u * (5.0 * (double)substeps), dy = (double)v * (5.0 * (double)substeps);  // half step
            const double sd0 = (double)e0.sin_decl, cd0 = d_sqrt_fast(d_fma(-sd0, sd0, 1.0));
            const double hsd = 0.5 * (double)(e0.sin_decl_rate * step_s), hcd = -(sd0 * d_rcp(cd0)) * hsd;
            const double f0 = sun_one_minus_sin_f64(sl0, cl0, x0, y0, sb0, cb0, sd0, cd0);
            const double f1 = sun_one_minus_sin_f64(sl0, cl0, x0 + dx, y0 + dy, sb1, cb1, sd0 + hsd, cd0 + hcd);
            const double f2 = sun_one_minus_sin_f64(sl0, cl0, x0 + 2.0 * dx, y0 + 2.0 * dy, sb2, cb2, sd0 + 2.0 * hsd, cd0 + 2.0 * hcd);
            const double m = 0.5 * (double)substeps;
            oms_c0 = (float)f0;
            oms_c1 = (float)((-f2 + 4.0 * f1 - 3.0 * f0) / (2.0 * m));
            oms_c2 = (float)((f2 - 2.0 * f1 + f0) / (2.0 * m * m));
          }
        }
        if (wave == 0) {
          wind_blend_corners(corners, wq, &u, &v);
          u += nu; v += nv;
        }
      } else if (wave == 1) {
        const Ephemeris e0 = ephemeris(c.start_unix + (int64_t)s.t_elapsed);
        fl0 = e0.flux; dfl = e0.flux_rate * 10.0f;
        flags |= hc.flags;
        // total_absorptivity's range check (thermal.py:142-145) on the balloon's own temperature, first value of the step
        flags |= (s.t_int < 12.3f) ? kFlagAbsorptivity : 0u;
      } else {
        // the power and envelope layers on the pre-step state (balloon.py:304-313); both state machines move independently
        // of the action
        int32_t sr0 = s.sunrise_h, ss0 = s.sunset, sr1 = s.sunrise_h, ss1 = s.sunset, sr2 = s.sunrise_h, ss2 = s.sunset;
        uint8_t pa0 = s.paused, pa1 = s.paused, pa2 = s.paused, e0 = s.env_fsm, e1 = s.env_fsm, e2 = s.env_fsm;
        const int q0 = envelope_safety(power_safety(kDown, s.t_elapsed, s.batt, &sr0, &ss0, &pa0), s.sp, &e0);
        const int q1 = envelope_safety(power_safety(kStay, s.t_elapsed, s.batt, &sr1, &ss1, &pa1), s.sp, &e1);
        const int q2 = envelope_safety(power_safety(kUp, s.t_elapsed, s.batt, &sr2, &ss2, &pa2), s.sp, &e2);
        s.sunrise_h = sr0; s.sunset = ss0; s.paused = pa0; s.env_fsm = e0;
        my_map = action_map(q0, q1, q2);
      }
    }
    // wave 2: the sun of stride 0 (and of every later stride one stride ahead)
    auto sun_at = [&](int kk) -> SunState {
      const float fkk = (float)kk;
      bool near;
      SunState r = sun_fast(f_fma(fkk, f_fma(fkk, oms_c2, oms_c1), oms_c0), &near);
      if (__builtin_expect(near, 0)) {
        const double dk = 10.0 * (double)kk;
        r = sun_exact((double)c.lat0_deg, (double)c.lng0_deg, d_fma(dk, (double)u, (double)x_start), d_fma(dk, (double)v, (double)y_start),
                      c.start_unix + (int64_t)(t_start + 10 * kk));
      }
      return r;
    };
    if (wave == 2 && live) {
      sun_next = sun_at(0);
      sh.sin_el0[lane] = sun_next.sin_el;
      sh.day0[lane] = sun_next.day ? 1u : 0u;
    }
    if (wave == 0) sh.map_alt[lane] = my_map;
    if (wave == 3) sh.map_pow_env[lane] = my_map;
    __syncthreads();
    const int eff = action_apply_any(sh.map_alt[lane], sh.map_pow_env[lane], act);
    if (wave == 1) { sun_sin = sh.sin_el0[lane]; sun_day = sh.day0[lane] != 0u; }

    // ================================================================ the strides
    bool active = live;
    int k_done = 0, last_rd = 0;       // strides this lane ran; the exchange parity of its last one
    int status3 = kOk; bool batt_empty = false;
#pragma unroll 1
    for (int k = 0; k < substeps; ++k) {
      if (__ballot(active) == 0ull) break;      // (the same decision in all four waves: `active` derives from shared words)
      const int wr = (xk + 1) & 1;
      if (active) {
        const double rp = d_rcp(p);
        if (wave == 0) {
          const double yc = inv_cbrt_volume(vol);
          const double p_new = stride_pressure(win, lc, p, rp, vol, n_air, t_amb, t_at_p, yc);
          x = f_fma(u, kStride, x); y = f_fma(v, kStride, y);
          t_amb = t_at_p;
          t_at_p = stride_ambient_advance(win, lc, p, rp, t_at_p, p_new);
          p = p_new;
          sh.p[wr][lane] = p; sh.t_amb[wr][lane] = t_amb;
        } else if (wave == 1) {
          const float pf = (float)p;
          const float flux = f_fma((float)k, dfl, fl0);
          const double yc = inv_cbrt_volume(vol);
          const float att = solar_attenuation(sun_sin, pf, sun_day);
          t_int = stride_internal_temperature(vol, yc, t_int, t_amb, p, flux, att, q_earth);
          sh.t_int[wr][lane] = t_int;
        } else if (wave == 2) {
          const float pf = (float)p;
          const SunState sun = sun_next;
          const float att = solar_attenuation(sun.sin_el, pf, sun.day);
          // the ACS power of this stride, as wave 3 evaluates it (acs_down_poly on the same inputs)
          double w_down, mdot_down;
          acs_down_poly(sh.acs_poly, d_max(sp, 0.0) * rp, &w_down, &mdot_down);
          acs_w = eff == kDown ? (float)w_down : 0.0f;
          stride_power(sun, att, acs_w, &charge, &load, &batt);
          x = f_fma(u, kStride, x); y = f_fma(v, kStride, y);
          sun_next = sun_at(k + 1);
          sh.sin_el[wr][lane] = sun_next.sin_el; sh.batt[wr][lane] = batt;
          sh.code2[wr][lane] = (batt <= 0.0f ? 1u : 0u) | (sun_next.day ? 2u : 0u);
        } else {
          double vol_new, sp_new, mdot_d;
          superpressure_volume_f64(n_air, t_int, p, rp, &vol_new, &sp_new);
          stride_acs(sh.acs_poly, eff, sp, p, rp, t_int, &acs_w, &mdot_d);
          mdot = (float)mdot_d;
          n_air = stride_mols_air(n_air, mdot_d);
          vol = vol_new; sp = sp_new;
          // balloon.py:479-482: burst above 2 380 Pa, zero pressure at <= 0 (later checks override earlier ones)
          sh.code3[wr][lane] = sp_new <= 0.0 ? (uint32_t)kZeroPressure : (!(sp_new <= 2380.0) ? (uint32_t)kBurst : 0u);
          sh.vol[wr][lane] = vol; sh.n_air[wr][lane] = n_air; sh.sp[wr][lane] = sp;
        }
      }
      __syncthreads();
      ++xk;
      if (active) {
        const int rd = xk & 1;
        k_done = k + 1; last_rd = rd;
        const uint32_t c2 = sh.code2[rd][lane], c3 = sh.code3[rd][lane];
        if (wave == 0) { vol = sh.vol[rd][lane]; n_air = sh.n_air[rd][lane]; }
        else if (wave == 1) { p = sh.p[rd][lane]; t_amb = sh.t_amb[rd][lane]; vol = sh.vol[rd][lane]; sun_sin = sh.sin_el[rd][lane]; sun_day = (c2 & 2u) != 0u; }
        else if (wave == 2) { p = sh.p[rd][lane]; sp = sh.sp[rd][lane]; }
        else { p = sh.p[rd][lane]; t_int = sh.t_int[rd][lane]; }
        status3 = (int)c3; batt_empty = (c2 & 1u) != 0u;
        if (status3 != kOk || batt_empty) active = false;          // balloon.py:327-328
      }
    }

    // ================================================================ end of the step: float32 state, status, reward
    if (live) {
      const int rd = last_rd;         // this lane's last stride (it may have ended before the others): every wave fetches what it does not own
      int status = status3;
      if (batt_empty) status = kOutOfPower;
      if (wave != 0) { p = sh.p[rd][lane]; t_amb = sh.t_amb[rd][lane]; }
      if (wave != 1) t_int = sh.t_int[rd][lane];
      if (wave != 3) { vol = sh.vol[rd][lane]; n_air = sh.n_air[rd][lane]; sp = sh.sp[rd][lane]; }
      if (wave != 2) batt = sh.batt[rd][lane];
      s.p = (float)p; s.t_amb = (float)t_amb; s.t_int = (float)t_int; s.vol = (float)vol;
      s.sp = (float)sp; s.n_air = (float)n_air; s.batt = batt;
      s.t_elapsed += 10 * k_done;
      s.status = (uint8_t)status;
      if (wave == 0 || wave == 2) { s.x = x; s.y = y; }       // (waves 1 and 3 never read the position)
      if (wave == 1) flags |= (s.t_int < 12.3f) ? kFlagAbsorptivity : 0u;
      if (wave == 2) {
        s.acs_power = acs_w; s.charge = charge; s.load = load;
        // solar_atmospheric_attenuation's range check (solar.py:194-197); p moves < 3 kPa per step
        flags |= (s.p > 101325.0f || s.p < 0.0f || p0_in > 101325.0f || p0_in < 0.0f) ? kFlagSolarRange : 0u;
        // ---- reward (env/balloon_env.py:44-102), on the post-step state
        float r = reward_distance(s.x, s.y);
        if (act == kDown) {   // last_command is the RAW action (balloon.py:286)
          const SunState sun = sun_next;             // == sun_at(k_done)
          const float pw = solar_power(sun, solar_attenuation(sun.sin_el, s.p, sun.day));
          const bool excess = (pw > kDayLoad) && ((double)s.batt / 3058.56 > 0.99);   // balloon.py:231-238
          if (!excess) {
            const float scale = f_clamp((s.acs_power - 100.0f) * (1.0f / 200.0f), 0.0f, 1.0f);
            r *= f_fma(-0.3f, scale, 0.95f);
          }
        }
        if (!(isfinite(s.p) && isfinite(s.t_int) && isfinite(s.x) && isfinite(s.y) && isfinite(s.batt))) flags |= kFlagNonFinite;
        a.reward[o] = r;
        a.terminal[o] = s.status != kOk;
      }
      if (wave == 3) {
        s.acs_power = acs_w; s.mdot = mdot;
        if (a.effective_action) a.effective_action[o] = (uint8_t)eff;
      }
    } else if (in_range) {  // balloon.py:288-290 raises; a vectorised env freezes the lane instead
      if (wave == 2) { a.reward[o] = 0.0f; a.terminal[o] = 1; }
      if (wave == 3 && a.effective_action) a.effective_action[o] = a.action[o];
    }
    if (wave == 0 && a.active_count) {
      const unsigned long long m = __ballot(live);
      if (lane == 0 && m)
        atomicAdd(a.active_count + (int64_t)step * BLE_COUNT_SLOTS + (blockIdx.x & (BLE_COUNT_SLOTS - 1)), (unsigned long long)__popcll(m));
    }
    live = live && s.status == kOk;
    // the safety layers' state and the position live in their own waves; the next step's per-step part needs: wave 0 x, y
    // (own), wave 2 x, y (own), wave 3 sunrise / sunset / paused / env_fsm (own), wave 0 alt_fsm (own): nothing to exchange
  }

  if (was_live) {
    if (wave == 0) {
      st.x[i] = s.x; st.y[i] = s.y; st.pressure[i] = s.p; st.ambient_temperature[i] = s.t_amb;
      st.time_elapsed_s[i] = s.t_elapsed; st.alt_fsm[i] = s.alt_fsm;
    } else if (wave == 1) {
      st.internal_temperature[i] = s.t_int; st.status[i] = s.status; st.last_command[i] = (uint8_t)last_act;
    } else if (wave == 2) {
      st.battery_charge[i] = s.batt; st.solar_charging[i] = s.charge; st.power_load[i] = s.load;
    } else {
      st.envelope_volume[i] = s.vol; st.superpressure[i] = s.sp; st.mols_air[i] = s.n_air;
      st.acs_power[i] = s.acs_power; st.acs_mass_flow[i] = s.mdot;
      st.sunrise_h_rel[i] = s.sunrise_h; st.sunset_rel[i] = s.sunset; st.env_fsm[i] = s.env_fsm; st.power_paused[i] = s.paused;
    }
  }
  return flags;
}

}  // namespace ble
