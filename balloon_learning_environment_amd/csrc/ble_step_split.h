// ble_step_split.h -- the transition for SMALL batches: one environment on four wavefronts.
//
// ble_step_kernel (one lane per environment, one wave per workgroup) fills the chip at 65 536 environments = 1 024 waves =
// one per SIMD; a shard of 4 096 or 8 192 environments (BASELINE configs[1], one GPU's share of configs[3]) is 64 or 128
// waves on 1 024 SIMDs, and a lone wave issues one instruction per ~4.4 cycles whatever it is: the step takes the same
// 18 us however few environments there are.  Every right-hand side of a 10 s stride reads the OLD state only
// (balloon.py:322-325 commits afterwards), so the groups of state variables are independent inside a stride.  Here a
// workgroup is 4 waves = the 4 SIMDs of a CU, lane l of EVERY wave is environment 64 b + l, and each wave advances one group:
//
//   wave 0  vertical dynamics: p, T(p), ambient temperature, position           (stride_pressure, stride_ambient_advance);
//           per step: atmosphere window, altitude layer; the step's reward, terminal flag and range checks
//   wave 1  thermal model: internal temperature                                 (stride_internal_temperature)
//   wave 2  the sun, one stride AHEAD (it depends on the stride index only): sin el, panel factor, day -- the rare exact
//           solar chain runs here, off everybody's critical path -- and the envelope: volume, superpressure
//           (superpressure_volume_f64); per step: wind lookup, first solar node, the sun of stride 0
//   wave 3  ACS + power: mols of air, battery                                   (stride_acs, stride_power_from_factor)
//
// After every stride the waves publish what they own in LDS (double-buffered by stride parity), meet at ONE workgroup
// barrier and read what they need.  The per-step part is spread the same way around two barriers (atmosphere window |
// ephemeris | wind lookup | power + envelope layers; then the altitude layer and one solar node each on waves 1, 2, 3, wave 2
// the sun of stride 0 with it); the safety layers publish their action MAPS (a layer is a function of the action alone once
// its state machine has moved).  With a wind-noise generator (ABI 3) a step starts with the ten harmonic values, one 4-D
// simplex evaluation each, spread over the waves, and one more barrier.  The same template runs as TWO wavefronts per
// environment ({vertical, thermal} | {sun + envelope, ACS + power}: ble_step_pair_kernel, experiment builds only -- -DBLE_WITH_PAIR_FORM).
// Between agent steps every value goes through float32, exactly where ble_step_kernel keeps its state as float32.
//
// The stride loops are straight-line: a lane whose episode ended (or was over on entry) keeps computing on a shadow of its
// state and only its LDS writes are masked, so nobody carries a divergent `break`; every final value of a step is fetched
// from the LDS slot of the lane's own last stride.
//
// Same lane functions as agent_step (ble_step_core.h), same expressions around them: the results are bit for bit those of
// ble_step_kernel (tests/test_gpu_parity.py::test_split_kernel_equals_one_lane_kernel).  Selected by the host entry
// points for n <= BLE_SPLIT_MAX_ENVS = 32 768 (two waves per SIMD: measured 1.4x the one-lane kernel there, 1.9x at <= 16 384
// environments where every wave has a SIMD to itself; at 65 536 -- four waves per SIMD -- the one-lane kernel wins).
#pragma once
#include "ble_noise.h"
#include "ble_step_core.h"

// Timing build (profiles/build_variant.sh split_timing -DBLE_SPLIT_TIMING): every wave adds the shader-clock cycles it spends in
// the sections of a step to active_count[role * 8 + section] (used as a debug buffer: pass >= 32 zeroed uint64).  Sections:
// 0 per-step part, 1 waits at the per-step barriers, 2 stride right-hand sides, 3 publish + wait at the stride barrier,
// 4 reads after the barrier, 5 end of the step.  Empty in the product build.
#ifdef BLE_SPLIT_TIMING
#define BLE_SPLIT_T_DECL long long t_acc[6] = {0, 0, 0, 0, 0, 0}; long long t_last = (long long)__builtin_readcyclecounter()
#define BLE_SPLIT_T(sec) do { const long long t_now = (long long)__builtin_readcyclecounter(); t_acc[sec] += t_now - t_last; t_last = t_now; } while (0)
#define BLE_SPLIT_T_FLUSH(dbg, role) do { if ((dbg) != nullptr && lane == 0) { for (int q = 0; q < 6; ++q) atomicAdd((dbg) + (role) * 8 + q, (unsigned long long)t_acc[q]); atomicAdd((dbg) + (role) * 8 + 7, 1ull); } } while (0)
#else
#define BLE_SPLIT_T_DECL do {} while (0)
#define BLE_SPLIT_T(sec) do {} while (0)
#define BLE_SPLIT_T_FLUSH(dbg, role) do {} while (0)
#endif

namespace ble {

constexpr int kSplitWaves = 4;
// Which role evaluates the sun of the NEXT stride (it depends on the stride index alone): 2 = the envelope wave (rounds 4-5),
// 1 = the thermal wave, 3 = the ACS + power wave.  A/B knob (profiles/r05_raw/split_sun_role_ab.txt).
#ifndef BLE_SPLIT_SUN_ROLE
#define BLE_SPLIT_SUN_ROLE 2
#endif
constexpr int kSplitLanes = 64;

// LDS of one workgroup (12.6 KB)
struct SplitShared {
  double acs_poly[kAcsPolyDoubles];
  // ---- stride exchange, [parity][lane]; a lane's slots are written only while its episode runs
  double p[2][kSplitLanes], t_amb[2][kSplitLanes];                          // wave 0
  float x[2][kSplitLanes], y[2][kSplitLanes];
  double t_int[2][kSplitLanes];                                              // wave 1
  float sin_el[2][kSplitLanes], panel[2][kSplitLanes]; uint32_t day[2][kSplitLanes];      // wave 2: the sun of the NEXT stride
  double vol[2][kSplitLanes], sp[2][kSplitLanes];                            // wave 2
  uint32_t code_sp[2][kSplitLanes];                                          // wave 2: 0, kBurst or kZeroPressure after the stride
  double n_air[2][kSplitLanes];                                              // wave 3
  float batt[2][kSplitLanes], acs_w[2][kSplitLanes], mdot[2][kSplitLanes], charge[2][kSplitLanes], load[2][kSplitLanes];
  uint32_t code_batt[2][kSplitLanes];                                        // wave 3: kOutOfPower after the stride, or 0
  // ---- step exchange (written in the per-step part, read after its barriers; rewritten a step later, many barriers on)
  double eot_min[kSplitLanes]; float eph[3][kSplitLanes];                    // wave 1 -> waves 2, 3: the ephemeris fields the hour-angle nodes need
  double node_f[2][kSplitLanes];                                             // waves 1, 3 -> the sun role: the middle and end nodes of the step
  double node_f0[kSplitLanes];                                               // wave 2 -> the sun role (when that is another wave): the first node
  float u[kSplitLanes], v[kSplitLanes];                                      // wave 2 -> all: the wind of the step
  float sin_el0[kSplitLanes], panel0[kSplitLanes]; uint32_t day0[kSplitLanes];   // wave 2 -> waves 1, 3: the sun of stride 0
  uint32_t map_alt[kSplitLanes];                                             // wave 0 -> wave 3: the altitude layer's action map
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

// the in-kernel wind-noise generator of a fused rollout (ABI 3) in this form: the harmonics' draws of the workgroup's
// environments (fetched once per launch) and the ten harmonic values of a step, each evaluated by one of the waves
template <bool kNoise> struct SplitNoiseShared {};
template <> struct SplitNoiseShared<true> {
  alignas(16) float grad_lut[kGradLutFloats];      // the noise primitive's gradient weights (ble_noise.h)
  uint32_t draws[50][kSplitLanes];
  float nz[10][kSplitLanes];
};

struct SplitArgs {
  StateDev st;
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
  StepNoise gen;                     // read by the <kNoise> instantiations only
};

// One wave of a workgroup of 256 threads = 4 waves x 64 lanes; returns this thread's error flags.  `wave` is a
// compile-time constant: every role is its own instantiation -- its own registers, its own loops -- and the four agree on
// the number of barriers per step (two in the per-step part -- three with the noise generator --, one per stride) by construction.
template <int kWaves, int wave, bool kNoise>
BLE_FN uint32_t split_agent_steps(const SplitArgs& a, SplitShared& sh, SplitNoiseShared<kNoise>& shn) {
  static_assert((kWaves == 4 || kWaves == 2) && wave >= 0 && wave < kWaves, "four waves = one role each, two waves = two roles each");
  // the roles this wave plays: 0 vertical, 1 thermal, 2 sun + envelope, 3 ACS + power (kWaves == 2: {0, 1} and {2, 3})
  constexpr bool r0 = kWaves == 4 ? wave == 0 : wave == 0, r1 = kWaves == 4 ? wave == 1 : wave == 0;
  constexpr bool r2 = kWaves == 4 ? wave == 2 : wave == 1, r3 = kWaves == 4 ? wave == 3 : wave == 1;
  constexpr bool rs = BLE_SPLIT_SUN_ROLE == 1 ? r1 : (BLE_SPLIT_SUN_ROLE == 3 ? r3 : r2);      // the role that evaluates the strides' sun
  const int lane = (int)threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * kSplitLanes + lane;
  const int64_t n = a.n;
  const bool in_range = i < n;
  const int substeps = a.substeps;
  uint32_t flags = 0;
  EnvRegs s = {};
  EnvConst c = {};
  EpisodeCacheRow cached = {};
  bool live = false;
  const StateDev& st = a.st;
  // a lane beyond the batch flies a harmless shadow (never stored): finite, inside the first layer of the atmosphere
  s.p = 9000.0f; s.t_amb = 215.0f; s.t_int = 220.0f; s.vol = 1810.0f; s.sp = 300.0f; s.n_air = 1500.0f; s.batt = 2000.0f;
  s.status = kBurst; s.sunrise_h = 43200; s.sunset = 21600;
  c.ir = 300.0f; c.alpha = 0.5f; c.start_unix = 1356998400;
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
  for (int t = (int)threadIdx.x; t < kAcsPolyDoubles; t += kWaves * kSplitLanes) sh.acs_poly[t] = kAcsPoly.c[t];
  if constexpr (kNoise) {            // the harmonics' seeds and offsets of the workgroup's environments, once per launch
    grad_lut_fill(shn.grad_lut, (int)threadIdx.x, kWaves * kSplitLanes);
    if (wave == 0 && in_range)
      noise_draws_fetch(a.gen.seed, (uint64_t)i, (uint64_t)(i + a.gen.env_offset), a.gen.episode ? a.gen.episode[i] : 0u, a.gen.harmonic_cache, n,
                        &shn.draws[0][lane], kSplitLanes);
  }
  __syncthreads();
  const bool was_live = live;
  int last_act = 0;
  EnvHoisted hc;
  if (in_range && st.episode_cache != nullptr && episode_cache_hit(cached, c)) {
    hc = hoisted_from_cache(cached, c);
  } else {
    hc = hoist_constants(c);
    if (r0 && live && st.episode_cache != nullptr) episode_cache_store(st.episode_cache, n, i, c, hc);
  }
  int xk = 0;                        // exchange counter: stride parity, running across the steps
  BLE_SPLIT_T_DECL;

#pragma unroll 1
  for (int step = 0; step < a.n_steps; ++step) {
    const int64_t o = (int64_t)step * n + i;
    int act = kStay;
    if (in_range) act = a.action[o];
    if (live) last_act = act;

    // ================================================================ per-step part (all lanes, live or shadow)
    // the carried state of the strides; every wave advances its own part and refreshes the rest from LDS
    double p = (double)s.p, t_amb = (double)s.t_amb, t_int = (double)s.t_int, n_air = (double)s.n_air, vol = (double)s.vol,
           sp = (double)s.sp;
    float x = s.x, y = s.y, batt = s.batt;
    float acs_w = 0.0f, mdot = 0.0f, charge = 0.0f, load = 0.0f;
    float u = 0.0f, v = 0.0f;
    const float p0_in = s.p, x_start = s.x, y_start = s.y;
    const int32_t t_start = s.t_elapsed;
    const int64_t t0 = c.start_unix + (int64_t)s.t_elapsed;
    const float step_s = (float)(10 * substeps);
    uint32_t step_flags = 0;
    // role state
    AtmWindow win = {}; LayerCursor lc = {}; double t_at_p = 0.0;       // wave 0
    float fl0 = 0.0f, dfl = 0.0f; Ephemeris e0 = {};                     // wave 1 (e0: the part solar_nodes_time reads, on waves 1, 2, 3)
    SunQuadratic sq = {}; double node_f0 = 0.0;                          // wave 2
    float sun_sin = 0.0f, sun_panel = 0.0f; bool sun_day = false;        // waves 1, 3: the sun of the stride about to run
    uint32_t map_pow_env = 0; int eff = kStay;                           // wave 3

    if (r1) {
      e0 = ephemeris(t0);
      fl0 = e0.flux; dfl = e0.flux_rate * 10.0f;
      // what the hour-angle nodes of the step need of it (solar_nodes_time)
      sh.eot_min[lane] = e0.eot_min; sh.eph[0][lane] = e0.eot_min_rate; sh.eph[1][lane] = e0.sin_decl; sh.eph[2][lane] = e0.sin_decl_rate;
      step_flags |= hc.flags;
      // total_absorptivity's range check (thermal.py:142-145) on the balloon's own temperature, first value of the step
      step_flags |= (s.t_int < 12.3f) ? kFlagAbsorptivity : 0u;
    }
    if (r3) {
      // the power and envelope layers on the pre-step state (balloon.py:304-313); both state machines move independently of
      // the action, so the three evaluations share everything but the final selects
      int32_t sr0 = s.sunrise_h, ss0 = s.sunset, sr1 = s.sunrise_h, ss1 = s.sunset, sr2 = s.sunrise_h, ss2 = s.sunset;
      uint8_t pa0 = s.paused, pa1 = s.paused, pa2 = s.paused, f0 = s.env_fsm, f1 = s.env_fsm, f2 = s.env_fsm;
      const int q0 = envelope_safety(power_safety(kDown, s.t_elapsed, s.batt, &sr0, &ss0, &pa0), s.sp, &f0);
      const int q1 = envelope_safety(power_safety(kStay, s.t_elapsed, s.batt, &sr1, &ss1, &pa1), s.sp, &f1);
      const int q2 = envelope_safety(power_safety(kUp, s.t_elapsed, s.batt, &sr2, &ss2, &pa2), s.sp, &f2);
      if (live) { s.sunrise_h = sr0; s.sunset = ss0; s.paused = pa0; s.env_fsm = f0; }
      map_pow_env = action_map(q0, q1, q2);
    }
    if constexpr (kNoise) {
      // WindField.get_ground_truth's noise term at the pre-step position (wind_field.py:125-145): the ten harmonic values,
      // each one 4-D simplex evaluation, spread over the waves (harmonic k on wave k mod kWaves); role 2 adds them up below
      // in the reference's order -- the same functions in the same order as ble_wind_noise_f32
      if (in_range) {
        float x_km, y_km, t_h;
        noise_coords(s.x, s.y, s.t_elapsed, &x_km, &y_km, &t_h);
#pragma unroll 1
        for (int k = wave; k < 10; k += kWaves)
          shn.nz[k][lane] = noise_harmonic_value(k / 5, k % 5, harmonic_draw_from_rows(&shn.draws[0][lane], kSplitLanes, k), x_km, y_km, s.p, t_h,
                                                 shn.grad_lut);
      }
      __syncthreads();                                 // ---- barrier 0 (noise only): the harmonic values are there
    }
    // role 2: the wind at the PRE-step position/time (balloon_arena.py:194,270-275): WindField.get_ground_truth = forecast + noise
    if (r2) {
      const WindQuery wq = wind_query(s.x, s.y, s.p, s.t_elapsed);
      WindCorners corners;
      wind_gather(a.wind_grid + (in_range ? i : 0) * a.grid_env_stride, wq, &corners);
      float nu = 0.0f, nv = 0.0f;
      if constexpr (kNoise) {
        if (in_range) wind_noise_from_values(&shn.nz[0][lane], kSplitLanes, &nu, &nv);
      } else {
        if (a.noise_uv && in_range) { nu = a.noise_uv[2 * i]; nv = a.noise_uv[2 * i + 1]; }
      }
      wind_blend_corners(corners, wq, &u, &v);
      u += nu; v += nv;
      sh.u[lane] = u; sh.v[lane] = v;
    }
    BLE_SPLIT_T(0);
    __syncthreads();                                   // ---- barrier 1: the ephemeris and the wind are there
    BLE_SPLIT_T(1);
    if (!r2) { u = sh.u[lane]; v = sh.v[lane]; }
    if (!r1) { e0.eot_min = sh.eot_min[lane]; e0.eot_min_rate = sh.eph[0][lane]; e0.sin_decl = sh.eph[1][lane]; e0.sin_decl_rate = sh.eph[2][lane]; }
    if (r0) {
      // (after barrier 1: this wave came to it with the previous step's reward, and what gates barrier 2 is wave 2's node)
      win = atm_window_from(hc.atm, (double)c.alpha, p, &step_flags);
      lc.lay = 0; lc.lapse_cur = win.lapse_0; lc.kl_cur = (-kAirSpecificGasD / 9.80665) * win.lapse_0; lc.cur_hi = win.pb; lc.cur_lo = win.pt;
      double altitude;
      atm_at_pressure_f64(win, (double)c.alpha, p, &altitude, &t_at_p);
      uint8_t f0 = s.alt_fsm, f1 = s.alt_fsm, f2 = s.alt_fsm;
      const int q0 = altitude_safety(kDown, altitude, &f0), q1 = altitude_safety(kStay, altitude, &f1),
                q2 = altitude_safety(kUp, altitude, &f2);
      if (live) s.alt_fsm = f0;                         // (the state machine moves independently of the action)
      sh.map_alt[lane] = action_map(q0, q1, q2);
    }
    if (r1 || r2 || r3) {
      // the three solar nodes of the step, one per role (role 2 has the wind in registers, role 1 the ephemeris); each wave
      // forms the time-only half itself -- the same function of the same four numbers -- and uses its own nodes of it
      const SolarNodes nd = solar_nodes_time(e0, t0, c.lng0_deg, step_s);
      if (r1) sh.node_f[0][lane] = solar_node<1>(nd, hc.sin_lat0, hc.cos_lat0, s.x, s.y, u, v, substeps);
      if (r3) sh.node_f[1][lane] = solar_node<2>(nd, hc.sin_lat0, hc.cos_lat0, s.x, s.y, u, v, substeps);
      if (r2) {
        node_f0 = solar_node<0>(nd, hc.sin_lat0, hc.cos_lat0, s.x, s.y, u, v, substeps);
        // the sun of stride 0 needs the first node alone: the quadratic at index 0 is its constant term (the other two
        // coefficients are finite), so it is ready at the same barrier as the nodes
        sq.c0 = (float)node_f0; sq.c1 = 0.0f; sq.c2 = 0.0f; sq.thr = sun_thresholds(node_f0, substeps);
        const SunState sun0 = sun_at_stride(0, sq, c, u, v, x_start, y_start, t_start);
        sun_sin = sun0.sin_el; sun_panel = solar_panel_factor(sun0); sun_day = sun0.day;
        sh.sin_el0[lane] = sun_sin; sh.panel0[lane] = sun_panel; sh.day0[lane] = sun_day ? 1u : 0u;
        if (!rs) sh.node_f0[lane] = node_f0;
      }
    }
    BLE_SPLIT_T(0);
    __syncthreads();                                   // ---- barrier 2: the nodes, the sun of stride 0, the altitude layer's map
    BLE_SPLIT_T(1);
    if (rs) sq = solar_node_coefs(r2 ? node_f0 : sh.node_f0[lane], sh.node_f[0][lane], sh.node_f[1][lane], substeps);
    if ((r1 || r3) && !r2) { sun_sin = sh.sin_el0[lane]; sun_panel = sh.panel0[lane]; sun_day = sh.day0[lane] != 0u; }
    if (r3) eff = action_apply_any(sh.map_alt[lane], map_pow_env, act);
    if (live) flags |= step_flags;

    // ================================================================ the strides
    bool active = live;
    int k_done = 0, last_rd = 0, status = kOk;        // strides this lane ran; the exchange parity and the status of its last one
    const StrideK K = stride_k_vreg();                // (a role keeps the members its right-hand sides read; see d_vreg)
#pragma unroll 1
    for (int k = 0; k < substeps; ++k) {
      if (__ballot(active) == 0ull) break;            // (the same decision in all four waves: `active` derives from shared words)
      const int wr = (xk + 1) & 1;
      // who publishes: a running episode, and the shadow of one that was over on entry (nobody reads ITS finals, and its own
      // reads must find finite numbers); a lane that ended in this step keeps its slots -- they hold its final state
      const bool publish = active || !live;
      // every right-hand side reads the state of stride k (balloon.py:322-325): new values into *_n, committed after the barrier
      double p_n = p, t_amb_n = t_amb, t_at_p_n = t_at_p, t_int_n = t_int, vol_n = vol, sp_n = sp, n_air_n = n_air;
      float x_n = x, y_n = y, batt_n = batt, sun_sin_n = sun_sin, sun_panel_n = sun_panel; bool sun_day_n = sun_day;
      const double rp = d_rcp(p);
      if (r0) {
        const double yc = inv_cbrt_volume(vol);
        p_n = stride_pressure(win, lc, p, rp, vol, n_air, t_amb, t_at_p, yc, K);
        x_n = f_fma(u, kStride, x); y_n = f_fma(v, kStride, y);        // step 1 (balloon.py:394-395)
        t_amb_n = t_at_p;                                               // ambient_temperature' = T(p_old)  (balloon.py:457)
        t_at_p_n = stride_ambient_advance(win, lc, p, rp, t_at_p, p_n, K);
        if (publish) { sh.p[wr][lane] = p_n; sh.t_amb[wr][lane] = t_amb_n; sh.x[wr][lane] = x_n; sh.y[wr][lane] = y_n; }
      }
      if (r1) {
        const float flux = f_fma((float)k, dfl, fl0);
        const double yc = inv_cbrt_volume(vol);
        const float att = solar_attenuation(sun_sin, (float)p, sun_day);
        t_int_n = stride_internal_temperature(vol, yc, t_int, t_amb, p, flux, att, hc.q_earth, K);
        if (publish) sh.t_int[wr][lane] = t_int_n;
      }
      if (rs) {
        const SunState sn = sun_at_stride(k + 1, sq, c, u, v, x_start, y_start, t_start);
        sun_sin_n = sn.sin_el; sun_panel_n = solar_panel_factor(sn); sun_day_n = sn.day;
        if (publish) { sh.sin_el[wr][lane] = sun_sin_n; sh.panel[wr][lane] = sun_panel_n; sh.day[wr][lane] = sun_day_n ? 1u : 0u; }
      }
      if (r2) {
        // step 4: superpressure and volume (balloon.py:470-482): burst above 2 380 Pa, zero pressure at <= 0 (the later check overrides)
        superpressure_volume_f64(n_air, t_int, p, rp, &vol_n, &sp_n, K);
        const uint32_t code = sp_n <= 0.0 ? (uint32_t)kZeroPressure : (!(sp_n <= 2380.0) ? (uint32_t)kBurst : 0u);
        if (publish) { sh.vol[wr][lane] = vol_n; sh.sp[wr][lane] = sp_n; sh.code_sp[wr][lane] = code; }
      }
      if (r3) {
        const float att = solar_attenuation(sun_sin, (float)p, sun_day);
        double mdot_d;
        stride_acs(sh.acs_poly, eff, sp, p, rp, t_int, &acs_w, &mdot_d, K);
        mdot = (float)mdot_d;
        n_air_n = stride_mols_air(n_air, mdot_d);
        stride_power_from_factor(sun_day, sun_panel, att, acs_w, &charge, &load, &batt_n);
        if (publish) {
          sh.n_air[wr][lane] = n_air_n; sh.batt[wr][lane] = batt_n; sh.acs_w[wr][lane] = acs_w; sh.mdot[wr][lane] = mdot;
          sh.charge[wr][lane] = charge; sh.load[wr][lane] = load;
          sh.code_batt[wr][lane] = batt_n <= 0.0f ? (uint32_t)kOutOfPower : 0u;         // balloon.py:541-542
        }
      }
      BLE_SPLIT_T(2);
      __syncthreads();
      BLE_SPLIT_T(3);
      ++xk;
      const int rd = xk & 1;
      // commit: what this wave advanced from its registers, what it needs of the others from LDS
      if (r0) { p = p_n; t_amb = t_amb_n; t_at_p = t_at_p_n; x = x_n; y = y_n; } else if (r1 || r2 || r3) { p = sh.p[rd][lane]; }
      if (r1 && !r0) t_amb = sh.t_amb[rd][lane];
      if (r1) t_int = t_int_n; else if (r2 || r3) t_int = sh.t_int[rd][lane];
      if (r2) { vol = vol_n; sp = sp_n; } else { if (r0 || r1) vol = sh.vol[rd][lane]; if (r3) sp = sh.sp[rd][lane]; }
      if (r3) { n_air = n_air_n; batt = batt_n; } else if (r0 || r2) n_air = sh.n_air[rd][lane];
      if (rs) { sun_sin = sun_sin_n; sun_panel = sun_panel_n; sun_day = sun_day_n; }
      else if (r1 || r3) { sun_sin = sh.sin_el[rd][lane]; sun_day = sh.day[rd][lane] != 0u; if (r3) sun_panel = sh.panel[rd][lane]; }
      // later checks override earlier ones (balloon.py:479-482, 541-542): burst, zero pressure, out of power
      const uint32_t cb = sh.code_batt[rd][lane], cs = sh.code_sp[rd][lane];
      const int code = (int)(cb != 0u ? cb : cs);
      // (selects, not a branch: a rarely taken block in this loop measured +3 %)
      k_done = active ? k + 1 : k_done; last_rd = active ? rd : last_rd; status = active ? code : status;
      active = active && code == kOk;                 // balloon.py:327-328
      BLE_SPLIT_T(4);
    }

    // ================================================================ end of the step: float32 state, status, reward
    if (live) {
      const int rd = last_rd;         // this lane's last stride (it may have ended before the others)
      s.p = (float)sh.p[rd][lane]; s.t_amb = (float)sh.t_amb[rd][lane]; s.t_int = (float)sh.t_int[rd][lane];
      s.vol = (float)sh.vol[rd][lane]; s.sp = (float)sh.sp[rd][lane]; s.n_air = (float)sh.n_air[rd][lane];
      s.batt = sh.batt[rd][lane]; s.x = sh.x[rd][lane]; s.y = sh.y[rd][lane];
      s.t_elapsed += 10 * k_done;
      s.status = (uint8_t)status;
      if (r1) flags |= (s.t_int < 12.3f) ? kFlagAbsorptivity : 0u;
      if (r0) {       // (the vertical wave: its per-step part is the lightest)
        s.acs_power = sh.acs_w[rd][lane];
        // solar_atmospheric_attenuation's range check (solar.py:194-197); p moves < 3 kPa per step
        flags |= (s.p > 101325.0f || s.p < 0.0f || p0_in > 101325.0f || p0_in < 0.0f) ? kFlagSolarRange : 0u;
        // the sun at the end of the step = what the sun wave published at the lane's last stride: sun_at_stride(k_done); the reward
        // needs sin el and day (attenuation) and the panel factor (solar_power == solar_power_from_factor o solar_panel_factor)
        SunState sun_end = {};
        sun_end.sin_el = sh.sin_el[rd][lane]; sun_end.day = sh.day[rd][lane] != 0u;
        const float panel_end = sh.panel[rd][lane];
        float r = reward_distance(s.x, s.y);
        if (act == kDown) {   // last_command is the RAW action (balloon.py:286); step_reward() with the panel factor at hand
          const float pw = solar_power_from_factor(panel_end, solar_attenuation(sun_end.sin_el, s.p, sun_end.day));
          const bool excess = (pw > kDayLoad) && battery_above_99_percent(s.batt);   // balloon.py:231-238
          if (!excess) {
            const float scale = f_clamp((s.acs_power - 100.0f) * (1.0f / 200.0f), 0.0f, 1.0f);
            r *= f_fma(-0.3f, scale, 0.95f);
          }
        }
        if (!(isfinite(s.p) && isfinite(s.t_int) && isfinite(s.x) && isfinite(s.y) && isfinite(s.batt))) flags |= kFlagNonFinite;
        a.reward[o] = r;
        a.terminal[o] = s.status != kOk;
      }
      if (r3) {
        s.acs_power = sh.acs_w[rd][lane]; s.mdot = sh.mdot[rd][lane]; s.charge = sh.charge[rd][lane]; s.load = sh.load[rd][lane];
        if (a.effective_action) a.effective_action[o] = (uint8_t)eff;
      }
    } else if (in_range) {  // balloon.py:288-290 raises; a vectorised env freezes the lane instead
      if (r0) { a.reward[o] = 0.0f; a.terminal[o] = 1; }
      if (r3 && a.effective_action) a.effective_action[o] = a.action[o];
    }
#ifndef BLE_SPLIT_TIMING
    if (r0 && a.active_count) {
      const unsigned long long m = __ballot(live);
      if (lane == 0 && m)
        atomicAdd(a.active_count + (int64_t)step * BLE_COUNT_SLOTS + (blockIdx.x & (BLE_COUNT_SLOTS - 1)), (unsigned long long)__popcll(m));
    }
#endif
    live = live && s.status == kOk;
    BLE_SPLIT_T(5);
  }

  if (was_live) {
    if (r0) {
      st.x[i] = s.x; st.y[i] = s.y; st.pressure[i] = s.p; st.ambient_temperature[i] = s.t_amb;
      st.time_elapsed_s[i] = s.t_elapsed; st.alt_fsm[i] = s.alt_fsm;
    }
    if (r1) { st.internal_temperature[i] = s.t_int; st.status[i] = s.status; st.last_command[i] = (uint8_t)last_act; }
    if (r2) { st.battery_charge[i] = s.batt; st.envelope_volume[i] = s.vol; st.superpressure[i] = s.sp; st.mols_air[i] = s.n_air; }
    if (r3) {
      st.acs_power[i] = s.acs_power; st.acs_mass_flow[i] = s.mdot; st.solar_charging[i] = s.charge; st.power_load[i] = s.load;
      st.sunrise_h_rel[i] = s.sunrise_h; st.sunset_rel[i] = s.sunset; st.env_fsm[i] = s.env_fsm; st.power_paused[i] = s.paused;
    }
  }
  BLE_SPLIT_T_FLUSH(a.active_count, wave);
  return flags;
}

}  // namespace ble
