// ble_kernels.hip -- gfx950 (MI355X, CDNA4) kernels and the C ABI of libble_hip.so.
//
// Execution model: one wavefront lane per environment, 256-thread workgroups of four independent waves,
// so N = 65 536 environments is 256 workgroups = one per CU, one wave on every SIMD of the 256 CUs.
// The state is struct-of-arrays: every load/store below is a fully coalesced
// 64-lane x 4 B (or 1 B) transaction.  The 317 KB wind grid is shared by all lanes and is
// served from the per-XCD L2 after first touch; each lane gathers its 16 corners as
// 8 x (4 contiguous floats).  Nothing here is a dense contraction: no MFMA.
// Target: gfx950 only (hipcc --offload-arch=gfx950); no other backend, no shims.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>
#include <atomic>

// Instrumentation hooks of ble_step_kernel: empty in the product build.  A profiling build
// (profiles/build_variant.sh ... -DBLE_STEP_BLOCK=64 -DBLE_STEP_INSTR_HEADER='"../../profiles/instr/ble_step_instr.h"') takes per-wave clock
// marks from that header, which is not part of the package.
#ifdef BLE_STEP_INSTR_HEADER
#include BLE_STEP_INSTR_HEADER
#else
#define BLE_STEP_INSTR_BEGIN() do {} while (0)
#define BLE_STEP_MARK(i) do {} while (0)
#define BLE_STEP_INSTR_END() do {} while (0)
#define BLE_STEP_STEP_DONE(k) do {} while (0)
#define BLE_STEP_COUNTS_LIVE 1
#endif

#include "../../include/ble_abi.h"
#include "ble_reset.h"
#include "ble_step_core.h"
#include "ble_noise.h"
#include "ble_step_split.h"
#include "ble_observe.h"
#include "ble_decode.h"

using namespace ble;

namespace {

constexpr int kBlock = 64;  // one wavefront per workgroup
// ble_step_kernel's workgroup: BLE_STEP_BLOCK / 64 independent wavefronts (they share the ACS table's LDS copy and one barrier at entry).
// 256 = one workgroup per CU at 65 536 environments, a wave on each of its SIMDs: a quarter of the dispatches of 64-thread workgroups --
// measured 15.5-15.7 against 15.7-15.9 us per fused step and 23.6-23.8 against 24.2 us per one-step launch (profiles/r04_raw/step_block_ab.txt)
#ifndef BLE_STEP_BLOCK
#define BLE_STEP_BLOCK 256
#endif
constexpr int kStepBlock = BLE_STEP_BLOCK;

__device__ __forceinline__ void report_flags(uint32_t flags, uint32_t* err_flags) {
  // wave-level OR, one atomic per wave at most (normally none)
  if (err_flags == nullptr) return;
  if (__any(flags != 0)) {
    for (int off = 32; off > 0; off >>= 1) flags |= __shfl_xor(flags, off, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(err_flags, flags);
  }
}

// `n_steps` consecutive agent steps of the rank's environments in ONE launch: the state is
// loaded once, stays in registers across the steps and is stored once; per step only the
// action byte is read, the 16 wind-grid corners are gathered and reward / terminal are
// written.  n_steps == 1 is the plain BalloonArena.step; n_steps > 1 serves ble_step_n_f32
// (rollouts whose actions are known up front, e.g. the random policy of the headline config).
// action / reward / terminal are [n_steps][n]; active_count is [n_steps][BLE_COUNT_SLOTS].
// kNoise (ble_step_n_f32 with a noise generator, ABI 3): the SimplexWindNoise term of WindField.get_ground_truth
// (wind_field.py:125-145) is evaluated IN the kernel at every step's pre-step position -- the same lane function as
// ble_wind_noise_f32 (wind_noise_cached), hence the same bits as ble_wind_noise_f32 + ble_step_f32 step by step.  A
// separate instantiation: the noise-free rollout keeps its register allocation.
// ble_step_kernel<noise>'s LDS as ONE object in this order: the gradient table (read five times per harmonic at a per-lane index) and the
// arrays the stride loop reads stay within ds_read's 16-bit offset; the 50 KB of harmonic draws, walked by a pointer, come last.  As
// separate objects the compiler put the draws first and the table at the end of 66 KB: every read of it paid an addition of its base.
struct StepNoiseShared {
  __attribute__((aligned(16))) float grad_lut[kGradLutFloats];      // the noise primitive's gradient weights
  double acs_poly[kAcsPolyDoubles];
  float term_save[kTermSaveRows * kStepBlock];
  uint32_t draws[50 * kStepBlock];      // the harmonics' seeds and offsets of the workgroup's environments, fetched once per launch
};
// V: the flight vehicle's constants -- VehicleDefault (compile-time: ble_state_f32.vehicle == NULL) or VehicleRt (a kernel argument, i.e.
// scalar registers: ABI 5) -- as the LAST argument, so that the default instantiation's argument layout is what it was.
template <bool kNoise, class V = VehicleDefault>
__global__ __launch_bounds__(kStepBlock) void ble_step_kernel(StateDev st, const uint8_t* __restrict__ action,
                                                          const float* __restrict__ wind_grid,
                                                          int64_t grid_env_stride,
                                                          const float* __restrict__ noise_uv,
                                                          float* __restrict__ reward,
                                                          uint8_t* __restrict__ terminal,
                                                          uint8_t* __restrict__ effective_action,
                                                          uint32_t* err_flags, unsigned long long* active_count,
                                                          int64_t n, int substeps, int lanes, int n_steps, StepNoise gen, V veh) {
  // `lanes` (64 or 32) = environments per wavefront.  32 leaves the upper half of the wave
  // idle and doubles the number of waves: an occupancy/latency experiment knob.
  // acs_poly: the ACS table's piecewise cubics; term_save: where a lane parks the state its episode ended with (agent_step), one block
  // per wave
  double* acs_poly; float* term_save; float* grad_lut = nullptr; uint32_t* noise_draws = nullptr;
  if constexpr (kNoise) {
    __shared__ StepNoiseShared shm;
    acs_poly = shm.acs_poly; term_save = shm.term_save; grad_lut = shm.grad_lut; noise_draws = shm.draws;
  } else {
    __shared__ double acs_poly_lds[kAcsPolyDoubles];
    __shared__ float term_save_lds[kTermSaveRows * kStepBlock];
    acs_poly = acs_poly_lds; term_save = term_save_lds;
  }
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * (kStepBlock / 64) + wave) * lanes + lane;
  const bool in_range = i < n && lane < lanes;
  uint32_t flags = 0;
  EnvRegs s;
  EnvConst c;
  EpisodeCacheRow cached = {};
  bool live = false;
  BLE_STEP_INSTR_BEGIN();
  if (in_range) {
    // every load is issued up front, unconditionally (one memory round trip)
    s.status = st.status[i];
    s.x = st.x[i]; s.y = st.y[i]; s.p = st.pressure[i]; s.t_amb = st.ambient_temperature[i];
    s.t_int = st.internal_temperature[i]; s.vol = st.envelope_volume[i]; s.sp = st.superpressure[i];
    s.n_air = st.mols_air[i]; s.batt = st.battery_charge[i];
    s.acs_power = 0.0f; s.mdot = 0.0f; s.charge = 0.0f; s.load = 0.0f;
    s.t_elapsed = st.time_elapsed_s[i]; s.sunrise_h = st.sunrise_h_rel[i]; s.sunset = st.sunset_rel[i];
    s.alt_fsm = st.alt_fsm[i]; s.env_fsm = st.env_fsm[i]; s.paused = st.power_paused[i];
    c.lat0_deg = st.center_lat_deg[i]; c.lng0_deg = st.center_lng_deg[i];
    c.ir = st.upwelling_infrared[i]; c.alpha = st.alpha[i]; c.start_unix = st.start_unix[i];
    if (st.episode_cache != nullptr) cached = episode_cache_load(st.episode_cache, n, i);
    live = s.status == kOk;
  }
  // the ACS table's piecewise cubics: a compile-time table, constant memory -> LDS (the loop reads it by a per-lane index)
  for (int j = (int)threadIdx.x; j < kAcsPolyDoubles; j += kStepBlock) acs_poly[j] = kAcsPoly.c[j];
  if (kNoise) grad_lut_fill(grad_lut, (int)threadIdx.x, kStepBlock);
  BLE_STEP_MARK(1);
  __syncthreads();
  BLE_STEP_MARK(2);
  const bool was_live = live;
  int last_act = 0;
  EnvHoisted hc;
  if (live) {
    // per-episode constants: from the cache unless its entry belongs to other constants (then: recompute, store)
    if (st.episode_cache != nullptr && episode_cache_hit(cached, c)) {
      hc = hoisted_from_cache(cached, c);
    } else {
      hc = hoist_constants(c);
      if (st.episode_cache != nullptr) episode_cache_store(st.episode_cache, n, i, c, hc);
    }
  }
  if (kNoise && in_range)
    noise_draws_fetch(gen.seed, (uint64_t)i, (uint64_t)(i + gen.env_offset), gen.episode ? gen.episode[i] : 0u, gen.harmonic_cache, n,
                      noise_draws + threadIdx.x, kStepBlock);
  const StrideK K = stride_k_vreg(veh.dry_mass, veh.lift, veh.v0);      // the stride loop's fp64 constants as register pairs, once per launch (see d_vreg)
  BLE_STEP_MARK(3);
#pragma unroll 1
  for (int k = 0; k < n_steps; ++k) {
    const int64_t o = (int64_t)k * n + i;
    if (live) {
      const int act = action[o];
      last_act = act;
      // wind at the PRE-step position/time (balloon_arena.py:194,270-275): gather now, blend later
      const WindQuery wq = wind_query(s.x, s.y, s.p, s.t_elapsed);
      WindCorners corners;
      wind_gather(wind_grid + i * grid_env_stride, wq, &corners);
      float nu = 0.0f, nv = 0.0f;
      if (kNoise) {
        wind_noise_from_rows(s.x, s.y, s.p, s.t_elapsed, noise_draws + threadIdx.x, kStepBlock, grad_lut, &nu, &nv);
        // the noise is a VALUE here as it is between ble_wind_noise_f32 and ble_step_f32: without this the compiler is free to
        // fuse the generator's last multiplication into agent_step's `u += noise_u` (one rounding instead of two)
        asm volatile("" : "+v"(nu), "+v"(nv));
      } else if (noise_uv) { nu = noise_uv[2 * i]; nv = noise_uv[2 * i + 1]; }
      float r;
      const int eff = agent_step(s, c, hc, act, corners, wq, nu, nv, substeps, acs_poly, K, term_save + wave * (kTermSaveRows * kTermSaveStride) + lane, &r, &flags, veh);
      if (!(isfinite(s.p) && isfinite(s.t_int) && isfinite(s.x) && isfinite(s.y) && isfinite(s.batt)))
        flags |= kFlagNonFinite;
      reward[o] = r;
      terminal[o] = s.status != kOk;
      if (effective_action) effective_action[o] = (uint8_t)eff;
    } else if (in_range) {  // balloon.py:288-290 raises; a vectorised env freezes the lane instead
      reward[o] = 0.0f;
      terminal[o] = 1;
      if (effective_action) effective_action[o] = action[o];
    }
    // live-environment count: one atomic per wave, spread over BLE_COUNT_SLOTS addresses and
    // issued after the step so that no load of this wave queues behind it
    if (BLE_STEP_COUNTS_LIVE && active_count) {
      const unsigned long long m = __ballot(live);
      if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(active_count + (int64_t)k * BLE_COUNT_SLOTS + (blockIdx.x & (BLE_COUNT_SLOTS - 1)),
                  (unsigned long long)__popcll(m));
    }
    live = live && s.status == kOk;
    BLE_STEP_STEP_DONE(k);
  }
  BLE_STEP_MARK(4);
  if (was_live) {
    st.x[i] = s.x; st.y[i] = s.y; st.pressure[i] = s.p; st.ambient_temperature[i] = s.t_amb;
    st.internal_temperature[i] = s.t_int; st.envelope_volume[i] = s.vol; st.superpressure[i] = s.sp;
    st.mols_air[i] = s.n_air; st.battery_charge[i] = s.batt;
    st.acs_power[i] = s.acs_power; st.acs_mass_flow[i] = s.mdot; st.solar_charging[i] = s.charge;
    st.power_load[i] = s.load;
    st.time_elapsed_s[i] = s.t_elapsed; st.sunrise_h_rel[i] = s.sunrise_h; st.sunset_rel[i] = s.sunset;
    st.status[i] = s.status; st.last_command[i] = (uint8_t)last_act;
    st.alt_fsm[i] = s.alt_fsm; st.env_fsm[i] = s.env_fsm; st.power_paused[i] = s.paused;
  }
  BLE_STEP_INSTR_END();
  report_flags(flags, err_flags);
}

// The same transition for small batches: one environment on the four wavefronts of a 256-thread workgroup (ble_step_split.h).
#ifndef BLE_SPLIT_WAVES_PER_EU
#define BLE_SPLIT_WAVES_PER_EU 2
#endif
template <bool kNoise>
__global__ __launch_bounds__(kSplitWaves * kSplitLanes, BLE_SPLIT_WAVES_PER_EU) void ble_step_split_kernel(SplitArgs a) {
  __shared__ SplitShared sh;
  __shared__ SplitNoiseShared<kNoise> shn;
  uint32_t flags;
  switch (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)) {       // (scalar: one role per wave)
    case 0: flags = split_agent_steps<4, 0, kNoise>(a, sh, shn); break;
    case 1: flags = split_agent_steps<4, 1, kNoise>(a, sh, shn); break;
    case 2: flags = split_agent_steps<4, 2, kNoise>(a, sh, shn); break;
    default: flags = split_agent_steps<4, 3, kNoise>(a, sh, shn); break;
  }
  report_flags(flags, a.err_flags);
}
// ... and on two: {vertical, thermal} | {sun + envelope, ACS + power}, 128-thread workgroups (two waves per SIMD at 65 536 environments).
// An A/B form that the automatic choice never took and that measured slower at every batch size (profiles/HISTORY.md): since round 6 it is
// NOT part of the product library -- profiles/build_variant.sh -DBLE_WITH_PAIR_FORM builds it for experiments (ble_set_step_form(2)).
#ifdef BLE_WITH_PAIR_FORM
template <bool kNoise>
__global__ __launch_bounds__(2 * kSplitLanes) void ble_step_pair_kernel(SplitArgs a) {
  __shared__ SplitShared sh;
  __shared__ SplitNoiseShared<kNoise> shn;
  uint32_t flags;
  if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0) flags = split_agent_steps<2, 0, kNoise>(a, sh, shn);
  else flags = split_agent_steps<2, 1, kNoise>(a, sh, shn);
  report_flags(flags, a.err_flags);
}
constexpr bool kHavePairForm = true;
#else
constexpr bool kHavePairForm = false;
#endif

__global__ __launch_bounds__(256) void ble_forecast_kernel(const float* __restrict__ wind_grid,
                                                           int64_t grid_env_stride, const float* __restrict__ x,
                                                           const float* __restrict__ y,
                                                           const float* __restrict__ pressure,
                                                           const int32_t* __restrict__ elapsed, float* __restrict__ u,
                                                           float* __restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // float32 query, fp64 interpolation like scipy's interpn; the float32 result is the correctly rounded reference value
  // (the transition's own, fused lookup blends in fp32: wind_blend_corners)
  double uu, vv;
  wind_forecast_f64(wind_grid + i * grid_env_stride, x[i], y[i], pressure[i], elapsed[i], &uu, &vv);
  u[i] = (float)uu; v[i] = (float)vv;
}

// get_forecast_column (grid_based_wind_field.py:96-132): one wave per column.  The wave
// first collapses the (x, y, t) axes: lanes 0..19 each own one (pressure node, component)
// and blend its 8 (x, y, t) corners -- the "local pressure column" -- into LDS; then every
// lane interpolates its pressure levels from the 10-node column held in LDS.
__global__ __launch_bounds__(kBlock) void ble_forecast_column_kernel(
    const float* __restrict__ wind_grid, int64_t grid_env_stride, const float* __restrict__ x,
    const float* __restrict__ y, const int32_t* __restrict__ elapsed, const float* __restrict__ levels,
    int n_levels, float* __restrict__ out_uv, int64_t n) {
  __shared__ double column[BLE_GRID_NP * 2];
  const int64_t env = blockIdx.x;
  if (env >= n) return;
  const int lane = threadIdx.x;
  const WindQueryD wq = wind_query_xyt_f64(x[env], y[env], elapsed[env]);
  const float* grid = wind_grid + env * grid_env_stride;
  if (lane < BLE_GRID_NP * 2) {
    const int ip = lane >> 1, comp = lane & 1;
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const double w = ((a ? wq.wx : 1.0 - wq.wx) * (b ? wq.wy : 1.0 - wq.wy)) * (d ? wq.wt : 1.0 - wq.wt);
          acc = d_fma((double)grid[((((wq.ix + a) * 21 + (wq.iy + b)) * 10 + ip) * 9 + (wq.it + d)) * 2 + comp], w, acc);
        }
    column[lane] = acc;
  }
  __syncthreads();
  for (int l = lane; l < n_levels; l += kBlock) {
    const float p = f_clamp(levels[l], 5000.0f, 14000.0f);
    int ip = (int)((p - 5000.0f) * (1.0f / 1000.0f));
    ip = ip > 8 ? 8 : ip;
    const double wp = ((double)p - (5000.0 + 1000.0 * (double)ip)) * 1e-3;
    const double u = d_fma(wp, column[(ip + 1) * 2] - column[ip * 2], column[ip * 2]);
    const double v = d_fma(wp, column[(ip + 1) * 2 + 1] - column[ip * 2 + 1], column[ip * 2 + 1]);
    out_uv[(env * n_levels + l) * 2] = (float)u;
    out_uv[(env * n_levels + l) * 2 + 1] = (float)v;
  }
}

// ble_state_rows_f64: struct of arrays -> records of BLE_ROW_DOUBLES doubles, the struct's member order
__global__ __launch_bounds__(64) void ble_state_rows_kernel(StateDev st, int64_t first, int64_t count, double* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= count) return;
  const int64_t i = first + r;
  double* o = out + r * BLE_ROW_DOUBLES;
  o[0] = st.x[i]; o[1] = st.y[i]; o[2] = st.pressure[i]; o[3] = st.ambient_temperature[i]; o[4] = st.internal_temperature[i];
  o[5] = st.envelope_volume[i]; o[6] = st.superpressure[i]; o[7] = st.mols_air[i]; o[8] = st.battery_charge[i];
  o[9] = st.acs_power[i]; o[10] = st.acs_mass_flow[i]; o[11] = st.solar_charging[i]; o[12] = st.power_load[i];
  o[13] = st.center_lat_deg[i]; o[14] = st.center_lng_deg[i]; o[15] = st.upwelling_infrared[i]; o[16] = st.alpha[i];
  o[17] = (double)st.start_unix[i]; o[18] = (double)st.time_elapsed_s[i]; o[19] = (double)st.sunrise_h_rel[i]; o[20] = (double)st.sunset_rel[i];
  o[21] = (double)st.status[i]; o[22] = (double)st.last_command[i]; o[23] = (double)st.alt_fsm[i]; o[24] = (double)st.env_fsm[i];
  o[25] = (double)st.power_paused[i];
}

__global__ __launch_bounds__(256) void ble_power_table_kernel(const float* __restrict__ pr,
                                                              const float* __restrict__ soc, float* __restrict__ watts,
                                                              uint32_t* err_flags, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t flags = 0;
  if (i < n) watts[i] = power_table_lookup(pr[i], soc[i], &flags);
  report_flags(flags, err_flags);
}

// ---- probes: the same lane functions, one element per lane ----
__global__ __launch_bounds__(256) void probe_atmosphere_kernel(const float* alpha, const float* pressure, float* height,
                                                               float* temperature, uint32_t* err_flags, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t flags = 0;
  if (i < n) {
    const double p = (double)pressure[i];
    const AtmWindow w = atm_window((double)alpha[i], p, &flags);
    double h, t;
    atm_at_pressure_f64(w, (double)alpha[i], p, &h, &t);
    height[i] = (float)h; temperature[i] = (float)t;
  }
  report_flags(flags, err_flags);
}
__global__ __launch_bounds__(256) void probe_at_height_kernel(const float* alpha, const double* height, double* pressure, double* temperature,
                                                              uint32_t* err_flags, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t flags = 0;
  if (i < n) atm_at_height_f64((double)alpha[i], height[i], &pressure[i], &temperature[i], &flags);
  report_flags(flags, err_flags);
}
__global__ __launch_bounds__(256) void probe_solar_kernel(const float* lat0, const float* lng0, const float* x,
                                                          const float* y, const int64_t* unix_s, float* el_deg,
                                                          float* flux, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Ephemeris e = ephemeris(unix_s[i]);
  int64_t sod = unix_s[i] % 86400;
  if (sod < 0) sod += 86400;
  const double b = (double)sod * (1.0 / 240.0) + 0.25 * e.eot_min + (double)lng0[i];
  double sl, cl;
  sincos_f64((double)lat0[i] * (kPiD / 180.0), &sl, &cl);
  double sb, cb;
  sincos_f64(b * (kPiD / 180.0), &sb, &cb);
  const double oms = sun_one_minus_sin_f64(sl, cl, (double)x[i], (double)y[i], sb, cb, (double)e.sin_decl, (double)e.cos_decl);
  const SunSC sun = sun_refract(sun_from_one_minus_sin((float)oms));
  el_deg[i] = atan2f(sun.sin_el, sun.cos_el) * kRadToDeg;
  flux[i] = e.flux;
}
// BalloonState.latlng (balloon.py:217-220 -> spherical_geometry.py:44-76): the latlng_f64 the observation and the exact solar
// chain evaluate, as degrees
__global__ __launch_bounds__(256) void probe_latlng_kernel(const float* lat0, const float* lng0, const float* x, const float* y,
                                                           double* lat_deg, double* lng_deg, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double sl, cl, lng;
  latlng_f64((double)lat0[i], (double)lng0[i], (double)x[i], (double)y[i], &sl, &cl, &lng);
  lat_deg[i] = asin(sl) * (180.0 / kPiD);          // (libm: the probe reports degrees to 1e-12; the kernels use sin / cos directly)
  lng = lng - 360.0 * floor((lng + 180.0) * (1.0 / 360.0));             // s2 LatLng.normalized(): [-180, 180)
  lng_deg[i] = lng;
}
__global__ __launch_bounds__(256) void probe_solar_power_kernel(const float* el_deg, const float* pressure, float* att,
                                                                float* power, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s, c;
  sincos_f64((double)el_deg[i] * (kPiD / 180.0), &s, &c);
  // thresholds decided on the fp64 input elevation, as the transition's exact path does
  const double el = (double)el_deg[i];
  SunState sun;
  sun.sin_el = (float)s; sun.cos_el = (float)c;
  sun.day = !(el < -4.242); sun.sh33 = el >= 37.738149050524044; sun.sh27 = el >= 34.39486500086289;
  const float a = solar_attenuation(sun.sin_el, pressure[i], sun.day);
  att[i] = a;
  power[i] = solar_power(sun, a);
}
__global__ __launch_bounds__(256) void probe_thermal_kernel(const float* volume, const float* t_int, const float* t_amb,
                                                            const float* pressure, const float* el_deg,
                                                            const float* flux, const float* ir, float* dtdt,
                                                            uint32_t* err_flags, int64_t n, double thermal_scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t flags = 0;
  if (i < n) {
    double s, c;
    sincos_f64((double)el_deg[i] * (kPiD / 180.0), &s, &c);
    const float att = solar_attenuation((float)s, pressure[i], !((double)el_deg[i] < -4.242));
    const double vol = (double)volume[i];
    double yc = (double)f_exp2((-1.0f / 3.0f) * f_log2(volume[i]));
    yc = yc * d_fma(-vol * yc, yc * yc, 4.0) * (1.0 / 3.0);
    flags |= (t_int[i] < 12.3f) ? kFlagAbsorptivity : 0u;
    dtdt[i] = (float)(0.1 * thermal_increment_f64(vol, yc, (double)t_int[i], (double)t_amb[i], (double)pressure[i],
                                                  (double)((flux[i] * att) * (0.25f * kSolarAbsorptivityTotal)),
                                                  earth_heat_per_area_f64((double)ir[i], &flags), stride_k_literal(), thermal_scale));
  }
  report_flags(flags, err_flags);
}
__global__ __launch_bounds__(256) void probe_sp_volume_kernel(const float* mols_air, const float* t_int,
                                                              const float* pressure, float* volume, float* sp,
                                                              int64_t n, double lift, double v0, double dvdp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v, s;
  superpressure_volume_f64((double)mols_air[i], (double)t_int[i], (double)pressure[i], d_rcp((double)pressure[i]), &v, &s,
                           stride_k_literal(VehicleDefault::dry_mass, lift, v0), dvdp, 4.0 * dvdp, 1.0 / dvdp);
  volume[i] = (float)v; sp[i] = (float)s;
}
__global__ __launch_bounds__(256) void probe_acs_kernel(const float* pr, float* power, float* eff, float* mdot,
                                                        int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // power and mass flow through the transition's piecewise-cubic form, the efficiency through the two-table form
  const double prm1 = (double)pr[i] - 1.0;
  double w, md;
  acs_down_poly(kAcsPoly.c, prm1, &w, &md);       // the compile-time table the transition copies into LDS
  power[i] = (float)w; eff[i] = (float)acs_efficiency_f64(kAcsEfficiency, prm1, acs_power_f64(prm1)); mdot[i] = (float)md;
}

// The three safety layers one at a time (ble_probe_safety_f32): the lane functions of agent_step on their own state bytes.
__global__ __launch_bounds__(256) void probe_safety_kernel(int layer, const uint8_t* action, const float* value,
                                                           const float* alpha, int32_t* clocks, double night_load_w,
                                                           double capacity_wh, uint8_t* fsm, uint8_t* effective,
                                                           uint32_t* err_flags, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t flags = 0;
  if (i < n) {
    uint8_t state = fsm[i];
    int eff;
    if (layer == 0) {                               // value = pressure: the altitude the transition compares (fp64)
      const double p = (double)value[i];
      const AtmWindow w = atm_window((double)alpha[i], p, &flags);
      double h, t;
      atm_at_pressure_f64(w, (double)alpha[i], p, &h, &t);
      eff = altitude_safety(action[i], h, &state);
    } else if (layer == 1) {                        // value = superpressure
      eff = envelope_safety(action[i], value[i], &state, alpha != nullptr ? (double)alpha[i] : VehicleDefault::max_sp);     // (ABI 5: `alpha` = the layer's maximum superpressure)
    } else {                                        // value = battery charge; clocks (now, sunrise + 1/2 h, sunset)
      int32_t sr = clocks[3 * i + 1], ss = clocks[3 * i + 2];
      eff = power_safety(action[i], clocks[3 * i], value[i], &sr, &ss, &state, night_load_w, capacity_wh);
      clocks[3 * i + 1] = sr; clocks[3 * i + 2] = ss;
    }
    fsm[i] = state; effective[i] = (uint8_t)eff;
  }
  report_flags(flags, err_flags);
}

// Decoder tail of the wind-field VAE (generative/vae.py:149-186): flow fields psi [n][7][7][90]
// (the last Dense layer's output, flow-field index fastest) -> half-pixel linear resize to
// 23 x 23 (jax.image.resize 'linear': triangle kernel, edge weights renormalised == clamped
// taps) -> central differences u = d psi / dy, v = -d psi / dx on the interior 21 x 21 ->
// the wind grid [n][21][21][10][9][2] the step kernel reads (grid_env_stride = 79 380).
// HBM-write-bound: 317.5 KB written per env against 17.6 KB read.  One workgroup = one environment, 4 groups of 90 threads
// (one per flow field; 24 idle): psi (17.6 KB) is staged in LDS, resized along the second axis once (P[7][23][90], 58 KB: the
// `lo` / `hi` of decode_resized, which depend on the source row alone), and every output row is then produced with the
// first-axis interpolation on the fly, the middle row's lattice points sliding through registers: 6 LDS reads per output and
// no integer division in the loops, against the 16 cached global loads + 4 divisions of the one-thread-per-output form (bound
// by its load issue rate at 1.6 TB/s of writes).  Stores: 720 contiguous bytes per 90 threads.  Same arithmetic, bit for bit.
#ifndef BLE_DECODE_GROUPS
#define BLE_DECODE_GROUPS 6          // (A/B knob of profiles/decode_ab.py: 4 .. 11 groups measured, 6 is the fastest)
#endif
constexpr int kDecodeGroups = BLE_DECODE_GROUPS, kDecodeThreads = (90 * kDecodeGroups + 63) / 64 * 64;
__global__ __launch_bounds__(kDecodeThreads) void ble_decode_flow_kernel(const float* __restrict__ flow, float* __restrict__ grid,
                                                                         int64_t n) {
  __shared__ float psi[7 * 7 * 90];
  __shared__ float part[7 * 23 * 90];        // psi resized along its second axis
  __shared__ int tap0[23];
  __shared__ float w1[23];
  const int64_t env = blockIdx.x;
  if (threadIdx.x < 23) resize_tap((int)threadIdx.x, &tap0[threadIdx.x], &w1[threadIdx.x]);
  for (int t = threadIdx.x; t < 7 * 7 * 90; t += kDecodeThreads) psi[t] = flow[env * (7 * 7 * 90) + t];
  __syncthreads();
  const int f = (int)threadIdx.x % 90, g = (int)threadIdx.x / 90;
  // the threads beyond the last group of 90 (kDecodeThreads rounds up to whole waves) run neither loop but DO reach the
  // barrier: their loops start at the end, there is no early return
  const int g_step = g < kDecodeGroups ? kDecodeGroups : 1;
  for (int rb = g < kDecodeGroups ? g : 7 * 23; rb < 7 * 23; rb += g_step) {
    const int r = rb / 23, b = rb - 23 * r;
    const int b0 = tap0[b], b_lo = b0 < 0 ? 0 : b0, b_hi = b0 + 1 > 6 ? 6 : b0 + 1;
    const float lo = psi[(r * 7 + b_lo) * 90 + f];
    part[rb * 90 + f] = f_fma(w1[b], psi[(r * 7 + b_hi) * 90 + f] - lo, lo);
  }
  __syncthreads();
  float2* out = reinterpret_cast<float2*>(grid + env * (int64_t)(21 * 21 * 90 * 2)) + f;
  for (int i = g < kDecodeGroups ? g : 21; i < 21; i += g_step) {
    // first-axis taps of the lattice rows i, i + 1, i + 2
    int lo_row[3], hi_row[3]; float wa[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int a0 = tap0[i + k];
      lo_row[k] = (a0 < 0 ? 0 : a0) * (23 * 90) + f; hi_row[k] = (a0 + 1 > 6 ? 6 : a0 + 1) * (23 * 90) + f; wa[k] = w1[i + k];
    }
    auto point = [&](int k, int b) {           // decode_resized's last line on the staged lo / hi
      const float lo = part[lo_row[k] + b * 90];
      return f_fma(wa[k], part[hi_row[k] + b * 90] - lo, lo);
    };
    float mid0 = point(1, 0), mid1 = point(1, 1);
    float2* row = out + (int64_t)i * (21 * 90);
#pragma unroll 3
    for (int j = 0; j < 21; ++j) {
      const float mid2 = point(1, j + 2);
      float u, v;
      decode_flow_from_lattice(point(2, j + 1), point(0, j + 1), mid2, mid0, &u, &v);
      row[j * 90] = make_float2(u, v);          // (non-temporal stores measured: 3.8 ms against 3.0 for 32 768 grids)
      mid0 = mid1; mid1 = mid2;
    }
  }
}

// mode 0: the wind noise (u, v) of every environment at its (x, y, pressure, elapsed);
// mode 1 (test probe): noise_uv[2 i] = simplex4(x, y, pressure, elapsed as float, seed) -- raw primitive.
__global__ __launch_bounds__(256) void ble_wind_noise_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ pressure,
                                                             const int32_t* __restrict__ elapsed, unsigned long long seed,
                                                             const uint32_t* __restrict__ episode, int mode,
                                                             uint32_t* harmonic_cache, float* __restrict__ noise_uv, int64_t n,
                                                             int64_t env_offset) {
  __shared__ __attribute__((aligned(16))) float grad_lut[kGradLutFloats];
  grad_lut_fill(grad_lut, (int)threadIdx.x, 256);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float u, v;
  if (mode == 0) {
    const uint32_t ep = episode ? episode[i] : 0u;
    if (harmonic_cache != nullptr)
      wind_noise_cached(x[i], y[i], pressure[i], elapsed[i], seed, (uint64_t)i, (uint64_t)(i + env_offset), ep, harmonic_cache, n, grad_lut, &u, &v);
    else
      wind_noise(x[i], y[i], pressure[i], elapsed[i], seed, (uint64_t)(i + env_offset), ep, grad_lut, &u, &v);
  } else {
    u = simplex4(x[i], y[i], pressure[i], (float)elapsed[i] * (1.0f / 3600.0f), (uint32_t)seed, grad_lut);
    v = 0.0f;
  }
  noise_uv[2 * i] = u; noise_uv[2 * i + 1] = v;
}

// Episode reset for the lanes selected by `mask` (all lanes if mask == nullptr).
// sample != 0: draw the initial conditions (utils/sampling.py, balloon_arena.py:228-268) from
// Philox(seed, env, episode[i]); sample == 0: keep x, y, pressure, centre lat/lng, IR, alpha,
// start_unix as they are.  Then the Newton cold start (stable_init.py:132-157), the sunrise /
// sunset search of PowerSafetyLayer.__init__ and fresh clocks / FSMs / battery (balloon.py:175-215).
template <class V = VehicleDefault>
__global__ __launch_bounds__(kBlock) void ble_reset_kernel(StateDev st, const uint8_t* __restrict__ mask,
                                                           unsigned long long seed, uint32_t* episode, int sample,
                                                           uint32_t* err_flags, int64_t n, int64_t env_offset, V veh) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  uint32_t flags = 0;
  if (i < n && (mask == nullptr || mask[i] != 0)) {
    float alpha, x, y, p, lat0, lng0, ir;
    int64_t start;
    if (sample) {
      const uint32_t ep = episode ? episode[i] : 0u;
      if (episode) episode[i] = ep + 1u;
      Philox g = philox_init(seed, (uint64_t)(i + env_offset), ep);        // keyed by the GLOBAL environment index
      alpha = (float)philox_uniform(g);                                                  // standard_atmosphere.py:82
      start = 1293840000LL + (int64_t)(philox_uniform(g) * (double)(1419984000LL - 1293840000LL));   // sampling.py:65-83
      const double ga = philox_gamma(g, 1.2), gb = philox_gamma(g, 2.0);                // Beta(1.2, 2.0)
      const double radius = 200000.0 * (ga / (ga + gb));                                // balloon_arena.py:153-154,246-247
      double sn, cs;
      sincos_f64(2.0 * kPiD * philox_uniform(g), &sn, &cs);
      x = (float)(cs * radius); y = (float)(sn * radius);
      lat0 = (float)(-10.0 + 20.0 * philox_uniform(g));                                 // sampling.py:37-62
      lng0 = (float)(-175.0 + 350.0 * philox_uniform(g));
      // pressure ~ U[6500, P(50 000 ft)]  (sampling.py:86-117; at_height standard_atmosphere.py:89-120, layer 0)
      const double l0 = atm_lapse_f64(0, (double)alpha);
      const double t_h = 300.0 + l0 * (15240.0 - -610.0);
      const double p_max = 108870.8213 * d_pow_fast(t_h / 300.0, -9.80665 / (kAirSpecificGasD * l0));
      p = (float)(6500.0 + (p_max - 6500.0) * philox_uniform(g));
      // upwelling IR: 315 * sigmoid(N(2, 315)), rejected below 225 (sampling.py:120-152, as written)
      double irs = 315.0;
#pragma unroll 1
      for (int it = 0; it < 64; ++it) {
        const double z = 2.0 + 315.0 * philox_normal(g);
        irs = z > 700.0 ? 315.0 : (z < -700.0 ? 0.0 : 315.0 / (1.0 + d_exp_fast(-z)));
        if (irs >= 225.0) break;
      }
      ir = (float)irs;
      const_cast<float*>(st.alpha)[i] = alpha; st.x[i] = x; st.y[i] = y; st.pressure[i] = p;
      const_cast<float*>(st.center_lat_deg)[i] = lat0; const_cast<float*>(st.center_lng_deg)[i] = lng0;
      const_cast<float*>(st.upwelling_infrared)[i] = ir; const_cast<int64_t*>(st.start_unix)[i] = start;
    } else {
      alpha = st.alpha[i]; x = st.x[i]; y = st.y[i]; p = st.pressure[i]; lat0 = st.center_lat_deg[i];
      lng0 = st.center_lng_deg[i]; ir = st.upwelling_infrared[i]; start = st.start_unix[i];
    }
    SunSite site;
    latlng_f64((double)lat0, (double)lng0, (double)x, (double)y, &site.sin_lat, &site.cos_lat, &site.lng_deg);
    double flux;
    const double el = solar_elevation_f64(site.sin_lat, site.cos_lat, site.lng_deg, start, &flux);
    const StableParams sp = stable_params((double)alpha, (double)p, el, flux, (double)ir, &flags, veh);
    int64_t sunrise, sunset;
    next_sunrise_sunset(site, start, &sunrise, &sunset);
    st.ambient_temperature[i] = (float)sp.t_amb; st.internal_temperature[i] = (float)sp.t_int;
    st.mols_air[i] = (float)sp.mols_air; st.envelope_volume[i] = (float)sp.volume; st.superpressure[i] = (float)sp.sp;
    st.battery_charge[i] = 2905.6f;                                                      // balloon.py:195
    st.acs_power[i] = 0.0f; st.acs_mass_flow[i] = 0.0f; st.solar_charging[i] = 0.0f; st.power_load[i] = 0.0f;
    st.time_elapsed_s[i] = 0;
    st.sunrise_h_rel[i] = (int32_t)(sunrise + 1800 - start);                             // power_safety.py:43-48
    st.sunset_rel[i] = (int32_t)(sunset - start);
    st.status[i] = kOk; st.last_command[i] = kStay; st.alt_fsm[i] = 0; st.env_fsm[i] = 0; st.power_paused[i] = 0;
    if (st.episode_cache != nullptr) {       // what the transition derives from this episode's constants alone
      EnvConst c;
      c.lat0_deg = lat0; c.lng0_deg = lng0; c.ir = ir; c.alpha = alpha; c.start_unix = start;
      episode_cache_store(st.episode_cache, n, i, c, hoist_constants(c));
    }
  }
  report_flags(flags, err_flags);
}

}  // namespace
// fp64 primitive probe (test-only entry point): op 0 rcp seed, 1 d_rcp, 2 rsq seed, 3 d_rsqrt,
// 4 d_sqrt_fast, 5 d_log_fast, 6 d_exp_fast, 7 sin (sincos_f64), 8 cos (sincos_f64)
__global__ __launch_bounds__(256) void probe_f64_kernel(const double* x, double* y, int op, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = 0.0, t = 0.0;
  switch (op) {
    case 0: r = d_rcp_seed(v); break;
    case 1: r = d_rcp(v); break;
    case 2: r = d_rsq_seed(v); break;
    case 3: r = d_rsqrt(v); break;
    case 4: r = d_sqrt_fast(v); break;
    case 5: r = d_log_fast(v); break;
    case 6: r = d_exp_fast(v); break;
    case 7: sincos_f64(v, &r, &t); break;
    default: sincos_f64(v, &t, &r); break;
  }
  y[i] = r;
}
namespace {
// what a kernel gets of the caller's ble_state_f32: its device pointers (the struct's prefix)
static_assert(sizeof(StateDev) == offsetof(ble_state_f32, vehicle) && offsetof(StateDev, episode_cache) == offsetof(ble_state_f32, episode_cache) &&
              offsetof(StateDev, start_unix) == offsetof(ble_state_f32, start_unix) && offsetof(StateDev, status) == offsetof(ble_state_f32, status),
              "StateDev is ble_state_f32 without its last member");
inline StateDev state_dev(const ble_state_f32* st) {
  StateDev d;
  __builtin_memcpy(&d, st, sizeof d);
  return d;
}
inline int env_lanes() { return kBlock; }   // one environment per lane, all 64 lanes (32 was measured: slower)
// Below BLE_SPLIT_MAX_ENVS environments the one-lane kernel leaves most SIMDs idle (n / 64 waves on 1 024 SIMDs) and the
// four-wave kernel still fits one wave per SIMD: it is the faster one (bit-identical results).  ble_set_step_form() forces a
// form (A/B runs and the parity test); BLE_STEP_SPLIT=0 / 1 / 2 / 4 in the process environment is read ONCE, when the
// library first needs it, as that switch's initial value (it used to be re-read by getenv on every launch: host work on the
// 3 us launch path and a data race with a concurrent setenv).
// g_step_form: -1 not initialised, 0 automatic, 1 / 2 / 4 wavefronts per environment.
std::atomic<int> g_step_form{-1};
inline int step_form_from_environment() {
  const char* e = getenv("BLE_STEP_SPLIT");
  if (e != nullptr && e[0] != 0 && e[1] == 0) {
    if (e[0] == '0') return 1;
    if (e[0] == '1' || e[0] == '4') return 4;
    if (kHavePairForm && e[0] == '2') return 2;
  }
  return 0;
}
inline int step_form() {
  int f = g_step_form.load(std::memory_order_relaxed);
  if (f < 0) {
    int expected = -1;
    const int init = step_form_from_environment();
    g_step_form.compare_exchange_strong(expected, init, std::memory_order_relaxed);
    f = g_step_form.load(std::memory_order_relaxed);
  }
  return f;
}
// returns the number of waves per environment: 1 (ble_step_kernel) or 4 (ble_step_split_kernel); 2 (ble_step_pair_kernel) in -DBLE_WITH_PAIR_FORM builds only
inline int split_waves(int64_t n) {
  const int f = step_form();
  return f != 0 ? f : (n <= BLE_SPLIT_MAX_ENVS ? 4 : 1);
}
inline int launch_split(const ble_state_f32* st, const uint8_t* action, const float* wind_grid, int64_t grid_env_stride,
                        const float* noise_uv, float* reward, uint8_t* terminal, uint8_t* effective_action, uint32_t* err_flags,
                        unsigned long long* active_count, int64_t n, int substeps, int n_steps, void* stream, int waves,
                        const ble_noise_gen* noise = nullptr);
// hipGetLastError is per-thread and sticky: an error left behind by an unrelated runtime call
// of the host application (torch probes pointers / peers at start-up) must not be reported as
// ours, so every launch first drains it, and the launch's own status is kept for
// ble_last_hip_error().
thread_local int g_last_hip_error = 0;
inline int launch_status() {
  const hipError_t e = hipGetLastError();
  g_last_hip_error = (int)e;
  return e == hipSuccess ? BLE_OK : BLE_E_LAUNCH;
}
#define BLE_LAUNCH(...)            \
  do {                             \
    (void)hipGetLastError();       \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)
inline unsigned blocks(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }
inline int launch_split(const ble_state_f32* st, const uint8_t* action, const float* wind_grid, int64_t grid_env_stride,
                        const float* noise_uv, float* reward, uint8_t* terminal, uint8_t* effective_action, uint32_t* err_flags,
                        unsigned long long* active_count, int64_t n, int substeps, int n_steps, void* stream, int waves,
                        const ble_noise_gen* noise) {
  SplitArgs a;
  a.st = state_dev(st); a.action = action; a.wind_grid = wind_grid; a.grid_env_stride = grid_env_stride; a.noise_uv = noise_uv;
  a.reward = reward; a.terminal = terminal; a.effective_action = effective_action; a.err_flags = err_flags;
  a.active_count = active_count; a.n = n; a.substeps = substeps; a.n_steps = n_steps;
  a.gen = noise ? StepNoise{noise->seed, noise->episode, noise->harmonic_cache, (long long)noise->env_offset} : StepNoise{0ull, nullptr, nullptr, 0ll};
  const dim3 grid(blocks(n, kSplitLanes));
#ifdef BLE_WITH_PAIR_FORM
  if (waves == 2) {
    if (noise != nullptr) BLE_LAUNCH(ble_step_pair_kernel<true>, grid, dim3(2 * kSplitLanes), 0, (hipStream_t)stream, a);
    else BLE_LAUNCH(ble_step_pair_kernel<false>, grid, dim3(2 * kSplitLanes), 0, (hipStream_t)stream, a);
    return launch_status();
  }
#endif
  (void)waves;
  if (noise != nullptr) BLE_LAUNCH(ble_step_split_kernel<true>, grid, dim3(kSplitWaves * kSplitLanes), 0, (hipStream_t)stream, a);
  else BLE_LAUNCH(ble_step_split_kernel<false>, grid, dim3(kSplitWaves * kSplitLanes), 0, (hipStream_t)stream, a);
  return launch_status();
}
inline bool state_ok(const ble_state_f32* st) {
  if (!st) return false;
  const void* const* p = reinterpret_cast<const void* const*>(st);
  for (size_t k = 0; k < offsetof(ble_state_f32, episode_cache) / sizeof(void*); ++k)      // (episode_cache is optional)
    if (p[k] == nullptr) return false;
  return true;
}

// ble_vehicle (the reference's dataclass fields) -> the derived constants the lane functions evaluate, in double on the host: the same
// expressions VehicleDefault folds at compile time (a vehicle equal to the defaults yields VehicleDefault's numbers bit for bit:
// tests/test_gpu_vehicle.py flies both instantiations side by side).
inline bool vehicle_ok(const ble_vehicle* v) {
  const double fields[] = {v->envelope_volume_base, v->envelope_volume_dv_pressure, v->envelope_mass, v->envelope_max_superpressure, v->envelope_cod,
                           v->payload_mass, v->nighttime_power_load_w, v->daytime_power_load_w, v->acs_valve_hole_diameter_m, v->battery_capacity_wh,
                           v->mols_lift_gas};
  for (double f : fields)
    if (!(f == f) || f - f != 0.0) return false;                                  // NaN / Inf
  return v->envelope_volume_base > 0.0 && v->envelope_volume_dv_pressure > 0.0 && v->envelope_mass > 0.0 && v->envelope_cod > 0.0 &&
         v->envelope_max_superpressure > 300.0 && v->battery_capacity_wh > 0.0 && v->payload_mass >= 0.0 && v->mols_lift_gas >= 0.0 &&
         v->acs_valve_hole_diameter_m >= 0.0 && v->nighttime_power_load_w >= 0.0 && v->daytime_power_load_w >= 0.0;
}
inline VehicleRt make_vehicle_rt(const ble_vehicle* v) {
  VehicleRt r;
  r.v0 = v->envelope_volume_base; r.dvdp = v->envelope_volume_dv_pressure;
  r.four_dvdp = 4.0 * r.dvdp; r.inv_dvdp = 1.0 / r.dvdp;
  r.inv_cbrt_v0 = 1.0 / cbrt(r.v0);
  r.lift = v->mols_lift_gas;
  r.envelope_mass = v->envelope_mass; r.payload_mass = v->payload_mass; r.he_mass = kHeMolarMassD * r.lift;
  r.dry_mass = r.he_mass + r.envelope_mass + r.payload_mass;                     // balloon.py:417-420 without the air term (kDryMassD's order)
  r.max_sp = v->envelope_max_superpressure;
  r.drag_arg = (2.0 * 9.80665 / v->envelope_cod) * (kGasConstantD / kAirMolarMassD);
  r.thermal_scale = 10.0 * 4.0 * kPiD * 0.38483473658887897 / (1500.0 * r.envelope_mass);
  const double d = v->acs_valve_hole_diameter_m;
  r.valve_k = -0.62 * (kPiD * d * d / 4.0);
  r.night_load_d = v->nighttime_power_load_w; r.capacity_d = v->battery_capacity_wh; r.day_load_d = v->daytime_power_load_w;
  r.day_load = (float)r.day_load_d; r.night_load = (float)r.night_load_d; r.capacity = (float)r.capacity_d;
  r.power_layer = v->power_safety_layer_enabled != 0;
  r.inv_capacity = 1.0 / r.capacity_d;
  r.ceiling_target = (r.payload_mass + r.envelope_mass + r.lift * kHeMolarMassD) * kGasConstantD / (kAirMolarMassD * r.v0);
  r.sp_hi = r.max_sp - 250.0;
  return r;
}

}  // namespace

extern "C" {

int ble_abi_version(void) { return BLE_ABI_VERSION; }

int ble_noise_primitive_version(void) { return BLE_NOISE_PRIMITIVE_VERSION; }

int ble_vehicle_default(ble_vehicle* v) {
  if (v == nullptr) return BLE_E_INVALID_ARG;
  v->envelope_volume_base = 1804.0; v->envelope_volume_dv_pressure = 0.0199; v->envelope_mass = 68.5; v->envelope_max_superpressure = 2380.0;
  v->envelope_cod = 0.25; v->payload_mass = 92.5; v->nighttime_power_load_w = 183.7; v->daytime_power_load_w = 120.4;
  v->acs_valve_hole_diameter_m = 0.04; v->battery_capacity_wh = 3058.56; v->mols_lift_gas = 6830.0; v->power_safety_layer_enabled = 1;
  v->reserved_ = 0;
  return BLE_OK;
}

int ble_last_hip_error(void) { return g_last_hip_error; }

int ble_set_step_form(int waves_per_env) {
  if (waves_per_env != 0 && waves_per_env != 1 && waves_per_env != 4 && !(kHavePairForm && waves_per_env == 2))
    return BLE_E_INVALID_ARG;
  const int before = step_form();
  g_step_form.store(waves_per_env, std::memory_order_relaxed);
  return before;
}

int ble_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return BLE_E_NO_DEVICE;
  return n;
}

int ble_step_f32(const ble_state_f32* st, const uint8_t* action, const float* wind_grid, int64_t grid_env_stride,
                 const float* noise_uv, float* reward, uint8_t* terminal, uint8_t* effective_action,
                 uint32_t* err_flags, unsigned long long* active_count, int64_t n, int substeps, void* stream) {
  if (!state_ok(st) || !action || !wind_grid || !reward || !terminal || n < 0 || substeps < 1 || substeps > BLE_MAX_SUBSTEPS ||
      grid_env_stride < 0)
    return BLE_E_INVALID_ARG;
  if (st->vehicle != nullptr && !vehicle_ok(st->vehicle)) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  const int lanes = env_lanes();
  if (st->vehicle != nullptr) {              // a run-time vehicle (ABI 5): the one-lane form's second instantiation, whatever the batch size
    BLE_LAUNCH((ble_step_kernel<false, VehicleRt>), dim3(blocks(n, lanes * (kStepBlock / 64))), dim3(kStepBlock), 0, (hipStream_t)stream, state_dev(st), action,
               wind_grid, grid_env_stride, noise_uv, reward, terminal, effective_action, err_flags,
               active_count, n, substeps, lanes, 1, StepNoise{0ull, nullptr, nullptr, 0ll}, make_vehicle_rt(st->vehicle));
    return launch_status();
  }
  const int waves = split_waves(n);          // (read once per launch: a concurrent ble_set_step_form cannot split the decision)
  if (waves != 1)
    return launch_split(st, action, wind_grid, grid_env_stride, noise_uv, reward, terminal, effective_action, err_flags, active_count, n,
                        substeps, 1, stream, waves);
  BLE_LAUNCH(ble_step_kernel<false>, dim3(blocks(n, lanes * (kStepBlock / 64))), dim3(kStepBlock), 0, (hipStream_t)stream, state_dev(st), action,
                     wind_grid, grid_env_stride, noise_uv, reward, terminal, effective_action, err_flags,
                     active_count, n, substeps, lanes, 1, StepNoise{0ull, nullptr, nullptr, 0ll}, VehicleDefault{});
  return launch_status();
}

int ble_step_n_f32(const ble_state_f32* st, const uint8_t* action, const float* wind_grid, int64_t grid_env_stride,
                   const ble_noise_gen* noise, float* reward, uint8_t* terminal, uint32_t* err_flags,
                   unsigned long long* active_count, int64_t n, int substeps, int n_steps, void* stream) {
  if (!state_ok(st) || !action || !wind_grid || !reward || !terminal || n < 0 || substeps < 1 || substeps > BLE_MAX_SUBSTEPS || n_steps < 0 ||
      grid_env_stride < 0 || (noise != nullptr && noise->env_offset < 0))      // (a negative offset would key other streams than the reset did)
    return BLE_E_INVALID_ARG;
  if (st->vehicle != nullptr && !vehicle_ok(st->vehicle)) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  if (n_steps == 0) return BLE_OK;
  const int lanes = env_lanes();
  if (st->vehicle != nullptr) {
    const VehicleRt veh = make_vehicle_rt(st->vehicle);
    if (noise != nullptr)
      BLE_LAUNCH((ble_step_kernel<true, VehicleRt>), dim3(blocks(n, lanes * (kStepBlock / 64))), dim3(kStepBlock), 0, (hipStream_t)stream, state_dev(st), action,
                 wind_grid, grid_env_stride, (const float*)nullptr, reward, terminal, (uint8_t*)nullptr, err_flags,
                 active_count, n, substeps, lanes, n_steps, StepNoise{noise->seed, noise->episode, noise->harmonic_cache, (long long)noise->env_offset}, veh);
    else
      BLE_LAUNCH((ble_step_kernel<false, VehicleRt>), dim3(blocks(n, lanes * (kStepBlock / 64))), dim3(kStepBlock), 0, (hipStream_t)stream, state_dev(st), action,
                 wind_grid, grid_env_stride, (const float*)nullptr, reward, terminal, (uint8_t*)nullptr, err_flags,
                 active_count, n, substeps, lanes, n_steps, StepNoise{0ull, nullptr, nullptr, 0ll}, veh);
    return launch_status();
  }
  const int waves = split_waves(n);
  if (waves != 1)          // (with or without the in-kernel noise generator)
    return launch_split(st, action, wind_grid, grid_env_stride, nullptr, reward, terminal, nullptr, err_flags, active_count, n, substeps,
                        n_steps, stream, waves, noise);
  if (noise != nullptr) {
    BLE_LAUNCH(ble_step_kernel<true>, dim3(blocks(n, lanes * (kStepBlock / 64))), dim3(kStepBlock), 0, (hipStream_t)stream, state_dev(st), action,
               wind_grid, grid_env_stride, (const float*)nullptr, reward, terminal, (uint8_t*)nullptr, err_flags,
               active_count, n, substeps, lanes, n_steps, StepNoise{noise->seed, noise->episode, noise->harmonic_cache, (long long)noise->env_offset},
               VehicleDefault{});
  } else {
    BLE_LAUNCH(ble_step_kernel<false>, dim3(blocks(n, lanes * (kStepBlock / 64))), dim3(kStepBlock), 0, (hipStream_t)stream, state_dev(st), action,
               wind_grid, grid_env_stride, (const float*)nullptr, reward, terminal, (uint8_t*)nullptr, err_flags,
               active_count, n, substeps, lanes, n_steps, StepNoise{0ull, nullptr, nullptr, 0ll}, VehicleDefault{});
  }
  return launch_status();
}

int ble_forecast_f32(const float* wind_grid, int64_t grid_env_stride, const float* x_m, const float* y_m,
                     const float* pressure, const int32_t* elapsed_s, float* u, float* v, int64_t n, void* stream) {
  if (!wind_grid || !x_m || !y_m || !pressure || !elapsed_s || !u || !v || n < 0 || grid_env_stride < 0)
    return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(ble_forecast_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, wind_grid,
                     grid_env_stride, x_m, y_m, pressure, elapsed_s, u, v, n);
  return launch_status();
}

int ble_forecast_column_f32(const float* wind_grid, int64_t grid_env_stride, const float* x_m, const float* y_m,
                            const int32_t* elapsed_s, const float* levels_pa, int n_levels, float* out_uv, int64_t n,
                            void* stream) {
  if (!wind_grid || !x_m || !y_m || !elapsed_s || !levels_pa || !out_uv || n < 0 || n_levels < 1 ||
      grid_env_stride < 0)
    return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(ble_forecast_column_kernel, dim3((unsigned)n), dim3(kBlock), 0, (hipStream_t)stream, wind_grid,
                     grid_env_stride, x_m, y_m, elapsed_s, levels_pa, n_levels, out_uv, n);
  return launch_status();
}

int ble_observe_forecast_f32(const ble_state_f32* st, const float* wind_grid, int64_t grid_env_stride, const float* forecast_levels,
                             const float* noise_uv, const uint8_t* reset_mask, const ble_gp_history_f32* hist, int append, float* obs,
                             uint32_t* err_flags, int64_t n, void* stream) {
  if (!state_ok(st) || !wind_grid || !hist || !hist->xyp || !hist->elapsed_s || !hist->err_uv || !hist->count || !obs ||
      n < 0 || grid_env_stride < 0)
    return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  GpHistory h;
  h.xyp = hist->xyp; h.elapsed_s = hist->elapsed_s; h.err_uv = hist->err_uv; h.count = hist->count;
  h.chol = hist->chol; h.n_chol = hist->n_chol; h.chol_stride = hist->chol_stride;
  // a slab shorter than the kernel's layout would be overrun (and overlap the next environment's)
  if (h.chol != nullptr && (h.n_chol == nullptr || h.chol_stride < (int64_t)kCholStride || (h.chol_stride & 1) != 0))
    return BLE_E_INVALID_ARG;
  if (st->vehicle != nullptr) {
    if (!vehicle_ok(st->vehicle)) return BLE_E_INVALID_ARG;
    BLE_LAUNCH(ble_observe_kernel<VehicleRt>, dim3((unsigned)n), dim3(kObsBlock), 0, (hipStream_t)stream, state_dev(st), wind_grid,
               grid_env_stride, noise_uv, reset_mask, h, append, obs, err_flags, n, make_vehicle_rt(st->vehicle), forecast_levels);
  } else {
    BLE_LAUNCH(ble_observe_kernel<VehicleDefault>, dim3((unsigned)n), dim3(kObsBlock), 0, (hipStream_t)stream, state_dev(st), wind_grid,
               grid_env_stride, noise_uv, reset_mask, h, append, obs, err_flags, n, VehicleDefault{}, forecast_levels);
  }
  return launch_status();
}

int ble_observe_f32(const ble_state_f32* st, const float* wind_grid, int64_t grid_env_stride, const float* noise_uv,
                    const uint8_t* reset_mask, const ble_gp_history_f32* hist, int append, float* obs,
                    uint32_t* err_flags, int64_t n, void* stream) {
  return ble_observe_forecast_f32(st, wind_grid, grid_env_stride, nullptr, noise_uv, reset_mask, hist, append, obs, err_flags, n, stream);
}

int ble_decode_flow_fields_f32(const float* flow, float* wind_grid, int64_t n, void* stream) {
  if (!flow || !wind_grid || n < 0 || n > 2147483647LL) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(ble_decode_flow_kernel, dim3((unsigned)n), dim3(kDecodeThreads), 0, (hipStream_t)stream, flow, wind_grid, n);
  return launch_status();
}

int ble_wind_noise_at_f32(const float* x_m, const float* y_m, const float* pressure, const int32_t* elapsed_s,
                          unsigned long long seed, const uint32_t* episode, int mode, uint32_t* harmonic_cache,
                          float* noise_uv, int64_t env_offset, int64_t n, void* stream) {
  if (!x_m || !y_m || !pressure || !elapsed_s || !noise_uv || n < 0 || env_offset < 0 || mode < 0 || mode > 1) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(ble_wind_noise_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, x_m, y_m, pressure,
             elapsed_s, seed, episode, mode, harmonic_cache, noise_uv, n, env_offset);
  return launch_status();
}

int ble_wind_noise_f32(const float* x_m, const float* y_m, const float* pressure, const int32_t* elapsed_s,
                       unsigned long long seed, const uint32_t* episode, int mode, uint32_t* harmonic_cache,
                       float* noise_uv, int64_t n, void* stream) {
  return ble_wind_noise_at_f32(x_m, y_m, pressure, elapsed_s, seed, episode, mode, harmonic_cache, noise_uv, 0, n, stream);
}

int ble_state_rows_f64(const ble_state_f32* st, int64_t first, int64_t count, double* out, int64_t n, void* stream) {
  if (!state_ok(st) || !out || first < 0 || count < 0 || n < 0 || first + count > n) return BLE_E_INVALID_ARG;
  if (count == 0) return BLE_OK;
  BLE_LAUNCH(ble_state_rows_kernel, dim3(blocks(count, 64)), dim3(64), 0, (hipStream_t)stream, state_dev(st), first, count, out);
  return launch_status();
}

int ble_power_table_f32(const float* pressure_ratio, const float* state_of_charge, float* watts, uint32_t* err_flags,
                        int64_t n, void* stream) {
  if (!pressure_ratio || !state_of_charge || !watts || n < 0) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(ble_power_table_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, pressure_ratio,
                     state_of_charge, watts, err_flags, n);
  return launch_status();
}

int ble_probe_atmosphere_f32(const float* alpha, const float* pressure, float* height, float* temperature,
                             uint32_t* err_flags, int64_t n, void* stream) {
  if (!alpha || !pressure || !height || !temperature || n < 0) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_atmosphere_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, alpha, pressure,
                     height, temperature, err_flags, n);
  return launch_status();
}

int ble_probe_atmosphere_at_height_f64(const float* alpha, const double* height_m, double* pressure, double* temperature, uint32_t* err_flags,
                                       int64_t n, void* stream) {
  if (!alpha || !height_m || !pressure || !temperature || n < 0) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_at_height_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, alpha, height_m, pressure, temperature, err_flags, n);
  return launch_status();
}

int ble_probe_solar_f32(const float* center_lat_deg, const float* center_lng_deg, const float* x_m, const float* y_m,
                        const int64_t* unix_s, float* el_deg, float* flux, int64_t n, void* stream) {
  if (!center_lat_deg || !center_lng_deg || !x_m || !y_m || !unix_s || !el_deg || !flux || n < 0)
    return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_solar_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, center_lat_deg,
                     center_lng_deg, x_m, y_m, unix_s, el_deg, flux, n);
  return launch_status();
}

int ble_probe_latlng_f64(const float* center_lat_deg, const float* center_lng_deg, const float* x_m, const float* y_m,
                         double* lat_deg, double* lng_deg, int64_t n, void* stream) {
  if (!center_lat_deg || !center_lng_deg || !x_m || !y_m || !lat_deg || !lng_deg || n < 0) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_latlng_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, center_lat_deg, center_lng_deg, x_m,
             y_m, lat_deg, lng_deg, n);
  return launch_status();
}

int ble_probe_solar_power_f32(const float* el_deg, const float* pressure, float* attenuation, float* power_w,
                              int64_t n, void* stream) {
  if (!el_deg || !pressure || !attenuation || !power_w || n < 0) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_solar_power_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, el_deg,
                     pressure, attenuation, power_w, n);
  return launch_status();
}

int ble_probe_thermal_vehicle_f32(const ble_vehicle* vehicle, const float* volume, const float* t_int, const float* t_amb, const float* pressure,
                                  const float* el_deg, const float* flux, const float* upwelling_ir, float* dtdt,
                                  uint32_t* err_flags, int64_t n, void* stream) {
  if (!volume || !t_int || !t_amb || !pressure || !el_deg || !flux || !upwelling_ir || !dtdt || n < 0 || (vehicle != nullptr && !vehicle_ok(vehicle)))
    return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_thermal_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, volume, t_int,
                     t_amb, pressure, el_deg, flux, upwelling_ir, dtdt, err_flags, n,
                     vehicle != nullptr ? make_vehicle_rt(vehicle).thermal_scale : VehicleDefault::thermal_scale);
  return launch_status();
}
int ble_probe_thermal_f32(const float* volume, const float* t_int, const float* t_amb, const float* pressure,
                          const float* el_deg, const float* flux, const float* upwelling_ir, float* dtdt,
                          uint32_t* err_flags, int64_t n, void* stream) {
  return ble_probe_thermal_vehicle_f32(nullptr, volume, t_int, t_amb, pressure, el_deg, flux, upwelling_ir, dtdt, err_flags, n, stream);
}

int ble_probe_sp_volume_vehicle_f32(const ble_vehicle* vehicle, const float* mols_air, const float* t_int, const float* pressure, float* volume,
                                    float* superpressure, int64_t n, void* stream) {
  if (!mols_air || !t_int || !pressure || !volume || !superpressure || n < 0 || (vehicle != nullptr && !vehicle_ok(vehicle))) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  const double lift = vehicle ? vehicle->mols_lift_gas : VehicleDefault::lift, v0 = vehicle ? vehicle->envelope_volume_base : VehicleDefault::v0;
  const double dvdp = vehicle ? vehicle->envelope_volume_dv_pressure : VehicleDefault::dvdp;
  BLE_LAUNCH(probe_sp_volume_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, mols_air, t_int,
                     pressure, volume, superpressure, n, lift, v0, dvdp);
  return launch_status();
}
int ble_probe_sp_volume_f32(const float* mols_air, const float* t_int, const float* pressure, float* volume,
                            float* superpressure, int64_t n, void* stream) {
  return ble_probe_sp_volume_vehicle_f32(nullptr, mols_air, t_int, pressure, volume, superpressure, n, stream);
}

int ble_reset_at_f32(const ble_state_f32* st, const uint8_t* mask, unsigned long long seed, uint32_t* episode,
                     int sample, uint32_t* err_flags, int64_t env_offset, int64_t n, void* stream) {
  if (!state_ok(st) || n < 0 || env_offset < 0) return BLE_E_INVALID_ARG;
  if (st->vehicle != nullptr && !vehicle_ok(st->vehicle)) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  if (st->vehicle != nullptr)
    BLE_LAUNCH(ble_reset_kernel<VehicleRt>, dim3(blocks(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, state_dev(st), mask, seed,
               episode, sample, err_flags, n, env_offset, make_vehicle_rt(st->vehicle));
  else
    BLE_LAUNCH(ble_reset_kernel<VehicleDefault>, dim3(blocks(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, state_dev(st), mask, seed,
               episode, sample, err_flags, n, env_offset, VehicleDefault{});
  return launch_status();
}

int ble_reset_f32(const ble_state_f32* st, const uint8_t* mask, unsigned long long seed, uint32_t* episode,
                  int sample, uint32_t* err_flags, int64_t n, void* stream) {
  return ble_reset_at_f32(st, mask, seed, episode, sample, err_flags, 0, n, stream);
}

int ble_probe_f64_prims(const double* x, double* y, int op, int64_t n, void* stream) {
  if (!x || !y || n < 0 || op < 0 || op > 8) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, op, n);
  return launch_status();
}

int ble_probe_acs_f32(const float* pressure_ratio, float* power_w, float* efficiency, float* mass_flow, int64_t n,
                      void* stream) {
  if (!pressure_ratio || !power_w || !efficiency || !mass_flow || n < 0) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_acs_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, pressure_ratio,
                     power_w, efficiency, mass_flow, n);
  return launch_status();
}

int ble_probe_safety_f32(int layer, const uint8_t* action, const float* value, const float* alpha, int32_t* clocks,
                         double night_load_w, double capacity_wh, uint8_t* fsm, uint8_t* effective_action,
                         uint32_t* err_flags, int64_t n, void* stream) {
  if (layer < 0 || layer > 2 || !action || !value || !fsm || !effective_action || n < 0) return BLE_E_INVALID_ARG;
  if ((layer == 0 && !alpha) || (layer == 2 && (!clocks || !(capacity_wh > 0.0)))) return BLE_E_INVALID_ARG;
  if (n == 0) return BLE_OK;
  BLE_LAUNCH(probe_safety_kernel, dim3(blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, layer, action, value, alpha,
                     clocks, night_load_w, capacity_wh, fsm, effective_action, err_flags, n);
  return launch_status();
}

}  // extern "C"
