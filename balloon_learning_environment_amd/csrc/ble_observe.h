// Batched observation: PerciatelliFeatureConstructor (env/features.py:269-581) for N
// environments on the device -- one workgroup (4 waves) per environment.
//
//   history    ring of the last 128 (x, y, p, t; err_u, err_v) observations per env in HBM
//   phase 0    every per-env input requested at entry; append the new observation; roles: wave 0 the time-only
//              half of the solar calculator at 6 nodes, wave 1 latlng + the state-only ambient features, wave 2
//              the (x, y, t)-blended pressure column, wave 3 T(p) at the 20 search levels; then the 721-entry
//              solar-elevation table (fp64) filled by all lanes; the <= 120 observations of the 6 h window
//              (wind_gp.py:179-185) compacted into LDS (every wave forms the validity ballots itself)
//   phase 1    wave 0: sunrise/sunset searches on the table -> the remaining ambient features
//              wave 1: 22 cold-start Newton solves (20 levels + ceiling + floor) and the searches for the
//                      reachable pressure range (pressure_range_builder.py:203-275)
//              waves 2-3: the factor of K = s^2 exp(-|d/ls|) + 0.05 I (sklearn GaussianProcessRegressor.fit
//                      refits it every step):
//              * incremental (hist.chol given): the factor of the previous window lives in HBM as Lt D Lt^T
//                (61 KB per env with the drop vector and zeta / d, prefetched with 16-byte loads at kernel
//                entry); the window slides by dropping the oldest observation -- a rank-1 UPDATE of the trailing
//                factor whose only cross-row dependency, p = L22^-1 l21, was solved by the previous call's
//                sweep -- and zeta = Lt^-1 y slides with it (one prefix sum); the newest observation becomes
//                a bordering row after the sweep: O(n^2) instead of O(n^3)
//              * refit (no hist.chol, first call, or an inconsistent history): K built in LDS, then
//                (phase 2) a left-looking blocked Cholesky in panels of 8 columns; zeta from the sweep
//   phase 4    V = Lt^-1 [k_new | e_0 | K*^T] for the bordering row, the next drop vector and the reachable
//              query levels: blocked forward substitution on v_mfma_f64_16x16x4_f64, two 16-column tiles per
//              wave, V resident in registers, inverted 16 x 16 diagonal blocks; mean = v . zeta / d + forecast
//              (= K* K^-1 y), deviation = (s^2 - v^2 / d) / s^2
//   phase 5    (uncertainty, bearing, magnitude) triples centred on the balloon's level
//
// All GP algebra is fp64 like the reference's (cond(K) ~ 3e4).  LDS: 12 KB of row vectors and tables FIRST (inside the
// 16-bit offset of ds_read / ds_write), 8.7 KB (block inverses, aliased with the solar table), then the 58 KB factor
// = 79 KB and <= 256 registers: two workgroups per CU, which is what hides the latency-bound single-wave phases.
// DESIGN.md 3b has the cycle budget.
#pragma once
#include <cstddef>
#include <type_traits>

#include "ble_reset.h"

// Instrumentation hooks.  The product build defines them empty; profiling builds (profiles/build_variant.sh:
// -DBLE_OBS_INSTR_HEADER='"../../profiles/instr/ble_observe_instr.h"' plus -DBLE_OBS_TIMING / -DBLE_OBS_SOLO /
// -DBLE_OBS_PHASE_PROFILE) take their definitions -- in-kernel cycle marks, one-workgroup-per-CU padding, early
// returns after a phase -- from that header, which is not part of the package.
#ifdef BLE_OBS_INSTR_HEADER
#include BLE_OBS_INSTR_HEADER
#else
#define BLE_OBS_INSTR_SHARED
#define BLE_OBS_INSTR_BEGIN() do {} while (0)
#define BLE_OBS_INSTR_END() do {} while (0)
#define BLE_MARK() do {} while (0)
#define BLE_SUB(i) do {} while (0)
#define BLE_SW(i) do {} while (0)
#define BLE_BLK(i) do {} while (0)
#define BLE_STOP(k) do {} while (0)
#define BLE_ROLE_ENTRY_DONE() do {} while (0)
#define BLE_ROLE_BEGIN() do {} while (0)
#define BLE_ROLE_END() do {} while (0)
#endif

namespace ble {

constexpr int kObsLevels = 181;
constexpr int kObsColumn = 2 * kObsLevels - 1;      // 361
constexpr int kObsDim = 3 * kObsColumn + 16;        // 1099
constexpr int kGpCapacity = 128;                    // ring entries per env (BLE_GP_CAPACITY)
constexpr int kGpMax = 120;                         // 6 h / 180 s
constexpr int kGpRows = 128;
constexpr int kCholTri = kGpMax * (kGpMax + 1) / 2;      // 7260 doubles: the packed factor
constexpr int kCholStride = kCholTri + 3 * kGpMax;       // + the drop vector p and zeta_u / d, zeta_v / d (below): 7620 doubles = 60 960 B per environment
constexpr int kCholPrefetch = (kCholTri / 2 + 255) / 256;      // double2 loads per lane (15)
constexpr int kObsBlock = 256;
constexpr int kElevTable = 721;                     // t + 180 s * m, m in [-240, 480]
constexpr double kGpSigma2 = 3.6 * 3.6;             // wind_gp.py:36
constexpr double kGpNoise2 = 0.05;                  // wind_gp.py:37
constexpr int kGpHorizonS = 6 * 3600;               // wind_gp.py:63
constexpr uint32_t kFlagGpWindow = 64u;             // more than 120 observations inside 6 h
constexpr uint32_t kFlagPressureSearch = 128u;      // pressure_range_builder raised ValueError
constexpr uint32_t kFlagDayCycle = 256u;            // features.py:432-437 divides by zero (polar night: sunrise a day after sunset)

struct GpHistory {
  float* xyp;          // [n][128][3]
  int32_t* elapsed_s;  // [n][128]
  float* err_uv;       // [n][128][2]
  int32_t* count;      // [n]
  double* chol;        // [n][kCholStride] packed Cholesky factor of the current window, or nullptr
  int32_t* n_chol;     // [n] rows of `chol` that are valid (the window it was computed for ends at `count`)
  int64_t chol_stride; // doubles between consecutive environments' slabs (>= kCholStride, checked by the host entry point)
};

// LDS layout: everything the sweep addresses with a per-lane base + a compile-time offset sits in the first 24 KB,
// inside the 16-bit immediate of ds_read / ds_write; the 58 KB factor comes last.  (With the factor first -- rounds 1-4 --
// the small arrays lay above 64 KB: every access took a v_mov of its base and a v_add, ~250 of the 850 vector instructions of
// a wave's sum pass and tail.)
struct alignas(16) ObsShared {
  double exp2_frac[64];                  // s^2 2^(k / 64): the kernel matrix's exp table (gp_exp_neg_scaled)
  double a[kGpRows];                     // scaled squared (x, y, t) distance to the query column, in units of (ln2 / 32)^2
  double loc[kGpRows][4];                // x, y, p * (32 / ln2) / 326 Pa, t of the observations in the window
  double z[4][kGpRows];                  // 0, 1: error components, then Lt^-1 y;  2: Lt^-1 k_new;  3: Lt^-1 e_0
  double inv_diag[kGpRows];              // 1 / d[i]  (1 / L[i][i] while the refit Cholesky runs)
  double zeros16[16];                    // the off-diagonal part of a virtual identity row inside a diagonal block
  double column[20];                     // the (x, y, t)-blended forecast at the 10 pressure nodes, (u, v) interleaved
  double last[4];                        // new row: zeta_u, zeta_v of the newest observation, its d, (Lt^-1 e_0) there
  double lev[20], pot[20];
  double eph[6][3];                      // (sin decl, cos decl, equation-of-time term) at 6 nodes spanning the elevation table
  double site[3];                        // sin lat, cos lat, lng [deg] of the balloon (computed by one wave)
  double el_now, flux_now, el_next, p_floor;
  int lo_idx, hi_idx;                    // first / last reachable level of the 181
  int n_obs;
  int range_ok;
  int table_done;                        // waves that have finished their share of the elevation table (the rendezvous of phase 0)
  int pad_i;
  double pad[64];                        // sink of the masked stores of the drop recurrences (a select, not a branch)
  alignas(16) double pb[kGpMax][2];      // phase 1: (p_k, beta_k) of the rank-1 update that drops the oldest observation
  alignas(16) union {
    struct {
      double el_table[kElevTable];       // phases 0-1: solar elevation at now + 180 s * (k - 240)
      double pad0;
      double pb3[kGpMax][2];             // phase 1: wave 3's own (p_k, beta_k) pairs (the two drop waves do not synchronise)
      double brow[64];                   // phase 1: old factor row 64 (wave 3 overwrites it while wave 2 still reads it)
    };
    double dinv[kGpRows / 16][138];      // [136]: a zero, read by the lanes above the diagonal of the packed triangle
  };
  alignas(16) double l_guard[16];        // zeros in front of L: the sweep pads the factor at the TOP (virtual identity rows
                                         // 0 .. pad - 1 of the MFMA tiles), so "columns" -pad .. -1 of the first rows are read
  double L[kCholTri];                    // K + noise = Lt D Lt^T, packed lower triangle of rows 0 .. 119: unit-lower Lt
                                         // below the diagonal, d on it
  BLE_OBS_INSTR_SHARED
};
static_assert(offsetof(ObsShared, L) % 16 == 0 && offsetof(ObsShared, pb) % 16 == 0 && offsetof(ObsShared, pb3) % 16 == 0 &&
              offsetof(ObsShared, loc) % 16 == 0, "double2 views of L, pb, pb3 need 16-byte alignment");
static_assert(offsetof(ObsShared, l_guard) + sizeof(double[16]) == offsetof(ObsShared, L), "the zero guard sits directly in front of L");
static_assert(sizeof(ObsShared) <= 80 * 1024, "two workgroups per CU need <= 80 KB of LDS each");

// ---- the kernel matrix entry s^2 exp(-sqrt(a + dp^2)) on PRE-SCALED inputs (round 5: 22 -> 17 vector instructions per entry).
// Distances are carried in units of ln2 / 32 (kGpKappa = 32 / ln 2: the squared (x, y, t) distance `a` times kappa^2, pressures
// times kappa / 326 Pa), so that twice the square root IS the exponent in units of ln2 / 64 and no multiplication by 64 / ln2
// is left on the chain.
//   rs = 2 sqrt(X) = g (3 - g y),  y = rsq(X), g = X y     -- one Newton step written on the product: 4 instructions (the coupled
//        (g, h) form took 5); second-order error 3/8 (1 - X y^2)^2 ~ 4e-15 relative, rounding <= 3e-16
//   f = fract(rs) in [0, 1),  nn = trunc(-rs) = -floor(rs)  -- v_fract_f64 + v_cvt_i32_f64 with a free negation (no v_rndne, no
//        n ln2/64 - r fma, no integer negation)
//   s^2 exp(-r) = 2^(-rs / 64) s^2 = 2^(nn >> 6) * (s^2 2^((nn & 63) / 64)) * 2^(-f / 64)
// tab[k] = s^2 2^(k / 64); 2^(-f / 64) on [0, 1] by its degree-4 minimax polynomial (Remez, |error| <= 2.5e-15 relative; the
// centred Taylor form it replaces was 4e-14 and took the same five instructions).
constexpr double kGpKappa = 46.16624130844683;            // 32 / ln 2
constexpr double kGpInvKappa = 0.02166084939249829;       // ln 2 / 32
constexpr double kGpTwoKappa = 92.33248261689366;         // 64 / ln 2
__device__ __forceinline__ double two_sqrt(double X) {     // 2 sqrt(X), X >= 1e-300
  const double y = d_rsq_seed(X);
  const double g = X * y;
  return g * d_fma(-g, y, 3.0);
}
struct ExpStage { double f, t; int nn; };
// stage A = reduction + table request, stage B = polynomial, scale, exponent (the sweep puts a row block's MFMAs between
// them so that the table read flies under other work)
__device__ __forceinline__ ExpStage exp_scaled_stage_a(double rs, const double* tab) {
  ExpStage e;
  e.f = __builtin_amdgcn_fract(rs);
  e.nn = (int)(-rs);
  e.t = tab[e.nn & 63];
  return e;
}
__device__ __forceinline__ double exp_scaled_stage_b(const ExpStage& e) {
  const double q = d_fma(e.f, d_fma(e.f, d_fma(e.f, d_fma(e.f, 5.701900737872891e-10, -2.117286661914034e-07), 5.864904858491492e-05),
                               -0.010830424696128488), 0.9999999999999976);
  return d_ldexp(e.t * q, e.nn >> 6);
}
__device__ __forceinline__ double gp_exp_neg_scaled(double rs, const double* tab) { return exp_scaled_stage_b(exp_scaled_stage_a(rs, tab)); }

// inclusive prefix sum over the 64 lanes of a wave on DPP moves (no LDS round trips, unlike __shfl_up): Hillis-Steele
// inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of a row into the next row (row_bcast:15 on rows 1
// and 3) and lane 31 into the upper half (row_bcast:31 on rows 2 and 3).  Lanes without a source add 0.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_take(double v) {
  union { double d; int w[2]; } a, b;
  a.d = v;
  b.w[0] = __builtin_amdgcn_update_dpp(0, a.w[0], kCtrl, kRowMask, 0xF, false);
  b.w[1] = __builtin_amdgcn_update_dpp(0, a.w[1], kCtrl, kRowMask, 0xF, false);
  return b.d;
}
__device__ __forceinline__ double wave_inclusive_scan(double v, int /*lane*/) {
  v += dpp_take<0x111, 0xF>(v);        // row_shr:1
  v += dpp_take<0x112, 0xF>(v);        // row_shr:2
  v += dpp_take<0x114, 0xF>(v);        // row_shr:4
  v += dpp_take<0x118, 0xF>(v);        // row_shr:8
  v += dpp_take<0x142, 0xA>(v);        // row_bcast:15 -> rows 1, 3
  v += dpp_take<0x143, 0xC>(v);        // row_bcast:31 -> rows 2, 3
  return v;
}

BLE_FN int tri(int i) { return i * (i + 1) / 2; }

// solar._find_solar_elevation_binary_search (solar.py:295-372) in table-index space: every instant the search
// touches is min_t + 180 s * integer, i.e. an entry of the elevation table.  mode 0 min, 1 max, 2 |el - target|.
// Same decisions as find_solar_elevation (ble_reset.h) on the same values.
__device__ inline int find_in_table(const double* tab, int k_lo, int k_hi, int mode, double target) {
  int low = 0, high = k_hi - k_lo;
  auto obj = [&](int idx) {
    const double el = tab[k_lo + idx];
    return mode == 0 ? el : (mode == 1 ? -el : __builtin_fabs(el - target));
  };
  double ol = obj(low), oh = obj(high);
#pragma unroll 1
  while (high > low + 1) {
    const int span = high - low;                 // midpoint = low + span / 2.0
    if (ol < oh) { high = low + (span + 1) / 2; oh = obj(high); }   // ceil
    else { low = low + span / 2; ol = obj(low); }                   // floor
  }
  return k_lo + ((ol < oh) ? low : high);
}

// Sum over the 4 lanes of a quad (lanes 4k .. 4k+3) with DPP quad_perm moves: no LDS traffic,
// unlike a ds_bpermute-based shuffle.
__device__ __forceinline__ double quad_swap(double v, int xor1) {
  union { double d; int w[2]; } a, b;
  a.d = v;
  if (xor1) {   // quad_perm [1,0,3,2]
    b.w[0] = __builtin_amdgcn_update_dpp(0, a.w[0], 0xB1, 0xF, 0xF, true);
    b.w[1] = __builtin_amdgcn_update_dpp(0, a.w[1], 0xB1, 0xF, 0xF, true);
  } else {      // quad_perm [2,3,0,1]
    b.w[0] = __builtin_amdgcn_update_dpp(0, a.w[0], 0x4E, 0xF, 0xF, true);
    b.w[1] = __builtin_amdgcn_update_dpp(0, a.w[1], 0x4E, 0xF, 0xF, true);
  }
  return b.d;
}

// Sum over the four lanes jq, jq + 16, jq + 32, jq + 48 of a wave, the total in all of them: gfx950's
// v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and even rows of a wave in the VALU (measured:
// swap(x, x) returns the lower part's values in every lane and the upper part's), no LDS crossbar round trip.
__device__ __forceinline__ double column_sum(double v) {
  union { double d; int w[2]; } a, lo, hi;
  a.d = v;
  auto r0 = __builtin_amdgcn_permlane32_swap(a.w[0], a.w[0], false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(a.w[1], a.w[1], false, false);
  lo.w[0] = r0[0]; lo.w[1] = r1[0]; hi.w[0] = r0[1]; hi.w[1] = r1[1];
  a.d = lo.d + hi.d;
  auto q0 = __builtin_amdgcn_permlane16_swap(a.w[0], a.w[0], false, false);
  auto q1 = __builtin_amdgcn_permlane16_swap(a.w[1], a.w[1], false, false);
  lo.w[0] = q0[0]; lo.w[1] = q1[0]; hi.w[0] = q0[1]; hi.w[1] = q1[1];
  return lo.d + hi.d;
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {   // src_lane must be wave-uniform
  union { double d; int w[2]; } a, b;
  a.d = v;
  b.w[0] = __builtin_amdgcn_readlane(a.w[0], src_lane);
  b.w[1] = __builtin_amdgcn_readlane(a.w[1], src_lane);
  return b.d;
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stable_params (ble_reset.h; stable_init.py:40-129) for the superpressure alone, with the TWO evaluations of each
// finite-difference Newton step on a PAIR of lanes (half = 0: T - delta / 2, half = 1: T + delta / 2; exchanged with one
// DPP quad swap): the same arithmetic, half the dependent chain.  Both lanes of a pair return the same value.
template <class V>
__device__ inline double stable_superpressure_paired(double alpha, double p, double el_deg, double flux, double ir, int half,
                                                     uint32_t* flags, const V& veh) {
  const AtmWindow w = atm_window(alpha, p, flags);
  double h, t_amb;
  atm_at_pressure_f64(w, alpha, p, &h, &t_amb);
  const double ma = ((p * kAirMolarMassD * veh.v0 / (kGasConstantD * t_amb) - veh.envelope_mass - veh.payload_mass - veh.he_mass) /
                     kAirMolarMassD);
  const double mols_air = ma > 0.0 ? ma : 0.0;
  const double att = solar_attenuation_f64(el_deg, p);
  double ti = 206.0;
  const double delta = 0.01;
  uint32_t ignored = 0;
  const double q_earth = earth_heat_per_area_f64(ir, &ignored);
#pragma unroll 1
  for (int k = 0; k < 10; ++k) {
    const double mine = thermal_dtdt_f64(veh.v0, veh.inv_cbrt_v0, half ? ti + delta / 2 : ti - delta / 2, t_amb, p, att, flux, q_earth, veh.thermal_scale);
    const double other = quad_swap(mine, 1);
    const double d1 = half ? other : mine, d2 = half ? mine : other;
    // (reciprocals instead of the two fp64 divisions of a step, ~10 instructions each: the iteration converges on a
    // residual of 1e-5, an ulp in the slope does not move its fixed point)
    const double d2t = (d2 - d1) * (1.0 / delta);
    const double mean = (d1 + d2) * 0.5;
    if (fabs(d2t) > 0.0) ti = d_fma(-mean, d_rcp(d2t), ti);
    if (fabs(mean) < 1e-5) break;
  }
  double volume, sp;
  superpressure_volume_f64(mols_air, ti, p, 1.0 / p, &volume, &sp, stride_k_literal(veh.dry_mass, veh.lift, veh.v0), veh.dvdp, veh.four_dvdp, veh.inv_dvdp);
  return sp;
}

// The pressure-range searches run on ONE wave whose lane k < 20 holds search level k (pressure, p / T and, after the
// cold starts, superpressure): the reference's loops over the 20 levels become ballots and lane reads instead of
// 60 dependent LDS round trips on one lane.  `idx` arguments of lane_read must be wave-uniform.
__device__ __forceinline__ double lane_read(double v, int idx) { return readlane_f64(v, __builtin_amdgcn_readfirstlane(idx)); }

// interp1d(p/T -> p, linear, extrapolating) at the float ceiling (pressure_range_builder.py:236-247)
__device__ inline double pressure_ceiling_wave(double lev_l, double pot_l, int lane, double target = VehicleDefault::ceiling_target) {
  const unsigned long long stop = __ballot(lane < 20 && !(pot_l < target));      // searchsorted(side='left'): first level not below
  int i = stop ? __ffsll((long long)stop) - 1 : 20;
  i = i < 1 ? 1 : (i > 19 ? 19 : i);
  const double l1 = lane_read(lev_l, i), l0 = lane_read(lev_l, i - 1), q1 = lane_read(pot_l, i), q0 = lane_read(pot_l, i - 1);
  const double slope = (l1 - l0) / (q1 - q0);
  return slope * (target - q0) + l0;
}

// _search_for_safe_pressure (:111-182) over superpressures that are already solved (sp_l: lane k < 20 = level k).
__device__ inline double safe_pressure_search_wave(double lev_l, double sp_l, int lane, double significant, double sp_sig,
                                                   bool upward, int* ok, double hi = VehicleDefault::sp_hi) {
  const double lo = 250.0;
  if (sp_sig >= lo && sp_sig <= hi) return significant;
  // levels on the far side of `significant`, in scan order (upward: 0 -> 19, else 19 -> 0); the first one whose
  // superpressure is inside [lo, hi] ends the scan, `last` is the level scanned just before it
  const bool valid = lane < 20 && !(upward ? (lev_l < significant) : (lev_l > significant));
  const unsigned long long bv = __ballot(valid), bi = __ballot(valid && !(sp_l > hi || sp_l < lo));
  if (bi == 0) { *ok = 0; return significant; }
  int first, last;
  if (upward) {
    first = __ffsll((long long)bi) - 1;
    const unsigned long long before = bv & ((1ull << first) - 1ull);
    last = before ? 63 - __clzll((long long)before) : -1;
  } else {
    first = 63 - __clzll((long long)bi);
    const unsigned long long before = bv & ~((2ull << first) - 1ull);
    last = before ? __ffsll((long long)before) - 1 : -1;
  }
  const double p = lane_read(lev_l, first), sfirst = lane_read(sp_l, first);
  const double last_p = last >= 0 ? lane_read(lev_l, last >= 0 ? last : 0) : significant;
  const double last_sp = last >= 0 ? lane_read(sp_l, last >= 0 ? last : 0) : sp_sig;
  // _compute_safe_pressure (:73-108): p1 < p2
  const double p1 = upward ? last_p : p, s1 = upward ? last_sp : sfirst;
  const double p2 = upward ? p : last_p, s2 = upward ? sfirst : last_sp;
  double target;
  if ((s1 < lo) != (s2 < lo)) target = lo;
  else if ((s1 > hi) != (s2 > hi)) target = hi;
  else { *ok = 0; return significant; }
  if (!(p1 < p2) || s1 == s2) { *ok = 0; return significant; }
  return fabs((target - s1) / (s2 - s1)) * (p2 - p1) + p1;
}

template <class Veh = VehicleDefault>
__global__ __launch_bounds__(kObsBlock, 2) void ble_observe_kernel(StateDev st, const float* __restrict__ wind_grid,
                                                                int64_t grid_env_stride,
                                                                const float* __restrict__ noise_uv,
                                                                const uint8_t* __restrict__ reset_mask, GpHistory hist,
                                                                int append, float* __restrict__ obs,
                                                                uint32_t* err_flags, int64_t n, Veh veh,
                                                                const float* __restrict__ forecast_levels) {
  __shared__ ObsShared sh;
  BLE_OBS_INSTR_BEGIN();        // (profiling builds only; nothing in the product build)
  BLE_MARK();
  // The phases before the sweep are latency-bound chains on few lanes; the sweep of the other resident workgroup
  // is throughput work.  Priority 1 here, 0 from the sweep on: -3 % per launch (measured).
  __builtin_amdgcn_s_setprio(1);
  const int64_t env = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: `if (wave == ...)` is a real branch, not an exec mask
  uint32_t flags = 0;

  // ---- state of this environment (uniform loads)
  const float xf = st.x[env], yf = st.y[env], pf = st.pressure[env];
  const int32_t elapsed = st.time_elapsed_s[env];
  const int64_t now = st.start_unix[env] + (int64_t)elapsed;
  const double alpha = (double)st.alpha[env];
  // (every per-environment input of every role is requested here, in one memory round trip: read where they are used,
  // the roles of wave 1 paid three dependent ones)
  const float lat0_f = st.center_lat_deg[env], lng0_f = st.center_lng_deg[env];
  const float batt_f = st.battery_charge[env], sp_f = st.superpressure[env], ir_f = st.upwelling_infrared[env];
  const int cmd = st.last_command[env];
  const int paused_bits = (int)st.power_paused[env] | (int)st.env_fsm[env] | (int)st.alt_fsm[env];
  const double x = (double)xf, y = (double)yf, p = (double)pf;
  float* out = obs + env * kObsDim;

  SunSite site;
  if (wave == 1) {           // BalloonState.latlng once per environment (one wave; the others get it through LDS)
    latlng_f64((double)lat0_f, (double)lng0_f, x, y, &site.sin_lat, &site.cos_lat,
               &site.lng_deg);
    if (lane == 0) {
      sh.site[0] = site.sin_lat; sh.site[1] = site.cos_lat; sh.site[2] = site.lng_deg;
    }
  }

  // ---- phase 0a: history ring
  int count = hist.count[env];
  int n_chol0 = hist.chol != nullptr ? hist.n_chol[env] : 0;
  if (reset_mask != nullptr && reset_mask[env] != 0) { count = 0; n_chol0 = 0; }
  const int count0 = count;
  // The stored factor (58 KB of the 61 KB slab) is requested now, 16 B per lane and 15 loads in flight, and lands in
  // LDS after the solar table has been computed: its HBM latency hides behind phase 0b.
  double* chol_g = hist.chol != nullptr ? hist.chol + env * hist.chol_stride : nullptr;
  const int chol_pairs = (tri(n_chol0 <= kGpMax ? n_chol0 : 0) + 1) >> 1;
  double2 chol_pre[kCholPrefetch];
#pragma unroll
  for (int i = 0; i < kCholPrefetch; ++i) {
    const int e2 = tid + kObsBlock * i;
    // (the whole slab, whatever n_chol says: the request does not wait for that load; unused pairs are never copied)
    chol_pre[i] = (chol_g != nullptr && e2 < kCholTri / 2) ? reinterpret_cast<const double2*>(chol_g)[e2] : make_double2(0.0, 0.0);
  }
  const double p_pre = (chol_g != nullptr && tid < kGpMax) ? chol_g[kCholTri + tid] : 0.0;
  const double brow_pre = (chol_g != nullptr && tid >= 128 && tid < 192) ? chol_g[tri(64) + tid - 128] : 0.0;
  const double zu_pre = (chol_g != nullptr && tid < kGpMax) ? chol_g[kCholTri + kGpMax + tid] : 0.0;          // carried zeta / d
  const double zv_pre = (chol_g != nullptr && tid < kGpMax) ? chol_g[kCholTri + 2 * kGpMax + tid] : 0.0;
  const float err_u = noise_uv ? noise_uv[env * 2] : 0.0f, err_v = noise_uv ? noise_uv[env * 2 + 1] : 0.0f;
  float* h_xyp = hist.xyp + env * (kGpCapacity * 3);
  int32_t* h_t = hist.elapsed_s + env * kGpCapacity;
  float* h_err = hist.err_uv + env * (kGpCapacity * 2);
  if (append) {
    if (tid == 0) {
      const int slot = count % kGpCapacity;
      h_xyp[slot * 3] = xf; h_xyp[slot * 3 + 1] = yf; h_xyp[slot * 3 + 2] = pf;
      h_t[slot] = elapsed;
      h_err[slot * 2] = err_u; h_err[slot * 2 + 1] = err_v;
    }
    count += 1;
  }
  const int m = count < kGpCapacity ? count : kGpCapacity;
  bool valid = false;
  float ox = 0, oy = 0, op = 0, oeu = 0, oev = 0;
  int32_t ot = 0;
  if (tid < kGpCapacity && tid < m) {
    if (append && tid == m - 1) {          // the entry lane 0 is writing right now
      ox = xf; oy = yf; op = pf; ot = elapsed; oeu = err_u; oev = err_v;
    } else {
      const int slot = (count - m + tid) % kGpCapacity;
      ox = h_xyp[slot * 3]; oy = h_xyp[slot * 3 + 1]; op = h_xyp[slot * 3 + 2];
      ot = h_t[slot]; oeu = h_err[slot * 2]; oev = h_err[slot * 2 + 1];
    }
  }
  // Every wave also reads the TIMES of ring entries lane and lane + 64: each forms both validity ballots itself, so the
  // window's size and shape need no exchange through LDS (one barrier less before the roles of phase 1).
  int32_t ta = elapsed, tb = elapsed;
  {
    const int ea = lane, eb = lane + 64;
    if (ea < m && !(append && ea == m - 1)) ta = h_t[(count - m + ea) % kGpCapacity];
    if (eb < m && !(append && eb == m - 1)) tb = h_t[(count - m + eb) % kGpCapacity];
  }
  // (the ring entries are consumed after the elevation table below: their HBM round trips -- count, then the
  // slots -- hide behind that computation instead of stalling all four waves here)

  // ---- phase 0b: solar elevation table, search levels, pressure column
  // The table needs the solar elevation at 721 instants 180 s apart.  The time-only half of the solar
  // calculator (declination, equation of time: five sincos) varies on the scale of days: it is evaluated
  // exactly at 6 nodes 7.2 h apart (lanes 0..5) and interpolated (degree 5, Newton forward form; error
  // |f^(6)| h^6 17 / 720 ~ 2e-16) -- only the site half (hour angle, zenith, refraction) runs per entry.
  double* el_table = sh.el_table;
  // (independent of the site and of the ephemeris nodes: before the first barrier, next to wave 0's nodes and
  // wave 1's latlng, instead of after the table where waves 2 / 3 made the others wait at B1)
  double p_floor = 0.0;            // (waves 3 and 1 only: the search levels now, the cold starts in phase 1)
  if (wave == 1 || wave == 3) {
    const double l0 = atm_lapse_f64(0, alpha);
    p_floor = 108870.8213 * d_pow_fast((300.0 + l0 * (15240.0 - -610.0)) / 300.0, -9.80665 / (kAirSpecificGasD * l0));
  }
  if (wave == 3 && lane < 20) {
    // np.linspace(1000, p_floor, 20); p / T(p) at each level (pressure_range_builder.py:222-235)
    const double level = lane == 19 ? p_floor : 1000.0 + (double)lane * ((p_floor - 1000.0) / 19.0);
    const AtmWindow w = atm_window(alpha, level, &flags);
    double h, t;
    atm_at_pressure_f64(w, alpha, level, &h, &t);
    sh.lev[lane] = level; sh.pot[lane] = level / t;
  }
  if (wave == 2 && lane < 20) {
    // get_forecast_column: blend (x, y, t) first, pressure afterwards (grid_based_wind_field.py:96-132), in fp64 from
    // the float32-packed query like scipy's interpn (the bearing feature is an arccos of this wind: an fp32 blend,
    // 1e-6 m/s off, moved it by up to 2e-4 where the wind points at or away from the station)
    const WindQueryD wq = wind_query_xyt_f64(xf, yf, elapsed);
    const float* grid = wind_grid + env * grid_env_stride;
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
    sh.column[lane] = acc;
  }
  if (wave == 2) {
    // (this wave has the shortest role: it also fills the exp table and writes the state-only ambient features)
    sh.exp2_frac[lane] = kGpSigma2 * d_exp_fast((double)lane * (6.93147180559945286227e-01 / 64.0));     // s^2 2^(k / 64)
    if (lane < 16) { sh.zeros16[lane] = 0.0; sh.l_guard[lane] = 0.0; }
    if (lane == 63) {
      // -- the ambient features that need only the state (features.py:400-470);   Reciprocals instead of fp64 divisions: <= 1 ulp of fp64 before the
      //    rounding to float32.
      const double soc = (double)batt_f * veh.inv_capacity;
      const double d2 = x * x + y * y;
      const double inv_d = d2 > 0.0 ? d_rsqrt(d2) : 0.0;
      const double dist_km = d2 * inv_d * 1e-3;
      const double sp_now = (double)sp_f;
      const double ratio = (p + (sp_now > 0.0 ? sp_now : 0.0)) * d_rcp(p);
      const bool paused = paused_bits != 0;
      auto unit = [](double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); };
      out[0] = (float)unit((p - 5000.0) * (1.0 / 9000.0));
      out[1] = (float)soc;
      out[5] = (float)(d2 > 0.0 ? -x * inv_d : 0.0);         // sin(atan2(-x, -y))
      out[6] = (float)(d2 > 0.0 ? -y * inv_d : -1.0);        // cos(atan2(-x, -y)); atan2(-0, -0) = -pi
      out[7] = (float)(dist_km * d_rcp(dist_km + 250.0));
      out[8] = cmd == kUp ? 1.0f : 0.0f; out[9] = cmd == kStay ? 1.0f : 0.0f; out[10] = cmd == kDown ? 1.0f : 0.0f;
      out[11] = paused ? 1.0f : 0.0f; out[12] = paused ? 0.0f : 1.0f;
      out[14] = (float)unit(((double)power_table_lookup_f64(ratio, soc, &flags) - 100.0) * (1.0 / 200.0));
      out[15] = (float)ratio;
    }
  }
  BLE_SUB(0);        // prologue issued (state, latlng on wave 1, ring + factor loads)
  if (tid < 6) {
    double jc, frac;
    unix_day_fraction(now - 43200 + 25920 * (int64_t)tid, &jc, &frac);
    const SolarEphemeris e = solar_ephemeris_f64(jc);
    sh.eph[tid][0] = e.sin_decl; sh.eph[tid][1] = e.cos_decl; sh.eph[tid][2] = e.eot_quarter_deg;
  } else if (tid == 6) {
    double jc, frac;
    unix_day_fraction(now, &jc, &frac);
    sh.flux_now = solar_flux_f64(jc);
  }
  // The carried factor, its drop vector, the copy of row 64 and zeta / d land in LDS HERE (round 5: they did after the elevation
  // table, in front of B2): requested at kernel entry, they have arrived under the roles above, the 60 registers of the prefetch
  // are free for the table, and -- what this buys -- after B1 the second slide wave has everything it reads and starts sliding
  // while the other three fill the table.  Unconditional: a refit overwrites what it does not use.
#pragma unroll
  for (int i = 0; i < kCholPrefetch; ++i) {
    const int e2 = tid + kObsBlock * i;
    if (e2 < chol_pairs) reinterpret_cast<double2*>(sh.L)[e2] = chol_pre[i];
  }
  if (tid < kGpMax) { sh.pb[tid][0] = p_pre; sh.loc[tid][0] = zu_pre; sh.loc[tid][1] = zv_pre; }      // (loc's x, y slots: unused with a carried factor)
  if (tid >= 128 && tid < 192) sh.brow[tid - 128] = brow_pre;
  if (tid == 0) sh.table_done = 0;
  BLE_ROLE_ENTRY_DONE();
  __syncthreads();   // B1
  BLE_SUB(1);        // ephemeris nodes + site ready
  site.sin_lat = sh.site[0]; site.cos_lat = sh.site[1]; site.lng_deg = sh.site[2];
  // ---- the window's shape (every wave forms the ballots itself), and whether the stored factor can be slid
  unsigned long long b0, b1;
  {
    const int32_t age_a = ta > elapsed ? ta - elapsed : elapsed - ta, age_b = tb > elapsed ? tb - elapsed : elapsed - tb;
    b0 = __ballot(lane < m && age_a < kGpHorizonS);              // strict, wind_gp.py:183
    b1 = __ballot(lane + 64 < m && age_b < kGpHorizonS);
  }
  const unsigned long long b_mine = wave == 0 ? b0 : b1;
  valid = wave < 2 && ((b_mine >> lane) & 1ull) != 0;
  const int pos = __popcll(b_mine & ((1ull << lane) - 1ull));
  const int count_w0 = __popcll(b0);
  int n_obs = count_w0 + __popcll(b1);
  int drop = 0;
  if (n_obs > kGpMax) { drop = n_obs - kGpMax; n_obs = kGpMax; flags |= kFlagGpWindow; }
  // Can the stored factor be slid to the new window?  The observations inside the 6 h window
  // must be a suffix of the ring (time only moves forward inside an episode), the stored factor
  // must cover a window that ends where this call started, and the new window must start inside it.
  bool incremental = false;
  int n_dropped = 0;
  {
    const int m0 = m < 64 ? m : 64, m1 = m - m0;
    const unsigned long long inv0 = ~b0 & (m0 >= 64 ? ~0ull : ((1ull << m0) - 1ull));
    const unsigned long long inv1 = ~b1 & (m1 >= 64 ? ~0ull : ((1ull << m1) - 1ull));
    const int last_invalid = inv1 ? 127 - __clzll(inv1) : (inv0 ? 63 - __clzll(inv0) : -1);
    const int first_valid = b0 ? __ffsll((long long)b0) - 1 : (b1 ? 63 + __ffsll((long long)b1) : m);
    n_dropped = (count - n_obs) - (count0 - n_chol0);
    // (at most one observation leaves the window per call when the agent steps are the reference's 180 s;
    // anything else takes the refit path)
    incremental = hist.chol != nullptr && drop == 0 && last_invalid < first_valid && n_dropped >= 0 &&
                  n_dropped <= 1 && n_dropped <= n_chol0 && n_chol0 <= kGpMax;
  }
  // With a factor to slide there is no workgroup barrier between the table and the roles of phase 1 (everything a slide wave
  // reads landed before B1): every wave fills ITS SHARE of the table, announces it in LDS and goes on to its role; only the
  // searches (wave 0) and the cold starts (wave 1) wait for the whole table.  The shares are sized by what follows them
  // (cycles with two workgroups per CU: searches 9.8 k, cold starts 11.0 k, slide rows 0-63 7.7 k, rows 64+ 13.0 k; one pass of
  // 64 entries 1.9 k): wave 2 four passes, waves 0 and 1 three, wave 3 two (the last one 17 entries + the elevation one second
  // from now as entry 721).  Refit: the same shares, then B2 as before.
  const bool early_slide = __builtin_amdgcn_readfirstlane((int)incremental) != 0;      // (workgroup-uniform)
  const int k_first = wave == 2 ? 0 : (wave == 0 ? 256 : (wave == 1 ? 448 : 640));
  const int k_end = wave == 2 ? 256 : (wave == 0 ? 448 : (wave == 1 ? 640 : kElevTable + 1));
  {
    double dd[3][6];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
#pragma unroll
      for (int j = 0; j < 6; ++j) dd[f][j] = sh.eph[j][f];
#pragma unroll
      for (int lvl = 1; lvl < 6; ++lvl)
#pragma unroll
        for (int j = 5; j >= lvl; --j) dd[f][j] -= dd[f][j - 1];          // forward differences, in place
    }
    int64_t days0 = now / 86400;
    int32_t sod_now = (int32_t)(now - days0 * 86400);
    if (sod_now < 0) sod_now += 86400;
    for (int k = k_first + lane; k < k_end; k += 64) {
      const bool in_table = k < kElevTable;                                 // (entry 721: one second from now, is_solar_afternoon, solar.py:239-256)
      const double u = in_table ? (double)k * (1.0 / 144.0) : (43200.0 + 1.0) * (1.0 / 25920.0);      // (t_k - t_0) / 25 920 s
      const double w2 = (u - 1.0) * 0.5, w3 = (u - 2.0) * (1.0 / 3.0), w4 = (u - 3.0) * 0.25, w5 = (u - 4.0) * 0.2;
      double val[3];
#pragma unroll
      for (int f = 0; f < 3; ++f)
        val[f] = d_fma(u, d_fma(w2, d_fma(w3, d_fma(w4, d_fma(w5, dd[f][5], dd[f][4]), dd[f][3]), dd[f][2]), dd[f][1]), dd[f][0]);
      int32_t sod = sod_now + (in_table ? 180 * (k - 240) : 1);
      sod = sod < 0 ? sod + 86400 : sod;                                   // |offset| <= 86 400 s
      sod = sod >= 86400 ? sod - 86400 : sod;
      const double el = solar_elevation_site_f64(site.sin_lat, site.cos_lat, site.lng_deg, (double)sod * (1.0 / 86400.0), val[0], val[1], val[2]);
      *(in_table ? &el_table[k] : &sh.el_next) = el;
    }
  }
  BLE_SUB(2);        // elevation table filled
  BLE_SUB(3);        // search levels / pressure column done
  BLE_MARK();
  BLE_STOP(1);

  // ---- phase 0c: compact the window into LDS (chronological)
  if (wave < 2 && valid) {
    const int at = pos + (wave == 1 ? count_w0 : 0) - drop;
    if (at >= 0) {
      sh.loc[at][2] = (double)op * (kGpKappa / 326.0);
      if (!incremental) {        // positions, times and raw errors: only the refit builds K and solves for zeta
        sh.loc[at][0] = (double)ox; sh.loc[at][1] = (double)oy; sh.loc[at][3] = (double)ot;
        sh.z[0][at] = (double)oeu; sh.z[1][at] = (double)oev;
      }
      const double dx = ((double)ox - x) * (kGpKappa / 357000.0), dy = ((double)oy - y) * (kGpKappa / 357000.0),
                   dt = ((double)ot - (double)elapsed) * (kGpKappa / 34560.0);          // (units of ln2 / 32: gp_exp_neg_scaled)
      sh.a[at] = dx * dx + dy * dy + dt * dt + 1e-300;     // (the guard keeps rsq finite when an observation sits at the query)
    }
  }
  const int n_pad = (n_obs + 15) & ~15;          // identity-padded to the 16-row MFMA tile
  const int n_fac = n_pad < kGpMax ? n_pad : kGpMax;   // rows that exist in LDS (120 is a multiple of the 8-column panel)
  // Rows of the factor the MFMA sweep works on.  Incremental: the window WITHOUT its newest observation
  // (that one becomes a bordering row, folded in after the sweep); refit: the whole window.
  const bool appended = count != count0;
  const bool has_last = incremental && appended;
  const int nr = has_last ? n_obs - 1 : n_obs;
  if (early_slide) {
    // the rendezvous: each wave announces its share of the table (waves 0 and 1 also their part of the compacted window,
    // which is read after B3 only); the searches (wave 0) and the cold starts (wave 1) wait for the whole table, the slide
    // waves need none of it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&sh.table_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (wave < 2) {
      while (__hip_atomic_load(&sh.table_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  } else {
    __syncthreads();   // B2 (refit): the elevation table and the compacted window are in LDS
  }
  const double el_now = sh.el_table[240], flux_now = sh.flux_now;      // entry 240 is `now`

  // ---- phase 1: four roles
  double dnew_keep = 0.0;            // new diagonal entry of the row a drop lane owns (written after B3)
  BLE_ROLE_BEGIN();
  if (wave == 0) {
    // -- ambient features (features.py:400-470).  solar.get_next_sunrise_sunset (solar.py:432-483) is two
    //    pairs of independent searches -- (noon, midnight), then (sunrise, sunset): each pair runs on lanes
    //    0 / 1 at once (same code, different bounds), the bounds of the second pair come by shuffle.
    const bool second = (lane & 1) != 0;
    const bool afternoon = sh.el_next < el_now;
    // (table index k <-> now + 180 s (k - 240); every range below lies inside [now - 12 h, now + 24 h] = [0, 720])
    // stage 1: noon (max elevation; lane 0) and midnight (min; lane 1), each inside its own half day:
    // noon in [t, t + 12 h] in the morning, else [t + 12 h, t + 24 h]; midnight in the other half
    const int lo1 = (second != afternoon) ? 480 : 240;
    const int ext = find_in_table(el_table, lo1, lo1 + 240, second ? 0 : 1, 0.0);
    const int noon = __shfl(ext, 0, 64), midnight = __shfl(ext, 1, 64);
    // stage 2: sunrise (lane 0) between the midnight before noon and noon; sunset (lane 1) between noon and midnight
    const int lo2 = second ? (afternoon ? noon - 480 : noon) : (afternoon ? midnight : midnight - 480);
    const int hi2 = second ? midnight : noon;
    int edge = find_in_table(el_table, lo2, hi2, 2, -4.242);
    if (edge < 240) edge += 480;                           // not before now (solar.py:478-481)
    const int sunrise = __shfl(edge, 0, 64), sunset = __shfl(edge, 1, 64);
    if (lane == 0) {
      // day: pi (now - previous sunrise) / (sunset - previous sunrise); night: pi + the same from the previous
      // sunset to the sunrise.  In units of 180 s the quotients are those of the reference's seconds.
      const bool day = sunset < sunrise;
      const int prev = (day ? sunrise : sunset) - 480;
      if ((day ? sunset : sunrise) == prev) flags |= kFlagDayCycle;     // the reference raises ZeroDivisionError here; the features are NaN
      const double frac = (double)(240 - prev) / (double)((day ? sunset : sunrise) - prev);
      const double cycle = day ? kPiD * frac : kPiD + kPiD * frac;
      double sc, cc;
      sincos_f64(cycle, &sc, &cc);
      const double elc = (el_now + 90.0) * (1.0 / 180.0);
      out[2] = (float)(elc < 0.0 ? 0.0 : (elc > 1.0 ? 1.0 : elc));
      out[3] = (float)sc; out[4] = (float)cc;
      const double soc = (double)batt_f * veh.inv_capacity;
      out[13] = (soc > 0.99 && solar_power_f64(el_now, p) > veh.day_load_d) ? 1.0f : 0.0f;     // balloon.py:231-238
    }
  } else if (wave == 1) {
    // lane k < 20: search level k; lane 20: the float ceiling; lane 21: the floor
    const double lev_l = lane < 20 ? sh.lev[lane] : 0.0, pot_l = lane < 20 ? sh.pot[lane] : 0.0;
    const double ceiling = pressure_ceiling_wave(lev_l, pot_l, lane, veh.ceiling_target);
    // cold starts: level lane / 2 on the lane pair (lane & ~1, lane | 1) -- 44 lanes; then back to one level per lane
    double sp_pair = 0.0;
    const int lvl = lane >> 1;
    const double lev_p = __shfl(lev_l, lvl < 20 ? lvl : 0, 64);
    if (lane < 44) {
      const double level = lvl < 20 ? lev_p : (lvl == 20 ? ceiling : p_floor);
      uint32_t local = 0;
      sp_pair = stable_superpressure_paired(alpha, level, el_now, flux_now, (double)ir_f, lane & 1, &local, veh);
      flags |= local;
    }
    const double sp_l = __shfl(sp_pair, lane < 22 ? 2 * lane : 0, 64);
    // ---- reachable pressure range (pressure_range_builder.py:249-275), as soon as the 22 superpressures exist
    int ok = 1;
    const double p_lo_w = safe_pressure_search_wave(lev_l, sp_l, lane, ceiling, lane_read(sp_l, 20), true, &ok, veh.sp_hi);
    const double p_hi_w = safe_pressure_search_wave(lev_l, sp_l, lane, p_floor, lane_read(sp_l, 21), false, &ok, veh.sp_hi);
    // first and last reachable level of the 181 (features.py:530-536: level >= p_lo && level <= p_hi), here and not
    // in front of the sweep where all four waves would walk these loops
    int lo_i = (int)((p_lo_w - 5000.0) * (1.0 / 50.0)), hi_i = (int)((p_hi_w - 5000.0) * (1.0 / 50.0));
    lo_i = lo_i < 0 ? 0 : (lo_i > 181 ? 181 : lo_i);
    hi_i = hi_i < -1 ? -1 : (hi_i > 180 ? 180 : hi_i);
    while (lo_i > 0 && 5000.0 + 50.0 * (double)(lo_i - 1) >= p_lo_w) --lo_i;     // exactly the reference's comparisons
    while (lo_i <= 180 && 5000.0 + 50.0 * (double)lo_i < p_lo_w) ++lo_i;
    while (hi_i < 180 && 5000.0 + 50.0 * (double)(hi_i + 1) <= p_hi_w) ++hi_i;
    while (hi_i >= 0 && 5000.0 + 50.0 * (double)hi_i > p_hi_w) --hi_i;
    if (lane == 0) { sh.lo_idx = lo_i; sh.hi_idx = hi_i; sh.range_ok = ok; }
  } else if (wave >= 2 && incremental) {
    if (wave == 2 || n_dropped == 1) {
      // ---- drop the oldest observation: lane owns rows `lane` and `lane + 64` of the new factor
      //   K = [k11 k21^T; k21 K22] = Lt D Lt^T with Lt = [1 0; l21 L22], D = diag(d1, D2)
      //   =>  K22 = L22 D2 L22^T + d1 l21 l21^T : a rank-1 update of the trailing LDL^T factor
      // (Gill, Golub, Murray, Saunders 1974, method C1) with weight alpha = d1, carried as gamma = 1 / alpha:
      //   p = L22^-1 l21,  gamma_k = gamma_{k-1} + p_k^2 / d_k,  d'_k = d_k gamma_k / gamma_{k-1},
      //   beta_k = p_k / (d_k gamma_k),  w^(k+1) = w^(k) - p_k L22[:, k],  L22'[:, k] = L22[:, k] + beta_k w^(k+1).
      // The forward substitution p = L22^-1 l21 is the only cross-row dependency of the update -- and p is
      // minus the first column of Lt^-1 below its first entry, i.e. one more right-hand side (e_0) of the
      // MFMA sweep that the PREVIOUS call ran on this very factor: it was stored next to the factor.  With p known
      // the gammas are a prefix sum and every row is an independent recurrence over its own columns: 119
      // wave-synchronous steps of two FMAs per row (was: a 120-step chain of cross-lane broadcasts and
      // reciprocals, 59 k cycles -- the critical path of the kernel).  The result is written one row up and one
      // column left of where L22 was read; step k writes column k and reads column k + 1 of the same storage
      // row, and the loads of step k + 1 are issued before the stores of step k.
      if (n_dropped == 1) {
        const int rows = n_chol0 - 1;                                     // == nr
        const bool own0 = lane < rows, own1 = lane + 64 < rows;
        const double dk0 = own0 ? sh.L[tri(lane + 1) + lane + 1] : 1.0;
        const double pk0 = own0 ? sh.pb[lane][0] : 0.0;
        const double idk0 = d_rcp(dk0);
        const double t0 = pk0 * pk0 * idk0;
        const double s0 = wave_inclusive_scan(t0, lane);
        const double d_first = sh.L[0];                                    // d1 of the dropped row
        const double gamma_start = d_rcp(d_first);
        const double gnew0 = gamma_start + s0, gprev0 = gamma_start + (s0 - t0);
        const double rg0 = d_rcp(gnew0), rgp0 = d_rcp(gprev0);
        // zeta = Lt^-1 y slides with the factor (it is carried as zeta / d next to it, so the sweep does not solve
        // for it): Lt^-1 y = [y_0; L22^-1 (y[1:] - l21 y_0)]  =>  b = L22^-1 y[1:] = zeta[1:] + y_0 p, and with
        // L22' = L22 T, T[j][k] = p_j beta_k (j > k):  zeta' = T^-1 b,  zeta'_i = b_i - p_i u_i / gamma_{i-1},
        // u_i = sum_{k<i} p_k b_k / d_k  (the recurrence s_{i+1} = (1 - beta_i p_i) s_i + beta_i b_i telescopes
        // because 1 - beta_i p_i = gamma_{i-1} / gamma_i): one prefix sum per component.
        const double y0u = sh.loc[0][0] * d_first, y0v = sh.loc[0][1] * d_first;
        const double bu0 = own0 ? d_fma(y0u, pk0, sh.loc[lane + 1][0] * dk0) : 0.0;
        const double bv0 = own0 ? d_fma(y0v, pk0, sh.loc[lane + 1][1] * dk0) : 0.0;
        const double cu0 = pk0 * idk0 * bu0, cv0 = pk0 * idk0 * bv0;
        const double su0 = wave_inclusive_scan(cu0, lane), sv0 = wave_inclusive_scan(cv0, lane);
        const double2* pbv;
        const double* old_row;
        double* new_row;
        double w, inv_new, p_mine = 0.0;
        bool own;
        int my_row;
        if (wave == 2) {
          // rows 0 .. 63.  Old row 64 -- the source of new row 63 -- is overwritten by wave 3's row 64: lane 63
          // reads the copy made before the barrier (brow[0] = l21, brow[1 + k] = column k)
          own = own0; my_row = lane; p_mine = pk0;
          dnew_keep = dk0 * gnew0 * rgp0;
          inv_new = idk0 * gprev0 * rg0;                                     // 1 / d'_k
          if (own0) {
            sh.z[0][lane] = d_fma(-pk0 * rgp0, su0 - cu0, bu0) * inv_new;
            sh.z[1][lane] = d_fma(-pk0 * rgp0, sv0 - cv0, bv0) * inv_new;
          }
          const double* src = lane == 63 ? sh.brow : sh.L + tri(own0 ? lane + 1 : 1);
          w = own0 ? src[0] : 0.0;                                           // l21
          old_row = src + 1;                                                 // old row r + 1 shifted one column
          new_row = sh.L + tri(lane);
          wave_sync_lds();                                                   // every p has been read
          if (own0) sh.pb[lane][1] = pk0 * rg0 * idk0;                        // beta_k
          pbv = reinterpret_cast<const double2*>(&sh.pb[0][0]);
        } else {
          // rows 64 .. 118, with a private copy of all the (p, beta) pairs (the waves do not synchronise)
          own = own1; my_row = lane + 64;
          const double dk1 = own1 ? sh.L[tri(lane + 65) + lane + 65] : 1.0;
          const double pk1 = own1 ? sh.pb[lane + 64][0] : 0.0;
          const double idk1 = d_rcp(dk1);
          const double t1 = pk1 * pk1 * idk1;
          const double s1 = wave_inclusive_scan(t1, lane) + readlane_f64(s0, 63);
          const double gnew1 = gamma_start + s1, gprev1 = gamma_start + (s1 - t1);
          const double rg1 = d_rcp(gnew1), rgp1 = d_rcp(gprev1);
          dnew_keep = dk1 * gnew1 * rgp1; p_mine = pk1;
          inv_new = idk1 * gprev1 * rg1;
          const double bu1 = own1 ? d_fma(y0u, pk1, sh.loc[lane + 65][0] * dk1) : 0.0;
          const double bv1 = own1 ? d_fma(y0v, pk1, sh.loc[lane + 65][1] * dk1) : 0.0;
          const double cu1 = pk1 * idk1 * bu1, cv1 = pk1 * idk1 * bv1;
          const double su1 = wave_inclusive_scan(cu1, lane) + readlane_f64(su0, 63);
          const double sv1 = wave_inclusive_scan(cv1, lane) + readlane_f64(sv0, 63);
          if (own1) {
            sh.z[0][lane + 64] = d_fma(-pk1 * rgp1, su1 - cu1, bu1) * inv_new;
            sh.z[1][lane + 64] = d_fma(-pk1 * rgp1, sv1 - cv1, bv1) * inv_new;
          }
          w = own1 ? sh.L[tri(lane + 65)] : 0.0;
          old_row = sh.L + tri(own1 ? lane + 65 : 1) + 1;
          new_row = sh.L + tri(lane + 64);
          double2* mine = reinterpret_cast<double2*>(sh.pb3);
          if (own0) mine[lane] = make_double2(pk0, pk0 * rg0 * idk0);
          if (own1) mine[lane + 64] = make_double2(pk1, pk1 * rg1 * idk1);
          pbv = mine;
        }
        wave_sync_lds();
        // Every row is slid from BOTH ends at once (lane = row r, columns 0 .. r - 1): columns 0 .. m - 1 forwards from
        // w^(0) = l21_r, columns r - 1 .. m backwards from the diagonal, m = (r + 1) / 2 -- w^(k+1) = sum_{k<j<r} p_j L22[r][j]
        // + p_r because l21 = L22 p, so w^(r) = p_r, L22'[r][k] = L22[r][k] + beta_k w^(k+1), w^(k) = w^(k+1) + p_k L22[r][k].
        // Two independent recurrences per lane, half the steps (59 for row 118): the loop is bound by the latency of its
        // dependent FMAs and LDS round trips, not by their number.  Step j of lane r - 1 reads (backwards) the storage entry
        // that lane r overwrites in the same step, and never one written in an earlier step: the loads of a round precede
        // its stores.  Forward: the (p, beta) pair is the same for all lanes; backward: per lane.
        if (wave == 2 || rows > 64) {
          // (lanes that own no row: r = 0, nothing stored; their loads stay inside the factor or the guard in front of it)
          const int r = own ? my_row : 0;
          const int m = (r + 1) >> 1;
          const int r_max = (wave == 2 ? (rows < 64 ? rows : 64) : rows) - 1;
          const int steps = (r_max + 1) >> 1;
          double wf = w, wb = p_mine;
          const double* fsrc = old_row;
          double* fdst = new_row;
          const double2* fpb = pbv;
          const double* bsrc = old_row + (r - 4);          // columns k_hi - 3 .. k_hi, k_hi = r - 1 - j
          const double2* bpb = pbv + (r - 4);
          double* bdst = new_row + (r - 4);
          // masked stores go to a sink (a select of the address, not a branch).  Lane L's sink is z[2] + L, so store i of
          // the round writes z[2][L + i] / z[2][L + 3 - i]: consecutive lanes hit consecutive banks (the former 4 L spacing put
          // every eighth lane on the same bank pair: a 4-way conflict on each of the 8 stores of a round for the lanes that
          // had finished their half -- profiles/r03_observe_lds.md); lanes overlap in the sink, nobody reads it
          double* sink = &sh.z[2][0] + lane;               // (z[2..3] are idle until the sweep)
          for (int j0 = 0; j0 < steps; j0 += 4) {
            double2 fp[4], bp[4]; double fl[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { fp[i] = fpb[i]; fl[i] = fsrc[i]; bp[i] = bpb[3 - i]; bl[i] = bsrc[3 - i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int kf = j0 + i, kb = r - 1 - j0 - i;
              wf = d_fma(-fp[i].x, fl[i], wf);
              const double nf = d_fma(fp[i].y, wf, fl[i]);
              const double nb = d_fma(bp[i].y, wb, bl[i]);
              wb = d_fma(bp[i].x, bl[i], wb);
              (kf < m ? fdst : sink)[i] = nf;
              (kb >= m ? bdst : sink)[3 - i] = nb;
            }
            fsrc += 4; fdst += 4; fpb += 4;
            // (a lane whose backward half is finished stops moving left: its reads stay inside its own storage row)
            const int back = r - 8 - j0 >= 0 ? 4 : 0;
            bsrc -= back; bpb -= back; bdst -= back;
          }
          if (own) sh.inv_diag[my_row] = inv_new;      // (the new diagonal itself is written after the barrier, below)
        }
      } else {
        // nothing left the window: the factor and zeta / d stand
        for (int r = lane; r < nr; r += 64) {
          sh.inv_diag[r] = d_rcp(sh.L[tri(r) + r]);
          sh.z[0][r] = sh.loc[r][0]; sh.z[1][r] = sh.loc[r][1];
        }
      }
      // rows nr .. of the MFMA tiles are virtual identity rows; they carry no weight in the sums below
      for (int i = nr + lane; i < kGpRows; i += 64) {
        sh.inv_diag[i] = 0.0; sh.z[0][i] = 0.0; sh.z[1][i] = 0.0;
        if (i >= n_obs) { sh.loc[i][2] = 0.0; sh.a[i] = 0.0; }
      }
    }
  } else if (wave >= 2) {
    // -- refit path: K + noise, packed lower triangle; rows n_obs .. n_pad-1 are identity (padding to a
    //    multiple of the panel width: they factor to themselves and contribute nothing)
    const int total = tri(n_fac);
    for (int e = tid - 128; e < total; e += 128) {
      int i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while (tri(i) > e) --i;
      while (tri(i + 1) <= e) ++i;
      const int j = e - tri(i);
      double k_ij;
      if (i < n_obs) {
        // (a - b) / length_scale as a multiplication by the rounded reciprocal: 1e-16 relative
        const double d0 = (sh.loc[i][0] - sh.loc[j][0]) * (1.0 / 357000.0), d1 = (sh.loc[i][1] - sh.loc[j][1]) * (1.0 / 357000.0),
                     d2 = (sh.loc[i][2] - sh.loc[j][2]) * kGpInvKappa, d3 = (sh.loc[i][3] - sh.loc[j][3]) * (1.0 / 34560.0);
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        const double r = r2 > 0.0 ? r2 * d_rsqrt(r2) : 0.0;
        k_ij = kGpSigma2 * d_exp_fast(-r) + (i == j ? kGpNoise2 : 0.0);
      } else {
        k_ij = i == j ? 1.0 : 0.0;
      }
      sh.L[e] = k_ij;
    }
    for (int i = n_obs + (tid - 128); i < n_pad; i += 128) { sh.z[0][i] = 0.0; sh.z[1][i] = 0.0; sh.loc[i][2] = 0.0; sh.a[i] = 0.0; }
    if (tid == 128) {        // slot 127 of the row vectors: what the sweep reads for its virtual rows
      sh.z[0][kGpRows - 1] = 0.0; sh.z[1][kGpRows - 1] = 0.0; sh.loc[kGpRows - 1][2] = 0.0; sh.a[kGpRows - 1] = 0.0; sh.inv_diag[kGpRows - 1] = 0.0;
    }
  }
  BLE_ROLE_END();
  __syncthreads();   // B3  (el_table is dead from here on: V may be overwritten)
  BLE_MARK();
  BLE_STOP(2);
  // the drop's new diagonal: wave 3 read the old diagonals of wave 2's rows, so they are replaced only now
  // (no reader before the epilogue: the sweep works on 1 / d in inv_diag and on strictly-lower entries)
  if (wave >= 2 && incremental && n_dropped == 1) {
    const int r = lane + 64 * (wave - 2);
    if (r < nr) sh.L[tri(r) + r] = dnew_keep;
  }

  // ---- phase 2: Cholesky, left-looking, panels of 8 columns.  Thread (slot, half): slot = row
  // of the trailing matrix, half = which half of the j range.
  {
    const int slot = tid >> 1, half = tid & 1;
    for (int c0 = 0; c0 < (incremental ? 0 : n_fac); c0 += 8) {
      const int rows = n_fac - c0;
      const bool is_matrix = slot < rows;
      const int i = c0 + slot;                                   // matrix row
      double* rowbase = sh.L + tri(is_matrix ? i : 0);
      const int width = slot < 8 ? slot + 1 : 8;   // stored columns of this row inside the panel
      double acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = 0.0;
      if (is_matrix) {
        const double* col[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) col[c] = sh.L + tri(c0 + c);
#pragma unroll 2
        for (int j = half; j < c0; j += 2) {
          const double lij = rowbase[j];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] = d_fma(lij, col[c][j], acc[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] += quad_swap(acc[c], 1);
      double av[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) av[c] = (is_matrix && c < width) ? rowbase[c0 + c] - acc[c] : 0.0;
      // the panel's own rows publish their updated entries: the 8 x 8 diagonal block
      if (is_matrix && slot < 8 && half == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < width) rowbase[c0 + c] = av[c];
      }
      __syncthreads();
      // barriers stay outside divergent code: a wave that executes both sides of a branch would
      // arrive twice
      double xr[8], invd[8], dd[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { xr[c] = 0.0; invd[c] = 0.0; dd[c] = 0.0; }
      if (is_matrix) {
        // every thread factors the diagonal block in registers (SIMT: free) ...
        double d[8][8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) d[r][c] = sh.L[tri(c0 + r) + c0 + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          d[k][k] = sqrt(d[k][k]);
          invd[k] = 1.0 / d[k][k];
          dd[k] = d[k][k];
#pragma unroll
          for (int r = k + 1; r < 8; ++r) d[r][k] *= invd[k];
#pragma unroll
          for (int r = k + 1; r < 8; ++r)
#pragma unroll
            for (int c = k + 1; c <= r; ++c) d[r][c] = d_fma(-d[r][k], d[c][k], d[r][c]);
        }
        // ... and solves its own row against it
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          double t = av[c];
#pragma unroll
          for (int k = 0; k < c; ++k) t = d_fma(-xr[k], d[c][k], t);
          xr[c] = t * invd[c];
        }
      }
      __syncthreads();        // all reads of the un-factored diagonal block are done
      if (is_matrix && half == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < width) rowbase[c0 + c] = c == slot ? dd[c] : xr[c];
        if (slot == 0) {
#pragma unroll
          for (int c = 0; c < 8; ++c) sh.inv_diag[c0 + c] = invd[c];
        }
      }
      __syncthreads();
    }
    if (!incremental) {
      // L L^T -> Lt D Lt^T (the form the incremental slide and the sweep below work on):
      // Lt[i][j] = L[i][j] / L[j][j], d[j] = L[j][j]^2
      const int total = tri(n_fac);
      for (int e = tid; e < total; e += kObsBlock) {
        int i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (tri(i) > e) --i;
        while (tri(i + 1) <= e) ++i;
        const int j = e - tri(i);
        const double v = sh.L[e];
        sh.L[e] = i == j ? v * v : v * sh.inv_diag[j];
      }
      __syncthreads();
      for (int j = tid; j < n_pad; j += kObsBlock) { const double r = j < n_fac ? sh.inv_diag[j] : 1.0; sh.inv_diag[j] = r * r; }
      __syncthreads();
    }
  }
  BLE_MARK();
  BLE_MARK();

  // ---- phases 4 + 5: the 181-level column
  const double p_clamped = p < 5000.0 ? 5000.0 : (p > 14000.0 ? 14000.0 : p);
  const int level_now = (int)d_rint((p_clamped - 5000.0) / 50.0);       // Python round(): half to even
  const int pad_above = kObsLevels - level_now - 1;
  const double dist2 = x * x + y * y;
  const double dist = dist2 > 0.0 ? dist2 * d_rsqrt(dist2) : 0.0;
  const double inv_dist = d_rcp(dist + 1e-5);
  const double to_station_x = -x * inv_dist, to_station_y = -y * inv_dist;
  // Rows of the sweep: nr padded to the 16-row MFMA tile with virtual identity rows AT THE TOP (MFMA row m = factor row
  // m - pad_top).  A partial block costs its K-steps only where it is a K dimension: at the top, block column 0 has
  // 4 - pad_top / 4 of them per row block instead of 4 (119 rows: 128 MFMAs per tile instead of 144); at the bottom it would
  // save nothing.
  const int n_pad_s = (nr + 15) & ~15;
  const int pad_top = n_pad_s - nr;
  // -- inverses of the 16 x 16 (unit lower) diagonal blocks of Lt (thread = (block, column): forward substitution)
  if (tid < 128) {
    sh.z[3][tid] = (tid == 0 && nr > 0) ? 1.0 : 0.0;            // right-hand side e_0 of the drop-vector column
    const int blk = tid >> 4, c = tid & 15, base = blk * 16;
    if (base < n_pad_s) {
      double xcol[16];
      const int shift = base - pad_top;                         // factor column of the block's column 0 (block 0: -pad_top)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double t = r == c ? 1.0 : 0.0;
        const int ir = base + r - pad_top;                      // factor row; virtual above the window
        // one select per row, not per entry; a real row's virtual columns (block 0) read the zero guard or finite
        // entries of earlier rows against x[k] = 0
        const double* lrow = ir >= 0 ? sh.L + tri(ir >= 0 ? ir : 0) + shift : sh.zeros16;
#pragma unroll
        for (int k = 0; k < r; ++k) t = d_fma(-lrow[k], xcol[k], t);
        xcol[r] = (r < c) ? 0.0 : t;                          // unit diagonal
      }
      if (base + c < pad_top) {                                 // a virtual column is an identity column
#pragma unroll
        for (int r = 0; r < 16; ++r) xcol[r] = r == c ? 1.0 : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r >= c) sh.dinv[blk][tri(r) + c] = xcol[r];
    }
  }
  else if (tid < 136) sh.dinv[tid - 128][136] = 0.0;
  __syncthreads();
  BLE_MARK();
  BLE_STOP(3);
  // (the pressure-range search ran on wave 1 in phase 1: two barriers ago)
  if (sh.range_ok == 0 && tid == 0) flags |= kFlagPressureSearch;

  // -- V = Lt^-1 [y | k_new | e_0 | K*^T] with v_mfma_f64_16x16x4 (K + noise = Lt D Lt^T, Lt unit lower, nr rows).
  // Columns 0, 1 = the two error vectors (zeta = Lt^-1 y); 2 = the kernel between the newest observation and the
  // older ones -- it sits AT the query column, so that is K* for the "level" p_balloon -- whose solution is the
  // bordering row of the factor; 3 = e_0, whose solution is next call's drop vector; then ONLY the reachable
  // levels lo_idx .. hi_idx -- the others are (0, 1, 1) whatever the GP says (features.py:530-536).  Typically
  // 117-124 of the 181 levels are reachable: 8 tiles of 16 columns = two per wave.  Tile T = 4 round + wave.
  // MFMA register layout (measured on gfx950): A lane l holds A[l % 16][l / 16], B lane l holds
  // B[l / 16][l % 16], D lane l register v holds D[4 v + l / 16][l % 16] -- a D tile is therefore directly
  // the four B operands of the next product, and V never leaves the registers.
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int g = lane >> 4, jq = lane & 15;
  const int nb = n_pad_s >> 4;
  const int lo_idx = sh.lo_idx, hi_idx = sh.hi_idx;                                // (wave 1, phase 1)
  const int n_reach = hi_idx >= lo_idx ? hi_idx - lo_idx + 1 : 0;
  // special columns: with a carried factor zeta slid with it in phase 1, so only k_new and e_0 ride the sweep
  const int kSpecial = incremental ? 2 : 4;
  const int c_new = incremental ? 0 : 2, c_e0 = incremental ? 1 : 3;
  const int n_tiles = (kSpecial + n_reach + 15) >> 4;                                // 1 .. 12
  // Each wave sweeps its tiles {wave, wave + 4} TOGETHER (independent accumulators keep the matrix pipe busy).  The
  // rare ninth .. twelfth tile (1 % of the environments once zeta is off the sweep) is a second, barrier-free pass
  // of the wave that owns it (kFirst = false: no special columns, nothing written to LDS).
  auto sweep = [&](auto nt_tag, auto first_tag, int tile_base) {
    constexpr int NT = decltype(nt_tag)::value;
    constexpr bool kFirst = decltype(first_tag)::value;
    d4 V[NT][8];                                    // (row blocks I >= nb are never formed and never read)
    int col[NT];
    double level[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      col[t] = 16 * (tile_base + 4 * t + wave) + jq;
      level[t] = 5000.0 + 50.0 * (double)(lo_idx + col[t] - kSpecial);
    }
    if (kFirst && wave == 0 && jq == c_new) level[0] = p;     // column c_new: the newest observation's own pressure
    double level_s[NT];                             // in units of the pressure length scale
#pragma unroll
    for (int t = 0; t < NT; ++t) level_s[t] = level[t] * (kGpKappa / 326.0);
    const double y_last_u = (double)err_u, y_last_v = (double)err_v;   // raw errors of the newest observation (when has_last)
    const int spec_sel = jq == c_e0 ? 3 : (jq < 2 ? jq : 0);        // refit: columns 0, 1 = y_u, y_v from z[0], z[1]
    const bool use_spec = jq < kSpecial && jq != c_new;
    // Row blocks below the first: the kernel-matrix value of a special column is multiplied away (R = K * mask0 - acc: the fma
    // replaces the subtraction, so the common path pays nothing -- the selects and the per-block loads of the right-hand sides
    // cost every wave 12 vector instructions per row block until round 4).  With a carried factor the only special right-hand
    // side left on the sweep is e_0, non-zero in factor row 0 alone, i.e. in row block 0; the refit's y_u, y_v are added under
    // a scalar branch.
    const double mask0 = (kFirst && wave == 0 && use_spec) ? 0.0 : 1.0;
    const bool refit_rhs = kFirst && wave == 0 && __builtin_amdgcn_readfirstlane((int)incremental) == 0;       // scalar
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    const int c0 = pad_top >> 2;          // K-steps 0 .. c0 - 1 of block column 0 hold only virtual rows: skipped
    // factor row of MFMA row m is m - pad_top; virtual rows (block 0 only) read the all-zero slot 127 of the row vectors
    // (as one per-lane offset computed once: slot of MFMA row 4 v + g of block 0, and g - pad_top for the blocks below,
    // whose rows are all real -- their indices are then compile-time offsets from it)
    const int off_rest = g - pad_top;
    const int r0 = jq - pad_top;
    const double* arow_base = sh.L + ((r0 * (r0 + 1)) >> 1) - pad_top + g;        // (r0 (r0 + 1) is even for negative r0 too)
    const int arow_step = 16 * r0;
    // packed lower triangle of a block inverse: the lanes above the diagonal read the zero at [136] (an offset chosen
    // once per lane, not a compare + select per load)
    int doff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) doff[c] = 4 * c + g <= jq ? tri(jq) + g + 4 * c : 136;
    auto row_slot = [&](int I, int v) { return I > 0 ? off_rest + (16 * I + 4 * v) : (4 * v + off_rest < 0 ? kGpRows - 1 : 4 * v + off_rest); };
#pragma unroll
    for (int I = 0; I < 8; ++I) {
      if (I < nb) {
        // the first product of each chain takes a literal-zero accumulator (no register zeroing)
        d4 acc[NT];
        // (row blocks I >= 1 hold real rows only; their virtual columns -pad_top .. -1 read the guard in front of L or
        // finite entries of the row above, against the zero rows of V)
        // (tri(16 I + r0) = tri(r0) + 16 I r0 + 8 I (16 I + 1), r0 = jq - pad_top: one multiply-add per row block on a per-lane
        // base and step instead of the ten instructions of the triangle index)
        const double* arow = I > 0 ? arow_base + I * arow_step + 8 * I * (16 * I + 1) : sh.L + g;
        // right-hand sides of the special columns (tile 0): requested here, consumed after the products below --
        // read next to their use, each of the 32 loads was a full LDS round trip of the slowest wave
        double spec[4];
        const bool need_spec = kFirst && wave == 0 && (I == 0 || refit_rhs);                 // scalar
        if (need_spec) {
#pragma unroll
          for (int v = 0; v < 4; ++v) spec[v] = sh.z[spec_sel][row_slot(I, v)];
        }
        // (likewise the per-row inputs of the kernel matrix and the packed inverse of the diagonal block)
        double a_rows[4], p_rows[4], dpk[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          a_rows[v] = sh.a[row_slot(I, v)]; p_rows[v] = sh.loc[row_slot(I, v)][2];
          dpk[v] = sh.dinv[I][doff[v]];
        }
        // The A operands of the products (rows of the factor): block columns 0 and 1 are requested here, column J + 1 in front
        // of column J's products (round 5).  The compiler read each pair of operands right in front of the MFMAs that consume it,
        // into the same registers: ~56 exposed LDS round trips per wave and sweep, a quarter of the sweep core's cycles.
        double a_col0[4] = {0.0, 0.0, 0.0, 0.0}, a_next[4] = {0.0, 0.0, 0.0, 0.0};
        if (I > 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a_col0[c] = arow[4 * c];          // (virtual columns: the zero guard or finite entries, unused)
        }
        if (I > 1) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a_next[c] = arow[16 + 4 * c];
        }
        __builtin_amdgcn_sched_barrier(0);
        // kernel matrix of the block, stage A (distance, square root, exp reduction, table request) BEFORE the block's
        // products: the table reads land under the MFMAs; stage B (polynomial) after them
        // (block 0: the rows 4 v + g of a register v < c0 are all virtual -- their K-steps of the block inverse are skipped
        // below, so the entries are neither evaluated nor read)
        ExpStage kst[NT][4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (I > 0 || v >= c0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const double dp = level_s[t] - p_rows[v];
              kst[t][v] = exp_scaled_stage_a(two_sqrt(d_fma(dp, dp, a_rows[v])), sh.exp2_frac);
            }
          }
        // one K-step of the block row: acc (+)= L[I][J](:, 4c .. 4c+3) V[J](4c .. 4c+3, :)
        auto kstep = [&](auto j_tag, auto c_tag, auto first_tag) {
          constexpr int J = decltype(j_tag)::value, c = decltype(c_tag)::value;
          const double a = a_col0[c];
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, V[t][J][c], decltype(first_tag)::value ? zero4 : acc[t], 0, 0, 0);
        };
        using std::integral_constant;
        if (I > 0) {
          // block column 0: K-steps c0 .. 3, the first one with the literal-zero accumulator
          typedef integral_constant<int, 0> J0; typedef std::true_type T; typedef std::false_type F;
          typedef integral_constant<int, 0> C0; typedef integral_constant<int, 1> C1;
          typedef integral_constant<int, 2> C2; typedef integral_constant<int, 3> C3;
          if (c0 == 0) { kstep(J0{}, C0{}, T{}); kstep(J0{}, C1{}, F{}); kstep(J0{}, C2{}, F{}); kstep(J0{}, C3{}, F{}); }
          else if (c0 == 1) { kstep(J0{}, C1{}, T{}); kstep(J0{}, C2{}, F{}); kstep(J0{}, C3{}, F{}); }
          else if (c0 == 2) { kstep(J0{}, C2{}, T{}); kstep(J0{}, C3{}, F{}); }
          else { kstep(J0{}, C3{}, T{}); }
        }
#pragma unroll
        for (int J = 1; J < I; ++J) {
          double a_cur[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) a_cur[c] = a_next[c];
          if (J + 1 < I) {
#pragma unroll
            for (int c = 0; c < 4; ++c) a_next[c] = arow[16 * (J + 1) + 4 * c];
          }
          __builtin_amdgcn_sched_barrier(0);          // the next block column's operands are requested BEFORE this one's products
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[c], V[t][J][c], acc[t], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        d4 R[NT];
#pragma unroll
        for (int v = 0; v < 4; ++v) if (I > 0 || v >= c0) {
          const int row = 16 * I + 4 * v + g;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            // branch-free: every lane evaluates the kernel (columns past the last reachable level are unused);
            // s^2 is folded into the exp table; a_row carries a 1e-300 guard for r2 == 0
            R[t][v] = exp_scaled_stage_b(kst[t][v]);
          }
          if (I == 0) {
            const bool real_row = row >= pad_top;
            if (pad_top > 0) {                                // scalar: only the first block holds virtual rows
#pragma unroll
              for (int t = 0; t < NT; ++t) R[t][v] = real_row ? R[t][v] : 0.0;
            }
            if (kFirst && wave == 0) {                        // scalar branch: tile 0 holds the special columns
              // unconditional LDS reads + selects (a load under a per-lane condition becomes an exec-mask branch);
              // z[3] holds e_0 until the solved column overwrites it
              R[0][v] = use_spec ? (real_row ? spec[v] : 0.0) : R[0][v];
            }
          } else {
            R[0][v] = d_fma(R[0][v], mask0, -acc[0][v]);
#pragma unroll
            for (int t = 1; t < NT; ++t) R[t][v] -= acc[t][v];
          }
        }
        if (I > 0 && need_spec) {                                      // (refit only: y_u, y_v in every row)
          asm volatile("; refit right-hand sides");                    // keeps this ONE scalar branch per row block (the optimiser
#pragma unroll                                                         // flattened it into five selects and adds per register for
          for (int v = 0; v < 4; ++v) R[0][v] += use_spec ? spec[v] : 0.0;      // every wave)
        }
        // V[I] = Dinv[I] R.  Block 0: the K-steps that hold only virtual rows (zero rows of R against identity columns)
        // are skipped like those of block column 0 above
        auto dstep = [&](auto c_tag, auto first_tag) {
          constexpr int c = decltype(c_tag)::value;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            V[t][I] = __builtin_amdgcn_mfma_f64_16x16x4f64(dpk[c], R[t][c], decltype(first_tag)::value ? zero4 : V[t][I], 0, 0, 0);
        };
        {
          typedef std::true_type T; typedef std::false_type F;
          typedef integral_constant<int, 0> C0; typedef integral_constant<int, 1> C1;
          typedef integral_constant<int, 2> C2; typedef integral_constant<int, 3> C3;
          if (I > 0 || c0 == 0) { dstep(C0{}, T{}); dstep(C1{}, F{}); dstep(C2{}, F{}); dstep(C3{}, F{}); }
          else if (c0 == 1) { dstep(C1{}, T{}); dstep(C2{}, F{}); dstep(C3{}, F{}); }
          else if (c0 == 2) { dstep(C2{}, T{}); dstep(C3{}, F{}); }
          else { dstep(C3{}, T{}); }
        }
      }
      BLE_BLK(I);
    }
    BLE_SW(0);       // core done (this wave)
    __builtin_amdgcn_s_setprio(1);      // what follows are short dependent chains again (-1.5 %)
    // (no barrier here: only wave 0 reads z before this point, and it writes after its own reads)
    if (kFirst && wave == 0 && jq < kSpecial) {       // the special columns of tile 0
      // omega and the e_0 solution leave raw: every other wave waits for this store, so it carries no loads (the
      // readers fold 1 / d in); only the refit's zeta_u, zeta_v leave divided by d, the form they are carried in
      const int dst = jq == c_e0 ? 3 : (jq == c_new ? 2 : jq);
      double* zdst = &sh.z[dst][0];
      if (dst < 2) {
        // (the 32 scale factors are loaded unconditionally and together: a load under a per-lane condition is an
        // exec-mask branch with its own LDS round trip -- 32 serial ones took 4.7 k cycles of the critical wave)
        double scale[8][4];
#pragma unroll
        for (int I = 0; I < 8; ++I)
#pragma unroll
          for (int v = 0; v < 4; ++v) scale[I][v] = sh.inv_diag[row_slot(I, v)];
#pragma unroll
        for (int I = 0; I < 8; ++I)
          if (I < nb) {
#pragma unroll
            for (int v = 0; v < 4; ++v) zdst[row_slot(I, v)] = V[0][I][v] * scale[I][v];
          }
      } else {
        // (virtual rows hold zeros and go to the all-zero slot 127)
#pragma unroll
        for (int I = 0; I < 8; ++I)
          if (I < nb) {
#pragma unroll
            for (int v = 0; v < 4; ++v) zdst[row_slot(I, v)] = V[0][I][v];
          }
      }
    }
    if constexpr (kFirst) __syncthreads();
    // k* K^-1 k* = sum w^2 / d,  k* K^-1 y = sum w zeta / d  (zeta = Lt^-1 y),  and -- for the bordering row --
    // the same sum against omega = Lt^-1 k_new
    BLE_SW(1);       // special columns in LDS
    double ssq[NT], mean_u[NT], mean_v[NT], cross[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { ssq[t] = 0.0; mean_u[t] = 0.0; mean_v[t] = 0.0; cross[t] = 0.0; }
#pragma unroll
    for (int I = 0; I < 8; ++I) {
      if (I < nb) {
        // (the 16 LDS operands of a row block are requested together, then consumed: one round trip per block)
        double inv_d[4], zu[4], zv[4], zw[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) if (I > 0 || v >= c0) {          // (all-virtual registers of block 0 hold zeros)
          const int row = row_slot(I, v);
          inv_d[v] = sh.inv_diag[row];
          zu[v] = sh.z[0][row]; zv[v] = sh.z[1][row]; zw[v] = sh.z[2][row];          // (zeta / d; omega raw)
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) if (I > 0 || v >= c0) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const double val = V[t][I][v];                // (Lt^-1 k*)_row
            const double vd = val * inv_d[v];
            ssq[t] = d_fma(val, vd, ssq[t]);
            mean_u[t] = d_fma(val, zu[v], mean_u[t]);
            mean_v[t] = d_fma(val, zv[v], mean_v[t]);
            cross[t] = d_fma(vd, zw[v], cross[t]);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {     // the four lanes g = 0..3 of a column hold disjoint rows
      ssq[t] = column_sum(ssq[t]); mean_u[t] = column_sum(mean_u[t]);
      mean_v[t] = column_sum(mean_v[t]); cross[t] = column_sum(cross[t]);
    }
    BLE_SW(2);       // sums accumulated and reduced
    // ---- the bordering row (the newest observation, window entry nr): Lt_full = [Lt 0; r^T 1], r = omega / d,
    //   d_new = k_nn + noise - sum omega^2 / d,   (Lt_full^-1 b)_last = b_last - sum_i r_i (Lt^-1 b)_i
    if (kFirst && wave == 0 && g == 0 && jq < kSpecial) {
      if (jq == c_new) {              // this lane's column is omega: its sums against zeta / d are r^T zeta
        sh.last[0] = y_last_u - mean_u[0];                              // zeta_u of the newest observation
        sh.last[1] = y_last_v - mean_v[0];
        sh.last[2] = (kGpSigma2 + kGpNoise2) - ssq[0];                  // d of the new row
      } else if (jq == c_e0) {
        sh.last[3] = -cross[0];                                         // (Lt_full^-1 e_0)_last
      }
    }
    // new row of the factor and next call's drop vector, straight from the solved columns
    if constexpr (kFirst) {
      if (has_last) {
        for (int i = tid; i < nr; i += kObsBlock) sh.L[tri(nr) + i] = sh.z[2][i] * sh.inv_diag[i];
      }
      __syncthreads();
      if (has_last && tid == 0) sh.L[tri(nr) + nr] = sh.last[2];
    }
    const double inv_dn = has_last ? d_rcp(sh.last[2]) : 0.0;
    const double zl_u = has_last ? sh.last[0] * inv_dn : 0.0, zl_v = has_last ? sh.last[1] * inv_dn : 0.0;
    if (kFirst && chol_g != nullptr) {          // p = -(Lt_full^-1 e_0)[1:], and zeta / d of the whole window
      for (int i = tid; i < n_obs; i += kObsBlock) {
        if (i + 1 < n_obs) chol_g[kCholTri + i] = (i + 1 < nr) ? -sh.z[3][i + 1] : -sh.last[3];
        chol_g[kCholTri + kGpMax + i] = i < nr ? sh.z[0][i] : zl_u;
        chol_g[kCholTri + 2 * kGpMax + i] = i < nr ? sh.z[1][i] : zl_v;
      }
    }
    {
      // after the xor reductions all four lanes of a column hold the totals: lane g finishes tile g (ONE instance of
      // the tail below for all tiles of the wave: the inputs are selected by g)
      int col_m = col[0];
      double level_m = level[0], ssq_m = ssq[0], mean_u_m = mean_u[0], mean_v_m = mean_v[0], cross_m = cross[0];
#pragma unroll
      for (int t = 1; t < NT; ++t) {
        const bool mine = g == t;
        col_m = mine ? col[t] : col_m; level_m = mine ? level[t] : level_m; ssq_m = mine ? ssq[t] : ssq_m;
        mean_u_m = mine ? mean_u[t] : mean_u_m; mean_v_m = mine ? mean_v[t] : mean_v_m; cross_m = mine ? cross[t] : cross_m;
      }
      const int level_idx = lo_idx + col_m - kSpecial;
      if (g < NT && col_m >= kSpecial && level_idx <= hi_idx) {
        // the newest observation's row: K*(level, newest) = s^2 exp(-|level - p| / 326) (same x, y, t as the query)
        const double dpl = (level_m - p) * (kGpTwoKappa / 326.0);
        const double val_last = gp_exp_neg_scaled(__builtin_fabs(dpl), sh.exp2_frac) - cross_m;        // (s^2 is in the table)
        const double ss = d_fma(val_last * val_last, inv_dn, ssq_m);
        const double mu = d_fma(val_last, zl_u, mean_u_m), mv = d_fma(val_last, zl_v, mean_v_m);
        // forecast at this level from the blended column -- or, for a forecast that is not a grid (ble_observe_forecast_f32: any
        // WindField.get_forecast_column of the caller's, features.py:499-503 -> wind_gp.py:218-222), as the caller evaluated it
        int ip = (int)((level_m - 5000.0) * 1e-3);
        ip = ip > 8 ? 8 : ip;
        const double wp = (level_m - (5000.0 + 1000.0 * (double)ip)) * 1e-3;
        double fu = d_fma(wp, sh.column[(ip + 1) * 2] - sh.column[ip * 2], sh.column[ip * 2]);
        double fv = d_fma(wp, sh.column[(ip + 1) * 2 + 1] - sh.column[ip * 2 + 1], sh.column[ip * 2 + 1]);
        if (forecast_levels != nullptr) {           // (wave-uniform)
          const float* f = forecast_levels + (env * kObsLevels + level_idx) * 2;
          fu = (double)f[0]; fv = (double)f[1];
        }
        const double u = mu + fu, v = mv + fv;
        double var = kGpSigma2 - ss;
        var = var < 0.0 ? 0.0 : var;
        // (reciprocal + Newton instead of the ~30-instruction fp64 division / sqrt sequences: 2e-15 relative)
        const double deviation = n_obs > 0 ? var * (1.0 / kGpSigma2) : 0.0;     // wind_gp.py:166-168
        const double s2 = u * u + v * v;
        const double speed = s2 > 0.0 ? s2 * d_rsqrt(s2) : 0.0;
        double angle;
        if (dist < 1e-5) {
          angle = 0.0;
        } else if (speed < 1e-5) {
          angle = kPiD;
        } else {
          double c = (u * to_station_x + v * to_station_y) * d_rcp(speed + 1e-5);
          c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
          angle = kPiD / 2 - d_asin(c);
        }
        float* o = out + 16 + 3 * (pad_above + level_idx);
        o[0] = (float)deviation; o[1] = (float)(angle * (1.0 / kPiD)); o[2] = (float)(speed * d_rcp(speed + 30.0));
      }
    }
  };
  // tile T = 4 t + wave for t = 0, 1; a tile 8 + wave exists only with 127+ reachable levels (1 % of the
  // environments; 125+ while zeta still rode the sweep: 19 %): that wave sweeps it alone afterwards.
  __builtin_amdgcn_s_setprio(0);
  sweep(std::integral_constant<int, 2>{}, std::true_type{}, 0);
  if (wave + 8 < n_tiles) sweep(std::integral_constant<int, 1>{}, std::false_type{}, 8);
  BLE_SW(3);         // per-level tail done
  // padding above and below the 181 real levels, and the unreachable levels: certain, wrong way, infinitely fast
  for (int c = tid; c < kObsColumn; c += kObsBlock) {
    if (c < pad_above + lo_idx || c > pad_above + hi_idx) {
      float* o = out + 16 + 3 * c;
      o[0] = 0.0f; o[1] = 1.0f; o[2] = 1.0f;
    }
  }
  BLE_SW(4);         // padding written
  BLE_OBS_INSTR_END();
  // the factor of this window goes back to HBM for the next call
  __syncthreads();                                 // the bordering row is in LDS
  if (chol_g != nullptr) {
    const int pairs = (tri(n_obs) + 1) >> 1;       // (an odd tail stores one unused double inside the slab)
    for (int e2 = tid; e2 < pairs; e2 += kObsBlock) reinterpret_cast<double2*>(chol_g)[e2] = reinterpret_cast<const double2*>(sh.L)[e2];
    if (tid == 0) hist.n_chol[env] = n_obs;
  }
  // every lane has read the old count long before this point (barriers above)
  if (tid == 0) hist.count[env] = count;
  if (err_flags != nullptr && flags != 0) atomicOr(err_flags, flags);
}

}  // namespace ble
