// ble_physics.h -- per-lane device functions of the MI355X BLE transition kernel.
//
// One wavefront lane advances one environment.  Everything here is straight-line
// fp32 VALU code with a handful of fp64 operations where fp32 cannot hold the
// reference's result to the 1e-5 parity bar (DESIGN.md "precision map"):
//   * the NOAA ephemeris block (time-only part of solar.py:43-174) -- fp64, evaluated
//     twice per agent step and interpolated linearly over the 18 substeps;
//   * the hour-angle accumulation (solar.py:113-120) -- fp64 add, fp32 sincos;
//   * the buoyancy difference rho*V - m (balloon.py:422-427) -- fp64 subtract;
//   * the altitude-safety height compare (altitude_safety.py:103-111) and the
//     power-safety charge forecast (power_safety.py:107-115) -- fp64, once per step.
// H(p+-1)-H(p) (balloon.py:438-442) uses the cancellation-free form
//   dH = (T(p)/L) * expm1(k * log1p(d/p)),  k = -R_d L / g
// instead of differencing two 17 km heights.
//
// The file also compiles as plain C++ (g++) so that tests/emul can triage numerics on
// a machine without a GPU; that build is test tooling and is never loaded by the
// package.  Reference paths are relative to /root/reference/balloon_learning_environment/.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BLE_FN __device__ __forceinline__
#define BLE_DEVICE_BUILD 1
#else
#define BLE_FN static inline
#define BLE_DEVICE_BUILD 0
#endif

namespace ble {

// ---------------------------------------------------------------- constants
// utils/constants.py:23-30
constexpr float kGravity = 9.80665f;
constexpr double kGasConstantD = 8.3144621;
constexpr double kAirMolarMassD = 0.028964922481160;
constexpr double kHeMolarMassD = 0.004002602;
constexpr double kAirSpecificGasD = kGasConstantD / kAirMolarMassD;  // R_d
constexpr float kAirMolarOverR = (float)(kAirMolarMassD / kGasConstantD);
constexpr float kGasConstant = (float)kGasConstantD;
constexpr double kPiD = 3.14159265358979323846;
constexpr float kPi = (float)kPiD;
constexpr float kDegToRad = (float)(kPiD / 180.0);
constexpr float kRadToDeg = (float)(180.0 / kPiD);

// balloon.py:156-182 flight-vehicle constants
constexpr float kVolumeBase = 1804.0f;
constexpr float kVolumeDvDp = 0.0199f;
constexpr float kEnvelopeMass = 68.5f;
constexpr float kMaxSuperpressure = 2380.0f;
constexpr float kEnvelopeCod = 0.25f;
constexpr float kMolsLiftGas = 6830.0f;
constexpr double kDryMassD = kHeMolarMassD * 6830.0 + 68.5 + 92.5;  // He + envelope + payload [kg]
constexpr float kNightLoad = 183.7f;
constexpr float kDayLoad = 120.4f;
constexpr float kBatteryCapacity = 3058.56f;
constexpr float kStride = 10.0f;  // balloon.py:269 inner stride [s]

// control.py / balloon.py enums
enum : int { kDown = 0, kStay = 1, kUp = 2 };
enum : int { kOk = 0, kOutOfPower = 1, kBurst = 2, kZeroPressure = 3 };
enum : uint32_t { kFlagPressureRange = 1u, kFlagAbsorptivity = 2u, kFlagSolarRange = 4u,
                  kFlagPowerTable = 16u, kFlagNonFinite = 32u };

// ---------------------------------------------------------------- fast math wrappers
#if BLE_DEVICE_BUILD
BLE_FN float f_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
BLE_FN float f_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
BLE_FN float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32
BLE_FN float f_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32
BLE_FN float f_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32
BLE_FN float f_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
BLE_FN double d_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
BLE_FN double d_rint(double x) { return __builtin_rint(x); }
BLE_FN double d_sqrt(double x) { return __builtin_sqrt(x); }
#else
BLE_FN float f_exp2(float x) { return exp2f(x); }
BLE_FN float f_log2(float x) { return log2f(x); }
BLE_FN float f_rcp(float x) { return 1.0f / x; }
BLE_FN float f_sqrt(float x) { return sqrtf(x); }
BLE_FN float f_rsqrt(float x) { return 1.0f / sqrtf(x); }
BLE_FN float f_fma(float a, float b, float c) { return fmaf(a, b, c); }
BLE_FN double d_fma(double a, double b, double c) { return fma(a, b, c); }
BLE_FN double d_rint(double x) { return rint(x); }
BLE_FN double d_sqrt(double x) { return sqrt(x); }
#endif
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
BLE_FN float f_exp(float x) { return f_exp2(x * kLog2e); }
BLE_FN float f_log(float x) { return f_log2(x) * kLn2; }
BLE_FN float f_pow(float x, float y) { return f_exp2(y * f_log2(x)); }  // x > 0
BLE_FN float f_div(float a, float b) { return a * f_rcp(b); }
BLE_FN float f_min(float a, float b) { return a < b ? a : b; }
BLE_FN float f_max(float a, float b) { return a > b ? a : b; }
BLE_FN float f_clamp(float x, float lo, float hi) { return f_min(f_max(x, lo), hi); }

// log1p / expm1 for |x| <~ 1e-3 (series; next term < 1e-15 relative)
BLE_FN float log1p_small(float x) {
  return x * f_fma(x, f_fma(x, f_fma(x, -0.25f, 1.0f / 3.0f), -0.5f), 1.0f);
}
BLE_FN float expm1_small(float y) {
  return y * f_fma(y, f_fma(y, f_fma(y, 1.0f / 24.0f, 1.0f / 6.0f), 0.5f), 1.0f);
}

// sin/cos of an angle already reduced to [-pi, pi] (fp32, ~1 ulp of 1.0 absolute).
BLE_FN void sincos_reduced(float a, float* s, float* c) {
  // quadrant reduction to [-pi/4, pi/4]
  float q = rintf(a * (float)(2.0 / kPiD));
  float r = f_fma(q, -1.5707963705062866f, a);        // pi/2 hi (fp32)
  r = f_fma(q, 4.371139000186241e-08f, r);            // pi/2 lo
  float r2 = r * r;
  float sp = r * f_fma(r2, f_fma(r2, f_fma(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), 1.0f);
  float cp = f_fma(r2, f_fma(r2, f_fma(r2, f_fma(r2, 2.443315711809948e-5f, -1.388731625493765e-3f),
                                       4.166664568298827e-2f), -0.5f), 1.0f);
  int iq = (int)q & 3;
  float ss = (iq & 1) ? cp : sp;
  float cc = (iq & 1) ? sp : cp;
  *s = (iq & 2) ? -ss : ss;
  *c = ((iq + 1) & 2) ? -cc : cc;
}

// fp64 sincos for arbitrary moderate |x| (< 1e5 rad): Cody-Waite to [-pi/4, pi/4] + Taylor.
BLE_FN void sincos_f64(double x, double* s, double* c) {
  double q = d_rint(x * (2.0 / kPiD));
  double r = d_fma(q, -1.57079632679489655800e+00, x);
  r = d_fma(q, -6.12323399573676603587e-17, r);
  double r2 = r * r;
  // Taylor to r^17 / r^16 (|r| <= 0.786: truncation < 2e-18)
  double sp = -2.8114572543455206e-15;                  // -1/17!
  sp = d_fma(sp, r2, 7.647163731819816e-13);            // 1/15!
  sp = d_fma(sp, r2, -1.6059043836821613e-10);          // -1/13!
  sp = d_fma(sp, r2, 2.505210838544172e-08);            // 1/11!
  sp = d_fma(sp, r2, -2.7557319223985893e-06);          // -1/9!
  sp = d_fma(sp, r2, 1.984126984126984e-04);            // 1/7!
  sp = d_fma(sp, r2, -8.333333333333333e-03);           // -1/5!
  sp = d_fma(sp, r2, 1.6666666666666666e-01);           // 1/3!
  sp = d_fma(-sp * r2, r, r);
  double cp = 4.779477332387385e-14;                    // 1/16!
  cp = d_fma(cp, r2, -1.1470745597729725e-11);          // -1/14!
  cp = d_fma(cp, r2, 2.08767569878681e-09);             // 1/12!
  cp = d_fma(cp, r2, -2.755731922398589e-07);           // -1/10!
  cp = d_fma(cp, r2, 2.48015873015873e-05);             // 1/8!
  cp = d_fma(cp, r2, -1.3888888888888889e-03);          // -1/6!
  cp = d_fma(cp, r2, 4.1666666666666664e-02);           // 1/4!
  cp = d_fma(cp, r2, -0.5);
  cp = d_fma(cp, r2, 1.0);
  int iq = (int)((long long)q & 3);
  double ss = (iq & 1) ? cp : sp;
  double cc = (iq & 1) ? sp : cp;
  *s = (iq & 2) ? -ss : ss;
  *c = ((iq + 1) & 2) ? -cc : cc;
}

// ---------------------------------------------------------------- atmosphere
// env/balloon/standard_atmosphere.py.  One layer of the piecewise model, cached per lane.
struct AtmLayer {
  float p_base;   // P_i   (pressure at the bottom of layer i; p in (p_top, p_base])
  float p_top;    // P_{i+1}
  float t_base;   // T_i
  float lapse;    // L_i
  float h_base;   // H_i
  float k;        // -R_d L_i / g   (0 for an isothermal layer)
  float lapse_below;  // L_{i-1} (layer entered when p rises above p_base)
  float lapse_above;  // L_{i+1}
  float t_top;    // T_{i+1}
  int index;
};

BLE_FN float atm_lapse(int i, float alpha) {  // standard_atmosphere.py:68-71,83-84
  switch (i) {
    case 0: return f_fma(alpha, -0.0058f - -0.007f, -0.007f);
    case 1: return f_fma(alpha, 0.005f - 0.006f, 0.006f);
    case 2: return 0.001f;
    case 3: return 0.0028f;
    case 4: return 0.0f;
    case 5: return -0.0028f;
    default: return -0.002f;
  }
}
BLE_FN float atm_height(int i) {  // standard_atmosphere.py:66-67
  switch (i) {
    case 0: return -610.0f;
    case 1: return 17000.0f;
    case 2: return 21000.0f;
    case 3: return 32000.0f;
    case 4: return 47000.0f;
    case 5: return 51000.0f;
    case 6: return 71000.0f;
    default: return 85000.0f;
  }
}
constexpr float kROverG = (float)(kAirSpecificGasD / 9.80665);  // R_d / g

// Selects the layer containing `p` (standard_atmosphere.py:122-154 loop, transitions
// :156-183).  Walks the chain from the ground; the operating band (5-14 kPa) is layers
// 0-1, so the loop normally exits after one or two iterations.
BLE_FN AtmLayer atm_select(float alpha, float p, uint32_t* flags) {
  AtmLayer l;
  float p_base = 108870.8213f, t_base = 300.0f;
  if (!(p <= p_base)) *flags |= kFlagPressureRange;
  int i = 0;
  float lapse = atm_lapse(0, alpha), h0 = atm_height(0), h1 = atm_height(1);
  float t_top = f_fma(lapse, h1 - h0, t_base);
  float p_top;
#pragma unroll 1
  for (;;) {
    if (lapse == 0.0f)
      p_top = p_base * f_exp(-(h1 - h0) * f_rcp(kROverG * t_top));
    else
      p_top = p_base * f_pow(t_top * f_rcp(t_base), -f_rcp(kROverG * lapse));
    if (p > p_top || i == 6) break;
    ++i;
    p_base = p_top; t_base = t_top; h0 = h1; h1 = atm_height(i + 1);
    lapse = atm_lapse(i, alpha);
    t_top = f_fma(lapse, h1 - h0, t_base);
  }
  if (!(p > p_top)) *flags |= kFlagPressureRange;
  l.p_base = p_base; l.p_top = p_top; l.t_base = t_base; l.t_top = t_top; l.lapse = lapse; l.h_base = h0;
  l.k = -kROverG * lapse; l.index = i;
  l.lapse_below = atm_lapse(i > 0 ? i - 1 : 0, alpha);
  l.lapse_above = atm_lapse(i < 6 ? i + 1 : 6, alpha);
  return l;
}

// T(p) inside layer l: T_i (p/P_i)^k  (== T_i + L_i (h - H_i), standard_atmosphere.py:149-150)
BLE_FN float atm_temperature(const AtmLayer& l, float p) {
  return l.t_base * f_exp2(l.k * f_log2(p * f_rcp(l.p_base)));
}
// Height above the boundary at pressure pb (temperature tb there), for q close to pb,
// inside a layer of lapse rate L:  (tb/L) expm1(k log1p((q-pb)/pb)).
BLE_FN float atm_height_rel_boundary(float q, float pb, float tb, float lapse) {
  float lg = log1p_small((q - pb) * f_rcp(pb));
  if (lapse == 0.0f) return -kROverG * tb * lg;
  return tb * f_rcp(lapse) * expm1_small(-kROverG * lapse * lg);
}
// H(p + d) - H(p) for |d| = 1 Pa (balloon.py:438-441), t_p = T(p).
BLE_FN float atm_delta_height(const AtmLayer& l, float p, float d, float t_p) {
  float q = p + d;
  if (q > l.p_base) {  // p+1 is in the layer below (higher pressure)
    return atm_height_rel_boundary(q, l.p_base, l.t_base, l.lapse_below) -
           atm_height_rel_boundary(p, l.p_base, l.t_base, l.lapse);
  }
  if (!(q > l.p_top)) {  // p-1 is in the layer above
    return atm_height_rel_boundary(q, l.p_top, l.t_top, l.lapse_above) -
           atm_height_rel_boundary(p, l.p_top, l.t_top, l.lapse);
  }
  float lg = log1p_small(d * f_rcp(p));
  if (l.lapse == 0.0f) return -kROverG * l.t_base * lg;
  return t_p * f_rcp(l.lapse) * expm1_small(l.k * lg);
}
// Absolute height (fp32), used by probes and by the guard band of the altitude layer.
BLE_FN float atm_height_f32(const AtmLayer& l, float p, float t_p) {
  if (l.lapse == 0.0f) return f_fma(-kROverG * l.t_base, f_log(p * f_rcp(l.p_base)), l.h_base);
  return f_fma(t_p - l.t_base, f_rcp(l.lapse), l.h_base);
}

// fp64 restatement of Atmosphere.at_pressure(p).height for the altitude safety layer
// (altitude_safety.py:73-111 compares it against three thresholds; the decision must
// be bit-exact, so this follows standard_atmosphere.py:122-183 operation by operation).
#if BLE_DEVICE_BUILD
BLE_FN double d_pow(double x, double y) { return __ocml_pow_f64(x, y); }
BLE_FN double d_exp(double x) { return __ocml_exp_f64(x); }
BLE_FN double d_log(double x) { return __ocml_log_f64(x); }
#else
BLE_FN double d_pow(double x, double y) { return pow(x, y); }
BLE_FN double d_exp(double x) { return exp(x); }
BLE_FN double d_log(double x) { return log(x); }
#endif
// fp64 layer of Atmosphere (standard_atmosphere.py:122-183), operation by operation as
// the reference: used once per agent step for the altitude-safety compare (bit-exact
// decision) and as the accurate start of the incremental T(p) update in the substeps.
struct AtmLayerD {
  double p_base, p_top, t_base, lapse, h_base, h_top;
  int index;
};
BLE_FN AtmLayerD atm_select_f64(double alpha, double p) {
  const double H[8] = {-610.0, 17000.0, 21000.0, 32000.0, 47000.0, 51000.0, 71000.0, 85000.0};
  const double LO[7] = {-0.007, 0.006, 0.001, 0.0028, 0.0, -0.0028, -0.002};
  const double HI[7] = {-0.0058, 0.005, 0.001, 0.0028, 0.0, -0.0028, -0.002};
  const double g = 9.80665;
  AtmLayerD l;
  l.t_base = 300.0; l.p_base = 108870.8213; l.h_base = H[0]; l.h_top = H[1]; l.lapse = 0.0; l.p_top = 0.0; l.index = 0;
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    double lapse = (1 - alpha) * LO[i] + alpha * HI[i];
    double t_top = l.t_base + lapse * (H[i + 1] - H[i]);
    double p_top;
    if (lapse == 0.0)
      p_top = l.p_base * d_exp(-(g * (H[i + 1] - H[i])) / (kAirSpecificGasD * t_top));
    else
      p_top = l.p_base * d_pow(t_top / l.t_base, -g / (kAirSpecificGasD * lapse));
    l.lapse = lapse; l.p_top = p_top; l.h_base = H[i]; l.h_top = H[i + 1]; l.index = i;
    if (p > p_top || i == 6) break;
    l.t_base = t_top; l.p_base = p_top;
  }
  return l;
}
// height and temperature at p inside layer l (standard_atmosphere.py:135-150)
BLE_FN void atm_at_pressure_f64(const AtmLayerD& l, double p, double* height, double* temperature) {
  const double g = 9.80665;
  double h;
  if (l.lapse == 0.0)
    h = ((-kAirSpecificGasD * l.t_base / g) * d_log(p / l.p_base) + l.h_base);
  else
    h = ((d_pow(p / l.p_base, -kAirSpecificGasD * l.lapse / g) - 1) * l.t_base / l.lapse + l.h_base);
  *height = h;
  *temperature = l.t_base + l.lapse * (h - l.h_base);
}
// T(p1) from T(p0) inside one layer: T1 = T0 (p1/p0)^k, k = -R_d L / g, |p1/p0 - 1| < 2e-2.
// Series in fp64 (log1p to x^7, exp to y^6): relative error < 1e-14.
BLE_FN double atm_temperature_advance(double t0, double p0, double p1, double lapse) {
  double x = (p1 - p0) / p0;
  double lg = x * d_fma(x, d_fma(x, d_fma(x, d_fma(x, d_fma(x, d_fma(x, 1.0 / 7.0, -1.0 / 6.0), 0.2), -0.25),
                                                 1.0 / 3.0), -0.5), 1.0);
  double y = (-kAirSpecificGasD / 9.80665) * lapse * lg;
  double em1 = y * d_fma(y, d_fma(y, d_fma(y, d_fma(y, d_fma(y, 1.0 / 720.0, 1.0 / 120.0), 1.0 / 24.0), 1.0 / 6.0), 0.5), 1.0);
  return d_fma(t0, em1, t0);
}

// fp32 view of an fp64 layer (for the dH series and the probes)
BLE_FN AtmLayer atm_layer_f32(const AtmLayerD& d, float alpha) {
  AtmLayer l;
  l.p_base = (float)d.p_base; l.p_top = (float)d.p_top; l.t_base = (float)d.t_base; l.lapse = (float)d.lapse;
  l.h_base = (float)d.h_base; l.k = -kROverG * l.lapse; l.index = d.index;
  l.t_top = (float)(d.t_base + d.lapse * (d.h_top - d.h_base));
  l.lapse_below = atm_lapse(d.index > 0 ? d.index - 1 : 0, alpha);
  l.lapse_above = atm_lapse(d.index < 6 ? d.index + 1 : 6, alpha);
  return l;
}

// ---------------------------------------------------------------- safety layers
// altitude_safety.py:33-111.  fsm: 0 NOMINAL, 1 LOW, 2 VERY_LOW.
BLE_FN int altitude_safety(int action, double altitude_m, uint8_t* fsm) {
  const double min_alt = 50000.0 * 0.3048, buffer = 500.0 * 0.3048, hyst = 500.0 * 0.3048;
  int s = *fsm;
  if (altitude_m < min_alt) s = 2;
  else if (altitude_m < min_alt + buffer) s = 1;
  else if (altitude_m < min_alt + buffer + hyst) s = (s == 2 || s == 1) ? 1 : 0;
  else s = 0;
  *fsm = (uint8_t)s;
  if (s == 2) return kUp;
  if (s == 1 && action == kDown) return kStay;
  return action;
}
// envelope_safety.py:40-157. fsm: 0 NOMINAL 1 LOW_CRITICAL 2 LOW 3 HIGH 4 HIGH_CRITICAL.
// Thresholds are evaluated in fp64 on the (fp32) superpressure exactly as the reference.
BLE_FN int envelope_safety(int action, float superpressure, uint8_t* fsm) {
  const double sp = superpressure, mx = 2380.0;
  int s = *fsm;
  if (sp < 150.0) s = 1;
  else if (sp < 250.0) s = 2;
  else if (sp < 250.0 + 50.0) s = (s == 1 || s == 2) ? 2 : 0;
  else if (sp < mx - 250.0 - 50.0) s = 0;
  else if (sp < mx - 250.0) s = (s == 3 || s == 4) ? 3 : 0;
  else if (sp < mx - 150.0) s = 3;
  else s = 4;
  *fsm = (uint8_t)s;
  if (s == 1 || s == 4) return kUp;
  if ((s == 2 || s == 3) && action == kDown) return kStay;
  return action;
}
// power_safety.py:52-126.  Times in seconds relative to start_unix.
#if defined(__clang__)
#define BLE_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define BLE_NO_CONTRACT
#endif
BLE_FN int power_safety(int action, int32_t now, float battery_wh, int32_t* sunrise_h, int32_t* sunset,
                        uint8_t* paused) {
  BLE_NO_CONTRACT
  int32_t sr = *sunrise_h, ss = *sunset;
  if (now > sr) sr += ((now - sr + 86399) / 86400) * 86400;   // while now > sr: sr += 1 day
  if (now > ss) ss += ((now - ss + 86399) / 86400) * 86400;
  *sunrise_h = sr; *sunset = ss;
  const int paused_action = (action == kDown) ? kStay : action;
  const double batt = battery_wh, cap = 3058.56;
  if (ss < sr) {  // daytime
    double soc = batt / cap;
    if (*paused && soc < 0.05) return paused_action;
    *paused = 0;
    return action;
  }
  if (*paused) return paused_action;
  double hours = (double)(sr - now) / 3600.0;
  double floating_charge = 183.7 * hours;
  double expected = (batt - floating_charge) / cap;
  if (expected < 0.025) { *paused = 1; return paused_action; }
  return action;
}

// ---------------------------------------------------------------- wind field
// grid_based_wind_field.py:70-94,134-187 + scipy interpn (linear).  The reference packs
// the query as float32 and blends in fp64; here the blend is fp32 (<= 1e-6 relative to
// the corner magnitudes).  Grid layout (x, y, p, t, uv) row-major = the reference ndarray.
struct WindQuery { int ix, iy, ip, it; float wx, wy, wp, wt; };

BLE_FN void wind_axis(float q, float g0, float inv_step, float step, int n, int* idx, float* w) {
  float f = (q - g0) * inv_step;
  int i = (int)f;                 // q >= g0 after clamping
  i = i > n - 2 ? n - 2 : i;
  *idx = i;
  *w = (q - f_fma((float)i, step, g0)) * inv_step;
}
BLE_FN WindQuery wind_query(float x_m, float y_m, float pressure, int32_t elapsed_s) {
  WindQuery wq;
  // x.kilometers -> clip -> float32 (correctly rounded fp32 division == fp64 division then cast)
  float x_km = f_clamp(x_m / 1000.0f, -500.0f, 500.0f);
  float y_km = f_clamp(y_m / 1000.0f, -500.0f, 500.0f);
  float p = f_clamp(pressure, 5000.0f, 14000.0f);
  float t_h;
  if (elapsed_s < 48 * 3600) {
    t_h = (float)elapsed_s / 3600.0f;
  } else {  // _boomerang(t, 48): fp64 like the reference, then float32
    double t = (double)elapsed_s / 3600.0;
    long long cyc = (long long)(t / 48.0);
    double rem = t - 48.0 * (double)cyc;   // exact for these magnitudes
    if (rem < 0.0) rem += 48.0;
    if (rem >= 48.0) rem -= 48.0;
    t_h = (float)((cyc & 1) ? 48.0 - rem : rem);
  }
  wind_axis(x_km, -500.0f, 1.0f / 50.0f, 50.0f, 21, &wq.ix, &wq.wx);
  wind_axis(y_km, -500.0f, 1.0f / 50.0f, 50.0f, 21, &wq.iy, &wq.wy);
  wind_axis(p, 5000.0f, 1.0f / 1000.0f, 1000.0f, 10, &wq.ip, &wq.wp);
  wind_axis(t_h, 0.0f, 1.0f / 6.0f, 6.0f, 9, &wq.it, &wq.wt);
  return wq;
}
// 16-corner blend.  Each (x, y, p) corner is 4 contiguous floats (t, t+1) x (u, v).
BLE_FN void wind_blend(const float* __restrict__ grid, const WindQuery& wq, float* u, float* v) {
  float au = 0.0f, av = 0.0f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* cell = grid + ((((wq.ix + a) * 21 + (wq.iy + b)) * 10 + (wq.ip + c)) * 9 + wq.it) * 2;
        float w3 = (a ? wq.wx : 1.0f - wq.wx) * (b ? wq.wy : 1.0f - wq.wy) * (c ? wq.wp : 1.0f - wq.wp);
        float w0 = w3 * (1.0f - wq.wt), w1 = w3 * wq.wt;
        float u0 = cell[0], v0 = cell[1], u1 = cell[2], v1 = cell[3];
        au = f_fma(u0, w0, au); av = f_fma(v0, w0, av);
        au = f_fma(u1, w1, au); av = f_fma(v1, w1, av);
      }
    }
  }
  *u = au; *v = av;
}

// ---------------------------------------------------------------- solar
// Time-only part of solar.solar_calculator (solar.py:65-134,170-172), fp64.
struct Ephemeris {
  double eot_min;    // degrees(equation_of_time): minutes of time (solar.py:107-115)
  double sin_decl, cos_decl;
  double flux;       // W/m^2
};
BLE_FN Ephemeris ephemeris(int64_t unix_s) {
  // solar.py:66-79.  julian_day_number + fraction_of_day == 2440587.5 + unix_s / 86400
  // (exact identity for the Gregorian formula at :71-75; verified against the oracle).
  int64_t days = unix_s / 86400;
  int64_t sod = unix_s - days * 86400;
  if (sod < 0) { sod += 86400; days -= 1; }
  double julian_time = (2440587.5 + (double)days) + (double)sod / 86400.0;
  double jc = (julian_time - 2451545.0) / 36525.0;
  const double d2r = kPiD / 180.0;
  double l0 = d2r * (280.46646 + jc * (36000.76983 + jc * 0.0003032));
  double s2l, c2l;
  sincos_f64(2.0 * l0, &s2l, &c2l);
  double s4l = 2.0 * s2l * c2l;
  double m0 = d2r * (357.52911 + jc * (35999.05029 - 0.0001537 * jc));
  double sm, cm;
  sincos_f64(m0, &sm, &cm);
  double s2m = 2.0 * sm * cm;
  double s3m = sm * (3.0 - 4.0 * sm * sm);
  double mean_obl = d2r * (23.0 + (26.0 + ((21.448 - jc * (46.815 + jc * (0.00059 - jc * 0.001813)))) / 60.0) / 60.0);
  double so, co;
  sincos_f64(d2r * (125.04 - 1934.136 * jc), &so, &co);
  double obl = mean_obl + d2r * (0.00256 * co);
  double sobl, cobl;
  sincos_f64(obl, &sobl, &cobl);
  double tan_half = sobl / (1.0 + cobl);
  double var_y = tan_half * tan_half;
  double ecc = 0.016708634 - jc * (0.000042037 + 0.0000001267 * jc);
  double eot = 4.0 * (var_y * s2l - 2.0 * ecc * sm + 4.0 * ecc * var_y * sm * c2l - 0.5 * var_y * var_y * s4l -
                      1.25 * ecc * ecc * s2m);
  double eoc = d2r * (sm * (1.914602 - jc * (0.004817 + 0.000014 * jc)) + s2m * (0.019993 - 0.000101 * jc) +
                      s3m * 0.000289);
  double app = l0 + eoc - d2r * (0.00569 - 0.00478 * so);
  double sa, ca;
  sincos_f64(app, &sa, &ca);
  Ephemeris e;
  e.eot_min = eot * (180.0 / kPiD);
  e.sin_decl = sobl * sa;
  e.cos_decl = d_sqrt(1.0 - e.sin_decl * e.sin_decl);
  double r = (1 + ecc) / (1 - ecc);
  e.flux = 1366.0 * (1 + 0.5 * (r * r - 1) * cm);
  return e;
}

// 1 - sin(uncorrected elevation) at one instant, fp64.
// BalloonState.latlng (balloon.py:217-220 -> spherical_geometry.py:44-76) folded into the
// zenith formula (solar.py:136-139); no atan2/asin/acos:
//   heading = atan2(x, y)            -> cos/sin(heading) = y/d, x/d
//   d_lng   = atan2(yy, xx)          -> cos/sin(B + d_lng) by rotating (cos B, sin B)
//   hour_angle = (B + d_lng) -+ 180  -> cos(hour_angle) = -cos(B + d_lng)   (solar.py:113-120)
// b_deg = 360 frac_day + eot/4 + lng0 [deg].
// Within one agent step x, y move linearly (constant wind) and the ephemeris is linear,
// so S(t) is smooth: agent_step evaluates this at 3 nodes and interpolates quadratically
// (|error| <= |d3S/dt3| h^3 * 0.064 <= 1.8e-8, and ~0 around local noon where d3S/dt3 -> 0).
struct SunSC { float sin_el, cos_el; };
BLE_FN double sun_one_minus_sin_f64(double sin_lat0, double cos_lat0, double x, double y, double b_deg,
                                    double sin_decl, double cos_decl) {
  double d2 = x * x + y * y;
  double d = d_sqrt(d2);
  double cos_h = 1.0, sin_h = 0.0;
  if (d2 > 0.0) { cos_h = y / d; sin_h = x / d; }
  double sin_a, cos_a;
  sincos_f64(d * (1.0 / 6371000.0), &sin_a, &cos_a);
  double sin_lat = cos_a * sin_lat0 + sin_a * cos_lat0 * cos_h;
  double cos_lat = d_sqrt(1.0 - sin_lat * sin_lat);
  double yy = sin_a * cos_lat0 * sin_h;
  double xx = cos_a - sin_lat0 * sin_lat;
  double sin_b, cos_b;
  sincos_f64(b_deg * (kPiD / 180.0), &sin_b, &cos_b);
  double cos_bl = (cos_b * xx - sin_b * yy) / d_sqrt(xx * xx + yy * yy);
  double s = sin_lat * sin_decl - cos_lat * cos_decl * cos_bl;
  return 1.0 - s;
}
BLE_FN SunSC sun_from_one_minus_sin(float oms) {
  oms = f_clamp(oms, 0.0f, 2.0f);
  SunSC r;
  r.sin_el = 1.0f - oms;
  r.cos_el = f_sqrt(oms * (2.0f - oms));
  return r;
}

// Atmospheric refraction (solar.py:141-157) applied as a small rotation of (S, C).
BLE_FN SunSC sun_refract(SunSC unc) {
  const float kSin85 = 0.99619472027f, kSin5 = 0.08715574443f, kSinM0575 = -0.01003547478f;
  const float s = unc.sin_el, c = unc.cos_el;
  float refr;  // arcseconds
  if (s > kSin85) {
    refr = 0.0f;
  } else if (s > kSin5) {
    float ct = c * f_rcp(s), ct2 = ct * ct;
    refr = ct * f_fma(ct2, f_fma(ct2, 0.000086f, -0.07f), 58.1f);
  } else if (s > kSinM0575) {
    float s2 = s * s;
    float e = kRadToDeg * s * f_fma(s2, f_fma(s2, f_fma(s2, 15.0f / 336.0f, 3.0f / 40.0f), 1.0f / 6.0f), 1.0f);
    refr = f_fma(e, f_fma(e, f_fma(e, f_fma(e, 0.711f, -12.79f), 103.4f), -518.2f), 1735.0f);
  } else {
    refr = -20.772f * c * f_rcp(s);
  }
  float dl = refr * (kDegToRad / 3600.0f);
  float d2 = dl * dl;
  float sd = dl * f_fma(d2, f_fma(d2, 1.0f / 120.0f, -1.0f / 6.0f), 1.0f);
  float cd = f_fma(d2, f_fma(d2, 1.0f / 24.0f, -0.5f), 1.0f);
  SunSC r;
  r.sin_el = f_fma(s, cd, c * sd);
  r.cos_el = f_fma(c, cd, -s * sd);
  return r;
}

constexpr float kSinMinSolarEl = -0.07396924496f;  // sin(-4.242 deg), solar.py:38

// solar_atmospheric_attenuation (solar.py:177-209) from sin(el).
BLE_FN float solar_attenuation(float sin_el, float pressure, uint32_t* flags) {
  if (pressure > 101325.0f || pressure < 0.0f) *flags |= kFlagSolarRange;
  if (sin_el < kSinMinSolarEl) return 0.0f;
  float t = 614.0f * sin_el;
  float root = f_sqrt(f_fma(t, t, 1229.0f));
  float diff = t > 0.0f ? 1229.0f * f_rcp(root + t) : root - t;   // sqrt(1229+t^2) - t without cancellation
  float airmass = 0.34764f * (pressure * (1.0f / 101325.0f)) * diff;
  return 0.5f * (f_exp(-0.65f * airmass) + f_exp(-0.95f * airmass));
}
// solar_power (solar.py:515-536) with balloon_shadow (:212-236) folded in.
BLE_FN float solar_power(float sin_el, float cos_el, float attenuation) {
  const float kCos35 = 0.81915204429f, kSin35 = 0.57357643635f;
  const float kCos65 = 0.42261826174f, kSin65 = 0.90630778704f;
  // shadow_el = degrees(atan2(sqrt(h (10.41603 + h)), 8.69275)), h = 3.3 / 2.7
  const float kSinShadow33 = 0.61205375195f;  // sin(37.738149 deg)
  const float kSinShadow27 = 0.56489306688f;  // sin(34.394865 deg)
  float sh33 = sin_el >= kSinShadow33 ? 0.4392f : 1.0f;
  float sh27 = sin_el >= kSinShadow27 ? 0.4392f : 1.0f;
  float c35 = f_fma(cos_el, kCos35, sin_el * kSin35);
  float c65 = f_fma(cos_el, kCos65, sin_el * kSin65);
  return 210.0f * attenuation * f_fma(4.0f * c35, sh33, 2.0f * c65 * sh27);
}

// ---------------------------------------------------------------- thermal
// thermal.py:52-230.
constexpr float kStefanBoltzmann = 0.000000056704f;
BLE_FN float absorptivity_ir(float t) { return f_fma(0.000232f, t - 210.0f, 0.04587f); }
BLE_FN float total_absorptivity(float a, uint32_t* flags) {   // reflectivity 0.0291
  float f = a * f_fma(1.0f - a - 0.0291f, 1.0f / (1.0f - 0.0291f), 1.0f);
  if (f < 0.0f || f > 1.0f) *flags |= kFlagAbsorptivity;
  return f;
}
constexpr float kSolarAbsorptivityTotal =
    0.01435f * (1.0f + (1.0f - 0.01435f - 0.0291f) / (1.0f - 0.0291f));
// Earth-IR heat per unit balloon area (thermal.py:209-213): constant over an episode.
BLE_FN float earth_heat_per_area(float upwelling_ir, uint32_t* flags) {
  float t_bb = f_sqrt(f_sqrt(upwelling_ir * (1.0f / kStefanBoltzmann)));
  return upwelling_ir * 0.4605f * total_absorptivity(absorptivity_ir(t_bb), flags);
}
// d_balloon_temperature_dt (thermal.py:175-230).  v23 = V^(2/3); rho = air density at (p, T_amb).
BLE_FN float thermal_dtdt(float v23, float t_int, float t_amb, float rho, float solar_flux_att,
                          float q_earth_per_area, uint32_t* flags) {
  const float kR2 = 0.38483473659f;         // (3 / (4 pi))^(2/3)
  float r2 = kR2 * v23;                   // radius^2
  float radius = f_sqrt(r2);
  float area = 4.0f * kPi * r2;
  float q_solar = solar_flux_att * 0.25f * kSolarAbsorptivityTotal;           // per area
  float t2 = t_int * t_int;
  float q_emit = kStefanBoltzmann * t2 * t2 * total_absorptivity(absorptivity_ir(t_int), flags);
  // convective_heat_air_factor (thermal.py:150-172)
  float viscosity = 1.458e-6f * t_amb * f_sqrt(t_amb) * f_rcp(t_amb + 110.4f);
  float conductivity = 0.0241f * f_pow(t_amb * (1.0f / 273.15f), 0.9f);
  float prandtl = f_fma(-3.25e-4f, t_amb, 0.804f);
  float dia = 2.0f * radius;
  float dt = t_amb - t_int;
  float rv = rho * f_rcp(viscosity);
  float grashof = 9.80665f * rv * rv * (dia * dia * dia) * f_rcp(t_amb) * fabsf(dt);
  float rayleigh = prandtl * grashof;
  float nusselt = 2.0f + 0.457f * f_sqrt(f_sqrt(rayleigh)) + f_pow(f_fma(2.69e-8f, rayleigh, 1.0f), 1.0f / 12.0f);
  float q_conv = nusselt * conductivity * f_rcp(dia) * dt;                    // per area
  return area * (q_solar + q_earth_per_area + q_conv - q_emit) * (1.0f / (1500.0f * kEnvelopeMass));
}

// ---------------------------------------------------------------- envelope
// calculate_superpressure_and_volume (balloon.py:552-609)
BLE_FN void superpressure_volume(float mols_air, float t_int, float p, float* volume, float* sp) {
  float vu = (kMolsLiftGas + mols_air) * kGasConstant * t_int * f_rcp(p);
  if (vu <= kVolumeBase) { *volume = vu; *sp = 0.0f; return; }
  float b = -(kVolumeBase - kVolumeDvDp * p);
  float c4 = 4.0f * kVolumeDvDp * vu * p;                 // -4c
  float v = 0.5f * (f_sqrt(f_fma(b, b, c4)) - b);
  *volume = v;
  *sp = p * (vu - v) * f_rcp(v);                          // p vu / v - p
}

// fp64 variant for the vertical-dynamics chain (see ble_step_core.h): the buoyancy
// difference rho V - m is an unstable map near float equilibrium, so V must be good to ~1e-9.
BLE_FN void superpressure_volume_f64(double mols_air, double t_int, double p, double* volume, double* sp) {
  double vu = ((6830.0 + mols_air) * kGasConstantD * t_int / p);
  if (vu <= 1804.0) { *volume = vu; *sp = 0.0; return; }
  double b = -(1804.0 - 0.0199 * p);
  double c = -(0.0199 * vu * p);
  double v = 0.5 * (-b + d_sqrt(b * b - 4 * c));
  *volume = v;
  *sp = (p * vu / v - p);
}

// ---------------------------------------------------------------- ACS
// acs.py:24-68.  prm1 = pressure_ratio - 1.
BLE_FN float acs_power(float prm1) {
  // interp1d([1.0,1.05,1.2,1.25,1.35] -> [100,100,300,400,400], extrapolate): flat end segments
  if (prm1 <= 0.05f) return 100.0f;
  if (prm1 <= 0.2f) return f_fma(prm1 - 0.05f, 200.0f / 0.15f, 100.0f);
  if (prm1 <= 0.25f) return f_fma(prm1 - 0.2f, 100.0f / 0.05f, 300.0f);
  return 400.0f;
}
#if BLE_DEVICE_BUILD
__device__ __constant__
#else
static
#endif
const float kAcsEfficiency[4][13] = {
    {0.4f, 0.4f, 0.3f, 0.2f, 0.2f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f},
    {0.4f, 0.3f, 0.3f, 0.30f, 0.25f, 0.23f, 0.20f, 0.15f, 0.12f, 0.10f, 0.0f, 0.0f, 0.0f},
    {0.0f, 0.3f, 0.25f, 0.25f, 0.25f, 0.20f, 0.20f, 0.20f, 0.2f, 0.15f, 0.13f, 0.12f, 0.11f},
    {0.0f, 0.23f, 0.23f, 0.23f, 0.23f, 0.23f, 0.20f, 0.20f, 0.20f, 0.18f, 0.16f, 0.15f, 0.13f}};
BLE_FN float acs_efficiency(float prm1, float power) {
  float fx = f_clamp((prm1 - 0.05f) * 40.0f, 0.0f, 12.0f);     // 13 nodes, step 0.025
  float fy = f_clamp((power - 100.0f) * 0.01f, 0.0f, 3.0f);    // 4 nodes, step 100 W
  int ix = (int)fx; ix = ix > 11 ? 11 : ix;
  int iy = (int)fy; iy = iy > 2 ? 2 : iy;
  float wx = fx - (float)ix, wy = fy - (float)iy;
  float z00 = kAcsEfficiency[iy][ix], z01 = kAcsEfficiency[iy][ix + 1];
  float z10 = kAcsEfficiency[iy + 1][ix], z11 = kAcsEfficiency[iy + 1][ix + 1];
  float lo = f_fma(wx, z01 - z00, z00), hi = f_fma(wx, z11 - z10, z10);
  return f_fma(wy, hi - lo, lo);
}

// power_table.lookup (power_table.py:21-38)
BLE_FN float power_table_lookup(float pr, float soc, uint32_t* flags) {
  if (!(pr >= 0.99f && pr <= 5.0f)) *flags |= kFlagPowerTable;
  const double prd = pr, s = soc;  // compare against the reference's double literals
  int i = (prd >= 1.08) + (prd >= 1.11) + (prd >= 1.14) + (prd >= 1.17) + (prd >= 1.2) + (prd >= 1.23) + (prd >= 1.26);
  double e0, e1, e2; float w1, w2, w3;
  switch (i) {
    case 0: e0 = 0.3; e1 = 0.4; e2 = 0.5; w1 = 150; w2 = 175; w3 = 200; break;
    case 1: e0 = 0.3; e1 = 0.4; e2 = 0.7; w1 = 200; w2 = 200; w3 = 225; break;
    case 2: e0 = 0.3; e1 = 0.4; e2 = 0.6; w1 = 225; w2 = 225; w3 = 250; break;
    case 3: e0 = 0.3; e1 = 0.4; e2 = 0.5; w1 = 200; w2 = 225; w3 = 250; break;
    case 4: e0 = 0.3; e1 = 0.4; e2 = 0.5; w1 = 225; w2 = 250; w3 = 275; break;
    case 5: e0 = 0.4; e1 = 0.5; e2 = 2.0; w1 = 275; w2 = 300; w3 = 300; break;
    case 6: e0 = 0.5; e1 = 0.6; e2 = 2.0; w1 = 300; w2 = 325; w3 = 325; break;
    default: e0 = 0.5; e1 = 0.6; e2 = 2.0; w1 = 325; w2 = 350; w3 = 350; break;
  }
  int j = (s >= e0) + (s >= e1) + (s >= e2);
  return j == 0 ? 0.0f : (j == 1 ? w1 : (j == 2 ? w2 : w3));
}

// ---------------------------------------------------------------- reward
// perciatelli_reward_function (env/balloon_env.py:44-102), base term.
BLE_FN float reward_distance(float x, float y) {
  float d = f_sqrt(f_fma(x, x, y * y));
  if (d <= 50000.0f) return 1.0f;
  return 0.4f * f_exp((-0.69314718056f / 100.0f) * ((d - 50000.0f) * 0.001f));
}

}  // namespace ble
