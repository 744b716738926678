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
// Reference paths are relative to /root/reference/balloon_learning_environment/.
#pragma once
// BLE_FN, BLE_CONST_TABLE and the thin intrinsic layer (f_exp2, f_log2, f_rcp, ..., d_rcp_seed, d_ldexp, d_min,
// d_max): ble_intrinsics.h maps them onto gfx950 instructions.  (tests/emul force-includes a libm-based
// stand-in with the same include guard to triage numerics on a machine without a GPU; nothing host-side
// lives in this directory.)
#include "ble_intrinsics.h"

// (profiling builds count how often the rare paths of the transition are taken: profiles/instr/ble_step_instr.h)
#ifndef BLE_STEP_EVENT
#define BLE_STEP_EVENT(i) do {} while (0)
#endif
#ifndef BLE_STEP_TICK
#define BLE_STEP_TICK(i) do {} while (0)
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

// BalloonState's flight-vehicle constants (balloon.py:156-173,183,200; include/ble_abi.h::ble_vehicle) in the DERIVED forms the lane
// functions evaluate.  Two carriers with the same member names: VehicleDefault -- the reference's defaults as compile-time constants, what
// every kernel flew before ABI 5 and what ble_state_f32.vehicle == NULL still selects: the same literals, the same folded arithmetic, the
// same bits --, and VehicleRt -- the same quantities derived on the HOST in double (make_vehicle_rt, ble_kernels.hip) and handed to a second
// instantiation of the kernels as an argument, i.e. in scalar registers.  The lane functions take the quantities as trailing parameters
// whose defaults are VehicleDefault's.
struct VehicleDefault {
  static constexpr double v0 = 1804.0, dvdp = 0.0199, four_dvdp = 4.0 * 0.0199, inv_dvdp = 1.0 / 0.0199;   // superpressure_volume_f64
  static constexpr double inv_cbrt_v0 = 0.08214626507693945;                                                  // 1804^(-1/3): the cold start
  static constexpr double lift = 6830.0, dry_mass = kDryMassD;                                                // mols of helium; He + envelope + payload [kg]
  static constexpr double envelope_mass = 68.5, payload_mass = 92.5, he_mass = kHeMolarMassD * 6830.0;       // the cold start subtracts them one by one
  static constexpr double max_sp = 2380.0;                                                                     // burst threshold, EnvelopeSafetyLayer
  static constexpr double drag_arg = 8.0 * 9.80665 * (kGasConstantD / kAirMolarMassD);                        // 2 g (R / M) / cod, cod = 0.25
  static constexpr double thermal_scale = 10.0 * 4.0 * kPiD * 0.38483473658887897 / (1500.0 * 68.5);          // 10 s x 4 pi (3 / (4 pi))^(2/3) / (c_p m_envelope)
  static constexpr double valve_k = -0.62 * (kPiD * 0.04 * 0.04 / 4.0);                                        // -C_d x the valve's area
  static constexpr double night_load_d = 183.7, capacity_d = 3058.56, day_load_d = 120.4;
  static constexpr float day_load = 120.4f, night_load = 183.7f, capacity = 3058.56f;
  static constexpr bool power_layer = true;                                                                    // power_safety_layer_enabled
  // the observation (csrc/ble_observe.h): battery_soc's reciprocal capacity, get_pressure_range's float ceiling p / T (empty mass x R / (M V0),
  // pressure_range_builder.py:236-245) and its upper superpressure bound (max - BUFFER, :224-228)
  static constexpr double inv_capacity = 1.0 / 3058.56;
  static constexpr double ceiling_target = (92.5 + 68.5 + 6830.0 * kHeMolarMassD) * kGasConstantD / (kAirMolarMassD * 1804.0);
  static constexpr double sp_hi = 2380.0 - 250.0;
};
struct VehicleRt {
  double v0, dvdp, four_dvdp, inv_dvdp, inv_cbrt_v0, lift, dry_mass, envelope_mass, payload_mass, he_mass, max_sp, drag_arg, thermal_scale, valve_k;
  double night_load_d, capacity_d, day_load_d;
  float day_load, night_load, capacity;
  bool power_layer;
  double inv_capacity, ceiling_target, sp_hi;
};

// control.py / balloon.py enums
enum : int { kDown = 0, kStay = 1, kUp = 2 };
enum : int { kOk = 0, kOutOfPower = 1, kBurst = 2, kZeroPressure = 3 };
enum : uint32_t { kFlagPressureRange = 1u, kFlagAbsorptivity = 2u, kFlagSolarRange = 4u,
                  kFlagPowerTable = 16u, kFlagNonFinite = 32u };

// ---------------------------------------------------------------- fp64 helpers over the intrinsic layer
// fp64 reciprocal / reciprocal-sqrt: hardware seed (v_rcp_f64 / v_rsq_f64, measured 4.3e-8 /
// 5.0e-8 relative on gfx950) + ONE Newton step -> ~2e-15 / 4e-15 relative.  That is five
// orders below what the vertical chain needs (1e-10) and avoids both the second step and the
// v_div_scale/v_div_fixup ladder (inputs here are normal, positive, far from overflow).
// x / b for a CONSTANT b, correctly rounded like the division instruction sequence (which is ~12 issue slots in fp32, ~30 in fp64):
// q0 = RN(x rb), rb = RN(1 / b); the remainder r = x - q0 b is exact in an fma; RN(q0 + r rb) is the correctly rounded quotient
// (Markstein's final division step) while the quotient is a normal number -- a subnormal one (|x_m| < 1e-35 m) may sit one subnormal
// step off, which the forecast's `x_km + 500` cannot see.  Checked against the true division exhaustively over the ranges the callers
// use: tests/test_kernel_numerics_host.py::test_constant_divisions_are_correctly_rounded.
BLE_FN float f_div_const(float x, float b, float rb) {
  BLE_NO_CONTRACT
  const float q0 = x * rb;
  return f_fma(f_fma(-q0, b, x), rb, q0);
}
BLE_FN double d_div_const(double x, double b, double rb) {
  BLE_NO_CONTRACT
  const double q0 = x * rb;
  return d_fma(d_fma(-q0, b, x), rb, q0);
}
BLE_FN double d_rcp(double x) {
  double r = d_rcp_seed(x);
  return d_fma(d_fma(-x, r, 1.0), r, r);
}
BLE_FN double d_rsqrt(double x) {
  double y = d_rsq_seed(x);
  return y * d_fma(-0.5 * x * y, y, 1.5);
}
// sqrt(x) = x rsqrt(x): 4e-15 relative, three instructions fewer than d_sqrt_fast's corrected form -- enough for the
// vertical chain (needs ~1e-10 after the map's amplification)
BLE_FN double d_sqrt_rs(double x) { return x * d_rsqrt(x); }   // x > 0
BLE_FN double d_sqrt_fast(double x) {   // x > 0
  double y = d_rsqrt(x);
  double sq = x * y;
  return d_fma(0.5 * y, d_fma(-sq, sq, x), sq);
}
// thermal.py's constants in the folded form thermal_increment_f64 evaluates (see there)
constexpr double kThermalMu = 1.458e-6 / (0.028964922481160 / 8.3144621);          // 1.458e-6 R / M
constexpr double kRayleighScale = 9.80665 * (1.0 / kThermalMu) * (1.0 / kThermalMu) * (6.0 / 3.14159265358979323846);
constexpr double kRayleighT = -3.25e-4 * kRayleighScale, kRayleighT0 = 0.804 * kRayleighScale;     // Prandtl(T) x the scale
constexpr double kEmitRho = 1.0 / (1.0 - 0.0291), kEmitC1 = 0.000232, kEmitC0 = 0.04587 - 0.000232 * 210.0;
constexpr double kEmitA = 0.000000056704 * (-kEmitRho * kEmitC1 * kEmitC1);
constexpr double kEmitB = 0.000000056704 * (2.0 * kEmitC1 - 2.0 * kEmitRho * kEmitC1 * kEmitC0);
constexpr double kEmitC = 0.000000056704 * (2.0 * kEmitC0 - kEmitRho * kEmitC0 * kEmitC0);
constexpr double kCondTenth = 0.1 * (0.0241 * 0.006415624181362592) / (2.0 * 0.62035049089940009);   // 0.1 x k(273.15 K) / 273.15^0.9 / (2 (3/(4 pi))^(1/3))
// The non-inline fp64 constants of a stride's right-hand sides that are ADDENDS of an fma (Horner coefficients, offsets).  The stride
// loops of the transition kernels make them once per agent step as opaque register pairs (stride_k_vreg, see d_vreg); every other caller
// passes the literals (stride_k_literal: the default argument) -- the same values, the same arithmetic.
struct StrideK {
  double ra_t0, five, eleven, emit_b, emit_c;          // thermal_increment_f64
  double dry_mass;                                     // stride_pressure
  double lg5, lg4, lg3, ex4, ex3;                      // atm_temperature_advance
  double acs_x0, forty;                                // acs_down_poly
  // ... and multiplicands / offsets whose scalar-register pairs the stride loop had to re-assemble (more constants than scalar registers)
  double t110, cond_tenth;                             // thermal_increment_f64
  double m_over_r, ten;                                // stride_pressure
  double lg6, ex5;                                     // atm_temperature_advance
  double lift, v0;                                     // superpressure_volume_f64
};
BLE_FN StrideK stride_k_literal(double dry_mass = VehicleDefault::dry_mass, double lift = VehicleDefault::lift, double v0 = VehicleDefault::v0) {
  StrideK k;
  k.ra_t0 = kRayleighT0; k.five = 5.0; k.eleven = 11.0; k.emit_b = kEmitB; k.emit_c = kEmitC;
  k.dry_mass = dry_mass;
  k.lg5 = 0.2; k.lg4 = -0.25; k.lg3 = 1.0 / 3.0; k.ex4 = 1.0 / 24.0; k.ex3 = 1.0 / 6.0;
  k.acs_x0 = 0.05; k.forty = 40.0;
  k.t110 = 110.4; k.cond_tenth = kCondTenth;
  k.m_over_r = kAirMolarMassD / kGasConstantD; k.ten = 10.0;
  k.lg6 = -1.0 / 6.0; k.ex5 = 1.0 / 120.0;
  k.lift = lift; k.v0 = v0;
  return k;
}
BLE_FN StrideK stride_k_vreg(double dry_mass = VehicleDefault::dry_mass, double lift = VehicleDefault::lift, double v0 = VehicleDefault::v0) {
  StrideK k = stride_k_literal(dry_mass, lift, v0);
  k.ra_t0 = d_vreg(k.ra_t0); k.five = d_vreg(k.five); k.eleven = d_vreg(k.eleven); k.emit_b = d_vreg(k.emit_b); k.emit_c = d_vreg(k.emit_c);
  k.dry_mass = d_vreg(k.dry_mass);
  k.lg5 = d_vreg(k.lg5); k.lg4 = d_vreg(k.lg4); k.lg3 = d_vreg(k.lg3); k.ex4 = d_vreg(k.ex4); k.ex3 = d_vreg(k.ex3);
  k.acs_x0 = d_vreg(k.acs_x0); k.forty = d_vreg(k.forty);
  k.t110 = d_vreg(k.t110); k.cond_tenth = d_vreg(k.cond_tenth);
  k.m_over_r = d_vreg(k.m_over_r); k.ten = d_vreg(k.ten);
  k.lg6 = d_vreg(k.lg6); k.ex5 = d_vreg(k.ex5);
  k.lift = d_vreg(k.lift); k.v0 = d_vreg(k.v0);
  return k;
}
// ln(x), x > 0 normal: x = m 2^e, m in [sqrt(.5), sqrt(2)); ln m = 2 atanh((m-1)/(m+1)).
BLE_FN double d_log_fast(double x) {
  double m = d_frexp_mant(x);
  int e = d_frexp_exp(x);
  if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
  double sq = (m - 1.0) * d_rcp(m + 1.0);
  double z = sq * sq;
  double pl = 1.0 / 23.0;
  pl = d_fma(pl, z, 1.0 / 21.0); pl = d_fma(pl, z, 1.0 / 19.0); pl = d_fma(pl, z, 1.0 / 17.0);
  pl = d_fma(pl, z, 1.0 / 15.0); pl = d_fma(pl, z, 1.0 / 13.0); pl = d_fma(pl, z, 1.0 / 11.0);
  pl = d_fma(pl, z, 1.0 / 9.0); pl = d_fma(pl, z, 1.0 / 7.0); pl = d_fma(pl, z, 0.2);
  pl = d_fma(pl, z, 1.0 / 3.0);
  double lm = d_fma(2.0 * sq * z, pl, 2.0 * sq);
  double de = (double)e;
  return d_fma(de, 6.93147180369123816490e-01, d_fma(de, 1.90821492927058770002e-10, lm));
}
// exp(t), |t| < 700
BLE_FN double d_exp_fast(double t) {
  double kq = d_rint(t * 1.44269504088896338700e+00);
  double r = d_fma(kq, -6.93147180369123816490e-01, t);
  r = d_fma(kq, -1.90821492927058770002e-10, r);
  double pe = 1.0 / 6227020800.0;                                   // 1/13!
  pe = d_fma(pe, r, 1.0 / 479001600.0); pe = d_fma(pe, r, 1.0 / 39916800.0); pe = d_fma(pe, r, 1.0 / 3628800.0);
  pe = d_fma(pe, r, 1.0 / 362880.0); pe = d_fma(pe, r, 1.0 / 40320.0); pe = d_fma(pe, r, 1.0 / 5040.0);
  pe = d_fma(pe, r, 1.0 / 720.0); pe = d_fma(pe, r, 1.0 / 120.0); pe = d_fma(pe, r, 1.0 / 24.0);
  pe = d_fma(pe, r, 1.0 / 6.0); pe = d_fma(pe, r, 0.5); pe = d_fma(pe, r, 1.0); pe = d_fma(pe, r, 1.0);
  return d_ldexp(pe, (int)kq);
}
BLE_FN double d_pow_fast(double x, double y) { return d_exp_fast(y * d_log_fast(x)); }

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
  const float lg = log1p_small((q - pb) * f_rcp(pb));
  const bool iso = lapse == 0.0f;
  const float inv_l = f_rcp(iso ? 1.0f : lapse);
  return iso ? -kROverG * tb * lg : tb * inv_l * expm1_small(-kROverG * lapse * lg);
}
// H(p + d) - H(p) for |d| = 1 Pa (balloon.py:438-441), t_p = T(p).
BLE_FN float atm_delta_height(const AtmLayer& l, float p, float d, float t_p) {
  const float q = p + d;
  const float lg = log1p_small(d * f_rcp(p));
  const bool iso = l.lapse == 0.0f;
  const float inv_l = f_rcp(iso ? 1.0f : l.lapse);
  float dh = iso ? -kROverG * t_p * lg : t_p * inv_l * expm1_small(l.k * lg);
  // p and p+-1 on different sides of a layer transition (rare: only within 1 Pa of it):
  // both heights are measured from the transition they straddle.
  const bool below = q > l.p_base;          // p+1 is in the layer below (higher pressure)
  const bool above = !(q > l.p_top);        // p-1 is in the layer above
  if (__builtin_expect(below || above, 0)) {
    const float pb = below ? l.p_base : l.p_top;
    const float tb = below ? l.t_base : l.t_top;
    const float lapse_q = below ? l.lapse_below : l.lapse_above;
    dh = atm_height_rel_boundary(q, pb, tb, lapse_q) - atm_height_rel_boundary(p, pb, tb, l.lapse);
  }
  return dh;
}
// Absolute height (fp32), used by probes and by the guard band of the altitude layer.
BLE_FN float atm_height_f32(const AtmLayer& l, float p, float t_p) {
  if (l.lapse == 0.0f) return f_fma(-kROverG * l.t_base, f_log(p * f_rcp(l.p_base)), l.h_base);
  return f_fma(t_p - l.t_base, f_rcp(l.lapse), l.h_base);
}

// fp64 atmosphere window (standard_atmosphere.py:122-183).  Built once per agent step around
// the layer i0 that contains the pre-step pressure: the two bounding transition pressures
// (pb > pt), the temperatures there, and the lapse rates of layers i0-1, i0, i0+1.  A balloon
// moves < 150 Pa per 10 s substep, so within one agent step it can only visit these three
// layers; crossing a boundary (it happens every substep for a balloon floating at one --
// the 17 km transition lies inside the operating band) costs no transcendental: T(p) is
// re-anchored at the boundary and advanced by a short series.
// The chain of transition pressures P_{i+1} = P_i (T_{i+1}/T_i)^(-g/(R L_i)) uses d_pow_fast.
struct AtmWindow {
  double pb, pt;      // pressures at the bottom (higher p) and top (lower p) of layer i0
  double tb, tt;      // temperatures there
  double hb;          // height at pb
  double lapse_m1, lapse_0, lapse_p1;
  double r_pb, r_pt;  // 1/pb, 1/pt
  int i0;
};
BLE_FN double atm_lapse_f64(int i, double alpha) {
  switch (i) {
    case 0: return (1 - alpha) * -0.007 + alpha * -0.0058;
    case 1: return (1 - alpha) * 0.006 + alpha * 0.005;
    case 2: return (1 - alpha) * 0.001 + alpha * 0.001;
    case 3: return (1 - alpha) * 0.0028 + alpha * 0.0028;
    case 4: return (1 - alpha) * 0.0 + alpha * 0.0;
    case 5: return (1 - alpha) * -0.0028 + alpha * -0.0028;
    default: return (1 - alpha) * -0.002 + alpha * -0.002;
  }
}
BLE_FN double atm_height_f64c(int i) {
  switch (i) {
    case 0: return -610.0; case 1: return 17000.0; case 2: return 21000.0; case 3: return 32000.0;
    case 4: return 47000.0; case 5: return 51000.0; case 6: return 71000.0; default: return 85000.0;
  }
}
// The alpha-only part of the window: lapse rates of layers 0-2 and the two transitions (17 km, 21 km) that
// bound every pressure a flying balloon can reach -- two pows; constant over an episode, so the fused
// multi-step kernel evaluates it once per launch.
struct AtmBase { double l0, l1, l2, t1, p1, t2, p2; };
// (the part that costs nothing: lapse rates and layer-top temperatures; the two transition pressures are the pows)
BLE_FN void atm_base_linear(double alpha, AtmBase* b) {
  b->l0 = atm_lapse_f64(0, alpha); b->l1 = atm_lapse_f64(1, alpha); b->l2 = atm_lapse_f64(2, alpha);
  b->t1 = 300.0 + b->l0 * (17000.0 - -610.0);
  b->t2 = b->t1 + b->l1 * (21000.0 - 17000.0);
}
BLE_FN AtmBase atm_base(double alpha) {
  const double g = 9.80665;
  AtmBase b;
  atm_base_linear(alpha, &b);
  b.p1 = 108870.8213 * d_pow_fast(b.t1 * (1.0 / 300.0), -g * d_rcp(kAirSpecificGasD * b.l0));
  b.p2 = b.p1 * d_pow_fast(b.t2 * d_rcp(b.t1), -g * d_rcp(kAirSpecificGasD * b.l1));
  return b;
}
BLE_FN AtmWindow atm_window_from(const AtmBase& b, double alpha, double p, uint32_t* flags) {
  const double g = 9.80665;
  AtmWindow w;
  *flags |= !(p <= 108870.8213) ? kFlagPressureRange : 0u;
  const double l0 = b.l0, l1 = b.l1, l2 = b.l2, t1 = b.t1, p1 = b.p1, t2 = b.t2, p2 = b.p2;
  const bool in0 = p > p1;
  w.i0 = in0 ? 0 : 1;
  w.pb = in0 ? 108870.8213 : p1; w.pt = in0 ? p1 : p2;
  w.tb = in0 ? 300.0 : t1;       w.tt = in0 ? t1 : t2;
  w.hb = in0 ? -610.0 : 17000.0;
  w.lapse_m1 = l0; w.lapse_0 = in0 ? l0 : l1; w.lapse_p1 = in0 ? l1 : l2;
  if (__builtin_expect(!(p > p2), 0)) {
    BLE_STEP_EVENT(3);
    double t_base = t2, p_base = p2, lapse = l2, t_top = t2, p_top = p2;
    int i = 2;
#pragma unroll 1
    for (;;) {
      lapse = atm_lapse_f64(i, alpha);
      const double dh = atm_height_f64c(i + 1) - atm_height_f64c(i);
      t_top = t_base + lapse * dh;
      if (lapse == 0.0)
        p_top = p_base * d_exp_fast(-(g * dh) * d_rcp(kAirSpecificGasD * t_top));
      else
        p_top = p_base * d_pow_fast(t_top * d_rcp(t_base), -g * d_rcp(kAirSpecificGasD * lapse));
      if (p > p_top || i == 6) break;
      t_base = t_top; p_base = p_top; ++i;
    }
    if (!(p > p_top)) *flags |= kFlagPressureRange;
    w.pb = p_base; w.pt = p_top; w.tb = t_base; w.tt = t_top; w.hb = atm_height_f64c(i); w.i0 = i;
    w.lapse_0 = lapse;
    w.lapse_m1 = atm_lapse_f64(i - 1, alpha);
    w.lapse_p1 = atm_lapse_f64(i < 6 ? i + 1 : 6, alpha);
  }
  w.r_pb = d_rcp(w.pb); w.r_pt = d_rcp(w.pt);
  return w;
}
BLE_FN AtmWindow atm_window(double alpha, double p, uint32_t* flags) {
  return atm_window_from(atm_base(alpha), alpha, p, flags);
}
// Atmosphere.at_height (standard_atmosphere.py:89-120): pressure and temperature at a height, walking the layers from the ground like
// the reference's transition tables (:169-202).  fp64; used once per episode on the host side (the 50 000 ft bound of the initial-condition
// sampler) -- the transition itself only ever goes from pressure to height.
BLE_FN void atm_at_height_f64(double alpha, double h, double* pressure, double* temperature, uint32_t* flags) {
  const double g = 9.80665;
  *flags |= !(h >= -610.0 && h < atm_height_f64c(7)) ? kFlagPressureRange : 0u;      // (the reference asserts the same range, :94-95)
  double t_base = 300.0, p_base = 108870.8213, p = 0.0, t = 0.0;
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    const double lapse = atm_lapse_f64(i, alpha);
    const double top = atm_height_f64c(i + 1);
    const bool here = h < top || i == 6;
    const double dh = (here ? h : top) - atm_height_f64c(i);
    t = t_base + lapse * dh;
    if (lapse == 0.0)
      p = p_base * d_exp_fast(-(g * dh) * d_rcp(kAirSpecificGasD * t));
    else
      p = p_base * d_pow_fast(t * d_rcp(t_base), -g * d_rcp(kAirSpecificGasD * lapse));
    if (here) break;
    t_base = t; p_base = p;
  }
  *pressure = p; *temperature = t;
}
// height and temperature at p inside layer i0 of the window (standard_atmosphere.py:135-150)
BLE_FN void atm_at_pressure_f64(const AtmWindow& w, double alpha, double p, double* height, double* temperature) {
  (void)alpha;
  // T = T_b (p / p_b)^k, k = -R_d L / g; h = h_b + (T - T_b) / L -- the reference's (pow - 1) T_b / L + h_b and T_b + L (h - h_b) with the
  // division by L taken once, as a refined reciprocal, and none by g; the isothermal layer (L = 0: T = T_b, h from the logarithm) by
  // selection (k = 0 gives pow = 1 there; 0 x 1/0 is discarded)
  const double lapse = w.lapse_0;
  const double lg = d_log_fast(p * w.r_pb);
  const double t = w.tb * d_exp_fast(((-kAirSpecificGasD / 9.80665) * lapse) * lg);
  const double h_iso = d_fma((-kAirSpecificGasD / 9.80665) * w.tb, lg, w.hb);
  const double h_lapse = d_fma(t - w.tb, d_rcp(lapse), w.hb);
  *height = lapse == 0.0 ? h_iso : h_lapse;
  *temperature = t;
}
// T(p1) from T(p0) inside one layer: T1 = T0 (p1/p0)^k, k = -R_d L / g = kl, |p1/p0 - 1| < 2e-2.
// rp0 = 1/p0.  Series in fp64 (log1p to x^6, expm1 to y^5): relative error < 2e-14.
// (kl = (-R_d / g) * lapse: constant inside a layer, carried by the caller)
BLE_FN double atm_temperature_advance(double t0, double p0, double rp0, double p1, double kl, const StrideK& K = stride_k_literal()) {
  double x = (p1 - p0) * rp0;
  double lg = x * d_fma(x, d_fma(x, d_fma(x, d_fma(x, d_fma(x, K.lg6, K.lg5), K.lg4), K.lg3), -0.5), 1.0);
  double y = kl * lg;
  double em1 = y * d_fma(y, d_fma(y, d_fma(y, d_fma(y, K.ex5, K.ex4), K.ex3), 0.5), 1.0);
  return d_fma(t0, em1, t0);
}
// three-way pick on VALUES (kept in registers; a select between struct members makes the
// compiler spill the struct to scratch and index it)
BLE_FN double pick3(int j, double m1, double c0, double p1) {
  double r = c0;
  r = j < 0 ? m1 : r;
  r = j > 0 ? p1 : r;
  return r;
}
// Height relative to a layer transition (pressure pb, temperature tb) for q within a few Pa
// of it, inside a layer of lapse rate L (fp64 version of atm_height_rel_boundary).
BLE_FN double atm_height_rel_boundary_f64(double q, double pb, double r_pb, double tb, double lapse) {
  const double x = (q - pb) * r_pb;
  const double lg = x * d_fma(x, d_fma(x, d_fma(x, -0.25, 1.0 / 3.0), -0.5), 1.0);
  if (lapse == 0.0) return (-kAirSpecificGasD / 9.80665) * tb * lg;
  const double y = (-kAirSpecificGasD / 9.80665) * lapse * lg;
  const double em1 = y * d_fma(y, d_fma(y, d_fma(y, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0);
  return (tb * em1) * d_rcp(lapse);          // (reciprocal + Newton, 2e-15: a true fp64 division is ~30 instructions)
}
// 1 / (H(p + d) - H(p)), d = +-1 Pa (balloon.py:438-442), fp64.  j = layer of p in the
// window, t_p = T(p), rp = 1/p.  The cancellation-free form
//   dH = (T(p)/L) expm1(k log1p(d/p)),  k = -R_d L / g
// replaces the difference of two ~17 km heights; when p and p + d lie on different sides
// of a layer transition both heights are measured from that transition.
BLE_FN double atm_inv_delta_height_f64(const AtmWindow& w, int j, double lapse, double kl, double cur_hi, double cur_lo,
                                       double p, double rp, double d, double t_p, const StrideK& K = stride_k_literal()) {
  const double x = d * rp;                      // |x| ~ 1e-4: log1p to x^3 (next term 2.5e-13 relative)
  const double lg = x * d_fma(x, d_fma(x, K.lg3, -0.5), 1.0);
  const bool iso = lapse == 0.0;
  const double y = kl * lg;                     // kl = (-R_d / g) lapse; |y| ~ 2e-5: expm1 to y^3
  const double em1 = y * d_fma(y, d_fma(y, K.ex3, 0.5), 1.0);
  // dH = t_p em1 / L  ->  1/dH; in an isothermal layer (the one at 47-51 km: no balloon flies there, the reference's atmosphere has it)
  // dH = -(R/g) t_p lg, rare path (the common expression gives NaN there: 0 * 1/0, replaced)
  double inv = lapse * d_rcp(t_p * em1);
  if (__builtin_expect(wave_any(iso), 0)) if (iso) inv = d_rcp(t_p * ((-kAirSpecificGasD / 9.80665) * lg));
  const double q = p + d;
  // cur_hi / cur_lo: the transition pressures that bound the layer of p (+-inf if unknown/far)
  const bool below = q > cur_hi;            // q in the layer with higher pressure
  const bool above = !(q > cur_lo);
  if (__builtin_expect(wave_any(below || above), 0)) if (below || above) {
    BLE_STEP_EVENT(2);
    // transition that separates p and q, and the lapse rate on q's side
    const bool at_pb = (j == 0) ? below : (j < 0);
    const double pb = at_pb ? w.pb : w.pt, r_pb = at_pb ? w.r_pb : w.r_pt, tb = at_pb ? w.tb : w.tt;
    const double lapse_q = pick3(below ? j - 1 : j + 1, w.lapse_m1, w.lapse_0, w.lapse_p1);
    const double dh = atm_height_rel_boundary_f64(q, pb, r_pb, tb, lapse_q) -
                      atm_height_rel_boundary_f64(p, pb, r_pb, tb, lapse);
    inv = d_rcp(dh);
  }
  return inv;
}

// which of the window's three layers holds p: -1 (below i0, higher pressure), 0, +1
BLE_FN int atm_window_layer(const AtmWindow& w, double p) { return p > w.pb ? -1 : (p > w.pt ? 0 : 1); }
// fp32 view of the layer `j` of the window (for the dH series)
BLE_FN AtmLayer atm_layer_f32(const AtmWindow& w, int j) {
  const float inf = __builtin_huge_valf();
  AtmLayer l;
  const float pb = (float)w.pb, pt = (float)w.pt, tb = (float)w.tb, tt = (float)w.tt;
  l.p_base = j < 0 ? inf : (j == 0 ? pb : pt);
  l.t_base = j == 0 ? tb : tt;
  l.p_top = j < 0 ? pb : (j == 0 ? pt : -inf);
  l.t_top = j < 0 ? tb : tt;
  const float lm1 = (float)w.lapse_m1, l0 = (float)w.lapse_0, lp1 = (float)w.lapse_p1;
  l.lapse = j < 0 ? lm1 : (j == 0 ? l0 : lp1);
  l.lapse_below = j <= 0 ? lm1 : l0;
  l.lapse_above = j < 0 ? l0 : lp1;
  l.h_base = (float)w.hb; l.k = -kROverG * l.lapse; l.index = w.i0 + j;
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
BLE_FN int envelope_safety(int action, float superpressure, uint8_t* fsm, double mx = VehicleDefault::max_sp) {
  const double sp = superpressure;
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
BLE_FN int power_safety(int action, int32_t now, float battery_wh, int32_t* sunrise_h, int32_t* sunset,
                        uint8_t* paused, double night_load_w = 183.7, double capacity_wh = 3058.56) {
  BLE_NO_CONTRACT
  int32_t sr = *sunrise_h, ss = *sunset;
  if (now > sr) sr += ((now - sr + 86399) / 86400) * 86400;   // while now > sr: sr += 1 day
  if (now > ss) ss += ((now - ss + 86399) / 86400) * 86400;
  *sunrise_h = sr; *sunset = ss;
  const int paused_action = (action == kDown) ? kStay : action;
  const double batt = battery_wh, cap = capacity_wh;   // (the transition's constants unless the probe says otherwise)
  // With the vehicle's own capacity (every call site of the transition: this folds at compile time) the two state-of-charge
  // tests are thresholds on the NUMERATOR: a correctly rounded division is monotone in it, so x / 3058.56 < 0.05 <=> x < the
  // smallest float32 whose quotient reaches 0.05 (152.92801 = 0x1.31db24p+7), and d / 3058.56 < 0.025 <=> d < the smallest
  // double whose quotient reaches 0.025 (76.464 = 0x1.31db22d0e5604p+6).  The same decisions (tests/test_kernel_numerics_host.py)
  // without two ~30-instruction fp64 divisions per agent step.
  const bool own_capacity = capacity_wh == 3058.56;
  if (ss < sr) {  // daytime
    const bool low = own_capacity ? battery_wh < 152.92801f : batt / cap < 0.05;
    if (*paused && low) return paused_action;
    *paused = 0;
    return action;
  }
  if (*paused) return paused_action;
  double hours = d_div_const((double)(sr - now), 3600.0, 1.0 / 3600.0);       // == (double)(sr - now) / 3600.0
  double floating_charge = night_load_w * hours;
  const double left = batt - floating_charge;
  const bool short_of = own_capacity ? left < 0x1.31db22d0e5604p+6 : left / cap < 0.025;
  if (short_of) { *paused = 1; return paused_action; }
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
// float32 time coordinate of a query [h] (grid_based_wind_field.py:164-181: boomerang beyond 48 h, then float32)
BLE_FN float wind_time_coord(int32_t elapsed_s) {
  if (elapsed_s < 48 * 3600) return f_div_const((float)elapsed_s, 3600.0f, 1.0f / 3600.0f);   // == (float)elapsed_s / 3600.0f
  // _boomerang(t, 48): fp64 like the reference, then float32
  double t = (double)elapsed_s / 3600.0;
  long long cyc = (long long)(t / 48.0);
  double rem = t - 48.0 * (double)cyc;   // exact for these magnitudes
  if (rem < 0.0) rem += 48.0;
  if (rem >= 48.0) rem -= 48.0;
  return (float)((cyc & 1) ? 48.0 - rem : rem);
}
BLE_FN WindQuery wind_query(float x_m, float y_m, float pressure, int32_t elapsed_s) {
  WindQuery wq;
  // x.kilometers -> clip -> float32 (correctly rounded fp32 division == fp64 division then cast)
  float x_km = f_clamp(f_div_const(x_m, 1000.0f, 1.0f / 1000.0f), -500.0f, 500.0f);      // == x_m / 1000.0f
  float y_km = f_clamp(f_div_const(y_m, 1000.0f, 1.0f / 1000.0f), -500.0f, 500.0f);
  float p = f_clamp(pressure, 5000.0f, 14000.0f);
  const float t_h = wind_time_coord(elapsed_s);
  wind_axis(x_km, -500.0f, 1.0f / 50.0f, 50.0f, 21, &wq.ix, &wq.wx);
  wind_axis(y_km, -500.0f, 1.0f / 50.0f, 50.0f, 21, &wq.iy, &wq.wy);
  wind_axis(p, 5000.0f, 1.0f / 1000.0f, 1000.0f, 10, &wq.ip, &wq.wp);
  wind_axis(t_h, 0.0f, 1.0f / 6.0f, 6.0f, 9, &wq.it, &wq.wt);
  return wq;
}
// The same query with fp64 interpolation weights, as scipy's interpn forms them: the query point is packed as
// float32 (grid_based_wind_field.py:181) and everything after it -- (q - grid[i]) / (grid[i+1] - grid[i]), the weight
// products, the accumulation -- is fp64.  Used by the observation's forecast column, whose bearing feature puts the
// blended wind through an arccos (the transition's own lookup stays fp32: it feeds a 180 s displacement).
struct WindQueryD { int ix, iy, it; double wx, wy, wt; };
BLE_FN WindQueryD wind_query_xyt_f64(float x_m, float y_m, int32_t elapsed_s) {
  const WindQuery q = wind_query(x_m, y_m, 5000.0f, elapsed_s);
  const float x_km = f_clamp(f_div_const(x_m, 1000.0f, 1.0f / 1000.0f), -500.0f, 500.0f);
  const float y_km = f_clamp(f_div_const(y_m, 1000.0f, 1.0f / 1000.0f), -500.0f, 500.0f);
  const float t_h = wind_time_coord(elapsed_s);
  WindQueryD d;
  d.ix = q.ix; d.iy = q.iy; d.it = q.it;
  d.wx = ((double)x_km - (-500.0 + 50.0 * (double)q.ix)) * (1.0 / 50.0);
  d.wy = ((double)y_km - (-500.0 + 50.0 * (double)q.iy)) * (1.0 / 50.0);
  d.wt = ((double)t_h - 6.0 * (double)q.it) * (1.0 / 6.0);
  return d;
}
// One get_forecast lookup the way the reference evaluates it (float32 query, fp64 interpolation): the standalone
// forecast entry points (ble_forecast_f32, ble_forecast_column_f32) and the observation use this form; the result is
// within an ulp of fp64 of scipy's interpn.
BLE_FN void wind_forecast_f64(const float* __restrict__ grid, float x_m, float y_m, float pressure, int32_t elapsed_s,
                              double* u, double* v) {
  const WindQueryD q = wind_query_xyt_f64(x_m, y_m, elapsed_s);
  const float pc = f_clamp(pressure, 5000.0f, 14000.0f);
  int ip = (int)((pc - 5000.0f) * (1.0f / 1000.0f));
  ip = ip > 8 ? 8 : ip;
  const double wp = ((double)pc - (5000.0 + 1000.0 * (double)ip)) * 1e-3;
  double au = 0.0, av = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* cell = grid + ((((q.ix + a) * 21 + (q.iy + b)) * 10 + (ip + c)) * 9 + q.it) * 2;
        const double w3 = ((a ? q.wx : 1.0 - q.wx) * (b ? q.wy : 1.0 - q.wy)) * (c ? wp : 1.0 - wp);
        const double w0 = w3 * (1.0 - q.wt), w1 = w3 * q.wt;
        au = d_fma((double)cell[0], w0, au); av = d_fma((double)cell[1], w0, av);
        au = d_fma((double)cell[2], w1, au); av = d_fma((double)cell[3], w1, av);
      }
  *u = au; *v = av;
}
// 16-corner gather + blend.  Each (x, y, p) corner is 4 contiguous floats (t, t+1) x (u, v):
// 8 x 16 B per query.  The gather is split from the blend so that the kernel can issue the
// loads as soon as the state has arrived and blend after the per-step constants.
struct WindCorners { float c[8][4]; };
BLE_FN void wind_gather(const float* __restrict__ grid, const WindQuery& wq, WindCorners* out) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* cell = grid + ((((wq.ix + a) * 21 + (wq.iy + b)) * 10 + (wq.ip + c)) * 9 + wq.it) * 2;
        float* o = out->c[a * 4 + b * 2 + c];
        o[0] = cell[0]; o[1] = cell[1]; o[2] = cell[2]; o[3] = cell[3];
      }
}
BLE_FN void wind_blend_corners(const WindCorners& wc, const WindQuery& wq, float* u, float* v) {
  float au = 0.0f, av = 0.0f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* o = wc.c[a * 4 + b * 2 + c];
        float w3 = (a ? wq.wx : 1.0f - wq.wx) * (b ? wq.wy : 1.0f - wq.wy) * (c ? wq.wp : 1.0f - wq.wp);
        float w0 = w3 * (1.0f - wq.wt), w1 = w3 * wq.wt;
        au = f_fma(o[0], w0, au); av = f_fma(o[1], w0, av);
        au = f_fma(o[2], w1, au); av = f_fma(o[3], w1, av);
      }
  *u = au; *v = av;
}
BLE_FN void wind_blend(const float* __restrict__ grid, const WindQuery& wq, float* u, float* v) {
  WindCorners wc;
  wind_gather(grid, wq, &wc);
  wind_blend_corners(wc, wq, u, v);
}

// ---------------------------------------------------------------- solar
// sin/cos of an angle given in DEGREES as fp64: reduced to [-45, 45] deg + quadrant in fp64
// (so that the 36000 deg/century mean longitudes keep their precision), evaluated in fp32.
BLE_FN void sincos_deg(double deg, float* s, float* c) {
  double q = d_rint(deg * (1.0 / 90.0));
  float r = (float)(d_fma(q, -90.0, deg) * (kPiD / 180.0));
  float r2 = r * r;
  float sp = r * f_fma(r2, f_fma(r2, f_fma(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), 1.0f);
  float cp = f_fma(r2, f_fma(r2, f_fma(r2, f_fma(r2, 2.443315711809948e-5f, -1.388731625493765e-3f),
                                       4.166664568298827e-2f), -0.5f), 1.0f);
  int iq = (int)((long long)q & 3);
  float ss = (iq & 1) ? cp : sp;
  float cc = (iq & 1) ? sp : cp;
  *s = (iq & 2) ? -ss : ss;
  *c = ((iq + 1) & 2) ? -cc : cc;
}

// Time-only part of solar.solar_calculator (solar.py:65-134,170-172).  The Julian century
// and the large mean angles are fp64; once reduced, the trigonometry is fp32 (the
// equation of time and the declination need ~1e-7 absolute, not 1e-16).
struct Ephemeris {
  double eot_min;    // degrees(equation_of_time): minutes of time (solar.py:107-115)
  float sin_decl, cos_decl;
  float flux;        // W/m^2
  // time derivatives per second (analytic; the per-step change is <= 2e-3 deg so the
  // linear term is exact to ~1e-9): d(eot_min)/dt, d(sin_decl)/dt, d(cos_decl)/dt, d(flux)/dt
  float eot_min_rate, sin_decl_rate, cos_decl_rate, flux_rate;
};
BLE_FN Ephemeris ephemeris(int64_t unix_s) {
  // solar.py:66-79.  julian_day_number + fraction_of_day == 2440587.5 + unix_s / 86400
  // (exact identity for the Gregorian formula at :71-75; verified against the oracle).
  double julian_time;
  if (__builtin_expect(unix_s >= 0 && unix_s < 4294967296LL, 1)) {   // 1970 .. 2106: 32-bit arithmetic
    const uint32_t t = (uint32_t)unix_s;
    const uint32_t days = t / 86400u;
    const uint32_t sod = t - days * 86400u;
    julian_time = (2440587.5 + (double)days) + (double)sod * (1.0 / 86400.0);
  } else {
    int64_t days = unix_s / 86400;
    int64_t sod = unix_s - days * 86400;
    if (sod < 0) { sod += 86400; days -= 1; }
    julian_time = (2440587.5 + (double)days) + (double)sod * (1.0 / 86400.0);
  }
  const double jc = (julian_time - 2451545.0) * (1.0 / 36525.0);
  const float jcf = (float)jc;
  const double l0_deg = 280.46646 + jc * (36000.76983 + jc * 0.0003032);
  const double m0_deg = 357.52911 + jc * (35999.05029 - 0.0001537 * jc);
  const double om_deg = 125.04 - 1934.136 * jc;
  float sl, cl, sm, cm, so, co;
  sincos_deg(l0_deg, &sl, &cl);
  sincos_deg(m0_deg, &sm, &cm);
  sincos_deg(om_deg, &so, &co);
  const float s2l = 2.0f * sl * cl, c2l = f_fma(cl, cl, -sl * sl);
  const float s4l = 2.0f * s2l * c2l, c4l = f_fma(c2l, c2l, -s2l * s2l);
  const float s2m = 2.0f * sm * cm, c2m = f_fma(cm, cm, -sm * sm);
  const float s3m = sm * f_fma(-4.0f * sm, sm, 3.0f), c3m = cm * f_fma(4.0f * cm, cm, -3.0f);
  const float mean_obl = 23.0f + (26.0f + ((21.448f - jcf * (46.815f + jcf * (0.00059f - jcf * 0.001813f)))) * (1.0f / 60.0f)) * (1.0f / 60.0f);
  float sobl, cobl;
  sincos_deg((double)(mean_obl + 0.00256f * co), &sobl, &cobl);
  const float tan_half = sobl * f_rcp(1.0f + cobl);
  const float var_y = tan_half * tan_half;
  const float ecc = 0.016708634f - jcf * (0.000042037f + 0.0000001267f * jcf);
  const float eot = 4.0f * (var_y * s2l - 2.0f * ecc * sm + 4.0f * ecc * var_y * sm * c2l - 0.5f * var_y * var_y * s4l -
                            1.25f * ecc * ecc * s2m);
  const float eoc_deg = sm * (1.914602f - jcf * (0.004817f + 0.000014f * jcf)) + s2m * (0.019993f - 0.000101f * jcf) +
                        s3m * 0.000289f;
  float sa, ca;
  sincos_deg(l0_deg + (double)(eoc_deg - (0.00569f - 0.00478f * so)), &sa, &ca);
  Ephemeris e;
  e.eot_min = (double)(eot * kRadToDeg);
  e.sin_decl = sobl * sa;
  e.cos_decl = f_sqrt(f_fma(-e.sin_decl, e.sin_decl, 1.0f));
  const float r = (1.0f + ecc) * f_rcp(1.0f - ecc);
  const float half_r2m1 = 0.5f * f_fma(r, r, -1.0f);
  e.flux = 1366.0f * f_fma(half_r2m1, cm, 1.0f);
  // rates: L0' and M' in rad/s (the obliquity, eccentricity and nutation terms move < 1e-9 deg per step)
  const float per_s = (float)(kPiD / 180.0 / (36525.0 * 86400.0));
  const float lp = (36000.76983f + 0.0006064f * jcf) * per_s;
  const float mp = (35999.05029f - 0.0003074f * jcf) * per_s;
  const float eot_rate = 4.0f * (2.0f * var_y * c2l * lp - 2.0f * ecc * cm * mp +
                                 4.0f * ecc * var_y * (cm * c2l * mp - 2.0f * sm * s2l * lp) -
                                 2.0f * var_y * var_y * c4l * lp - 2.5f * ecc * ecc * c2m * mp);
  e.eot_min_rate = eot_rate * kRadToDeg;
  const float eoc_rate = kDegToRad * mp * (cm * (1.914602f - jcf * 0.004817f) + 2.0f * c2m * 0.019993f + 3.0f * c3m * 0.000289f);
  e.sin_decl_rate = sobl * ca * (lp + eoc_rate);
  e.cos_decl_rate = -e.sin_decl * e.sin_decl_rate * f_rcp(e.cos_decl);
  e.flux_rate = -1366.0f * half_r2m1 * sm * mp;
  return e;
}

// 1 - sin(uncorrected elevation) at one instant, fp64.
// BalloonState.latlng (balloon.py:217-220 -> spherical_geometry.py:44-76) folded into the
// zenith formula (solar.py:136-139); no atan2/asin/acos:
//   heading = atan2(x, y)            -> cos/sin(heading) = y/d, x/d
//   d_lng   = atan2(yy, xx)          -> cos/sin(B + d_lng) by rotating (cos B, sin B)
//   hour_angle = (B + d_lng) -+ 180  -> cos(hour_angle) = -cos(B + d_lng)   (solar.py:113-120)
// (sin_b, cos_b) of b = 360 frac_day + eot/4 + lng0 [deg]; the three nodes of a step share
// one sincos and rotate by the half-step angle.
// Within one agent step x, y move linearly (constant wind) and the ephemeris is linear,
// so S(t) is smooth: agent_step evaluates this at 3 nodes and interpolates quadratically
// (|error| <= |d3S/dt3| h^3 * 0.064 <= 1.8e-8, and ~0 around local noon where d3S/dt3 -> 0).
struct SunSC { float sin_el, cos_el; };
BLE_FN double sun_one_minus_sin_f64(double sin_lat0, double cos_lat0, double x, double y, double sin_b, double cos_b,
                                    double sin_decl, double cos_decl) {
  const double d2 = x * x + y * y;
  const bool moved = d2 > 0.0;
  const double inv_d = moved ? d_rsqrt(moved ? d2 : 1.0) : 0.0;
  const double cos_h = moved ? y * inv_d : 1.0;      // heading = atan2(x, y); atan2(0, 0) = 0
  const double sin_h = x * inv_d;
  const double a = (d2 * inv_d) * (1.0 / 6371000.0);  // angle = |(x, y)| / R_earth  (< 0.2 rad)
  const double a2 = a * a;
  // Taylor to a^13 / a^12: |a| < 0.2 -> truncation < 1e-17
  double sin_a = d_fma(a2, d_fma(a2, d_fma(a2, d_fma(a2, d_fma(a2, d_fma(a2, 1.0 / 6227020800.0, -1.0 / 39916800.0),
                                                         1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0) * a;
  double cos_a = d_fma(a2, d_fma(a2, d_fma(a2, d_fma(a2, d_fma(a2, d_fma(a2, 1.0 / 479001600.0, -1.0 / 3628800.0),
                                                         1.0 / 40320.0), -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
  const double sin_lat = cos_a * sin_lat0 + sin_a * cos_lat0 * cos_h;
  const double cos_lat = d_sqrt_fast(d_fma(-sin_lat, sin_lat, 1.0));
  const double yy = sin_a * cos_lat0 * sin_h;
  const double xx = cos_a - sin_lat0 * sin_lat;
  const double cos_bl = (cos_b * xx - sin_b * yy) * d_rsqrt(xx * xx + yy * yy);
  const double s = sin_lat * sin_decl - cos_lat * cos_decl * cos_bl;
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
// Branch-free: the three formulas are evaluated and selected (they share 1/S).
// `above5`: the uncorrected elevation is above the 5 deg branch point of the reference's formula (solar.py:143-155) -- the one
// branch point where the two formulas differ by more than their own precision (1.8 arcsec), so the transition decides it on
// its accurate distance (sun_fast) instead of the fp32 sine
BLE_FN SunSC sun_refract(SunSC unc, bool above5) {
  const float kSin85 = 0.99619472027f, kSinM0575 = -0.01003547478f;
  const float s = unc.sin_el, c = unc.cos_el;
  const float ct = c * f_rcp(s), ct2 = ct * ct;                       // 1/tan(el); only used when |s| > 0.01
  const float r_mid = ct * f_fma(ct2, f_fma(ct2, 0.000086f, -0.07f), 58.1f);
  const float r_neg = -20.772f * ct;
  const float s2 = s * s;
  const float e = kRadToDeg * s * f_fma(s2, f_fma(s2, f_fma(s2, 15.0f / 336.0f, 3.0f / 40.0f), 1.0f / 6.0f), 1.0f);
  const float r_low = f_fma(e, f_fma(e, f_fma(e, f_fma(e, 0.711f, -12.79f), 103.4f), -518.2f), 1735.0f);
  float refr = s > kSinM0575 ? r_low : r_neg;                          // arcseconds
  refr = above5 ? r_mid : refr;
  refr = s > kSin85 ? 0.0f : refr;
  const float dl = refr * (kDegToRad / 3600.0f);
  const float d2 = dl * dl;
  const float sd = dl * f_fma(d2, f_fma(d2, 1.0f / 120.0f, -1.0f / 6.0f), 1.0f);
  const float cd = f_fma(d2, f_fma(d2, 1.0f / 24.0f, -0.5f), 1.0f);
  SunSC r;
  r.sin_el = f_fma(s, cd, c * sd);
  r.cos_el = f_fma(c, cd, -s * sd);
  return r;
}
BLE_FN SunSC sun_refract(SunSC unc) { return sun_refract(unc, unc.sin_el > 0.08715574443f); }      // sin(5 deg)

// ---------------------------------------------------------------- fp64 asin (fdlibm e_asin.c rational form)
BLE_FN double d_asin(double x) {
  // Branch-free: lanes of one wave sit on both sides of |x| = 0.5 (721 table entries per observation), and a
  // divergent fdlibm pays for every path.  One evaluation of the rational R serves both halves:
  //   |x| <  0.5:  asin(x) = x + x R(x^2)
  //   |x| >= 0.5:  asin(|x|) = pi/2 - 2 (s + s R(t)),  t = (1 - |x|) / 2,  s = sqrt(t)
  // (without fdlibm's head/tail split of s: 2e-16 absolute instead of the last ulp)
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const double ax = __builtin_fabs(x);
  const bool big = ax >= 0.5;
  const double t = big ? (1.0 - ax) * 0.5 : x * x;
  const double rt = d_sqrt_fast(t);                        // (NaN for t == 0: discarded by the select)
  const double s = big ? (t > 0.0 ? rt : 0.0) : ax;
  const double p = t * d_fma(t, d_fma(t, d_fma(t, d_fma(t, d_fma(t, pS5, pS4), pS3), pS2), pS1), pS0);
  const double q = d_fma(t, d_fma(t, d_fma(t, d_fma(t, qS4, qS3), qS2), qS1), 1.0);
  const double y = d_fma(s, p * d_rcp(q), s);             // q in [0.3, 1]: reciprocal + Newton, 2e-15
  const double res = big ? pio2_hi - (2.0 * y - pio2_lo) : y;
  return x < 0.0 ? -res : res;
}

// ---------------------------------------------------------------- full fp64 solar calculator
// solar.solar_calculator (solar.py:43-174) in two parts: the TIME-ONLY ephemeris (declination, equation of
// time, flux -- five sincos; smooth on the scale of days) and the SITE part (hour angle, zenith, refraction).
struct SolarEphemeris {
  double sin_decl, cos_decl;
  double eot_quarter_deg;    // 0.25 * degrees(equation_of_time): the equation-of-time term of the hour angle [deg]
  double flux;               // W/m^2
};
BLE_FN void unix_day_fraction(int64_t unix_s, double* julian_century, double* frac) {
  int64_t days = unix_s / 86400;
  int64_t sod = unix_s - days * 86400;
  if (sod < 0) { sod += 86400; days -= 1; }
  // (reciprocal multiplications: <= 1 ulp of fp64 from the reference's divisions, i.e. 4e-14 deg of hour angle)
  *frac = (double)sod * (1.0 / 86400.0);
  *julian_century = (((2440587.5 + (double)days) + *frac) - 2451545.0) * (1.0 / 36525.0);
}
BLE_FN SolarEphemeris solar_ephemeris_f64(double jc) {
  const double d2r = kPiD / 180.0;
  const double l0 = d2r * (280.46646 + jc * (36000.76983 + jc * 0.0003032));
  double s2l, c2l;
  sincos_f64(2.0 * l0, &s2l, &c2l);
  const double s4l = 2.0 * s2l * c2l;
  const double m0 = d2r * (357.52911 + jc * (35999.05029 - 0.0001537 * jc));
  double sm, cm;
  sincos_f64(m0, &sm, &cm);
  const double s2m = 2.0 * sm * cm, s3m = sm * (3.0 - 4.0 * sm * sm);
  const double mean_obl = d2r * (23.0 + (26.0 + ((21.448 - jc * (46.815 + jc * (0.00059 - jc * 0.001813)))) * (1.0 / 60.0)) * (1.0 / 60.0));
  double so, co;
  sincos_f64(d2r * (125.04 - 1934.136 * jc), &so, &co);
  const double obl = mean_obl + d2r * (0.00256 * co);
  double sobl, cobl;
  sincos_f64(obl, &sobl, &cobl);
  const double th = sobl * d_rcp(1.0 + cobl), var_y = th * th;          // tan(obl / 2)
  const double ecc = 0.016708634 - jc * (0.000042037 + 0.0000001267 * jc);
  const double eot = 4.0 * (var_y * s2l - 2.0 * ecc * sm + 4.0 * ecc * var_y * sm * c2l - 0.5 * var_y * var_y * s4l -
                            1.25 * ecc * ecc * s2m);
  const double eoc = d2r * (sm * (1.914602 - jc * (0.004817 + 0.000014 * jc)) + s2m * (0.019993 - 0.000101 * jc) + s3m * 0.000289);
  double sa, ca;
  sincos_f64(l0 + eoc - d2r * (0.00569 - 0.00478 * so), &sa, &ca);
  SolarEphemeris e;
  e.sin_decl = sobl * sa;
  e.cos_decl = d_sqrt_fast(d_fma(-e.sin_decl, e.sin_decl, 1.0));      // |decl| < 24 deg: argument > 0.83
  e.eot_quarter_deg = 0.25 * (eot * (180.0 / kPiD));
  const double r = (1 + ecc) * d_rcp(1 - ecc);
  e.flux = 1366.0 * (1 + 0.5 * (r * r - 1) * cm);
  return e;
}
// solar flux alone (solar.py:170-172)
BLE_FN double solar_flux_f64(double jc) {
  const double m0 = (kPiD / 180.0) * (357.52911 + jc * (35999.05029 - 0.0001537 * jc));
  double sm, cm;
  sincos_f64(m0, &sm, &cm);
  const double ecc = 0.016708634 - jc * (0.000042037 + 0.0000001267 * jc);
  const double r = (1 + ecc) * d_rcp(1 - ecc);
  return 1366.0 * (1 + 0.5 * (r * r - 1) * cm);
}
// elevation [deg], refraction corrected, at a site (sin lat, cos lat, lng [deg]) and day fraction
BLE_FN double solar_elevation_site_f64(double sin_lat, double cos_lat, double lng_deg, double frac, double sin_decl,
                                       double cos_decl, double eot_quarter_deg) {
  // cos(hour_angle) = -cos(radians(1440 frac + degrees(eot) + 4 lng) / 4)   (solar.py:113-120)
  double sh, ch;
  sincos_f64((kPiD / 180.0) * (360.0 * frac + eot_quarter_deg + lng_deg), &sh, &ch);
  double s = sin_lat * sin_decl - cos_lat * cos_decl * ch;
  s = s > 1.0 ? 1.0 : (s < -1.0 ? -1.0 : s);
  const double el = d_asin(s) * (180.0 / kPiD);      // 90 - degrees(acos(s))
  // refraction (solar.py:141-157) through one reciprocal: 1 / tan(el) = c / s  (fp64 divisions expand to ~30
  // instructions each and this runs 721 times per observation)
  const double c2 = 1.0 - s * s;
  const double c = c2 > 0.0 ? d_sqrt_fast(c2) : 0.0;
  const double it = c * d_rcp(s == 0.0 ? 1e-300 : s), it2 = it * it;
  double refr;
  if (el > 85.0) refr = 0.0;
  else if (el > 5.0) refr = it * (58.1 + it2 * (-0.07 + it2 * 0.000086));
  else if (el > -0.575) refr = 1735.0 + el * (-518.2 + el * (103.4 + el * (-12.79 + el * 0.711)));
  else refr = -20.772 * it;
  return el + refr * (1.0 / 3600.0);
}
BLE_FN double solar_elevation_f64(double sin_lat, double cos_lat, double lng_deg, int64_t unix_s, double* flux_out) {
  double jc, frac;
  unix_day_fraction(unix_s, &jc, &frac);
  const SolarEphemeris e = solar_ephemeris_f64(jc);
  if (flux_out) *flux_out = e.flux;
  return solar_elevation_site_f64(sin_lat, cos_lat, lng_deg, frac, e.sin_decl, e.cos_decl, e.eot_quarter_deg);
}

// BalloonState.latlng (spherical_geometry.py:44-76) as (sin lat, cos lat, lng [deg]) in fp64.
BLE_FN void latlng_f64(double lat0_deg, double lng0_deg, double x, double y, double* sin_lat, double* cos_lat,
                       double* lng_deg) {
  double sl0, cl0;
  sincos_f64(lat0_deg * (kPiD / 180.0), &sl0, &cl0);
  const double d = sqrt(x * x + y * y);
  double cos_h = 1.0, sin_h = 0.0;
  if (d > 0.0) { cos_h = y / d; sin_h = x / d; }
  double sa, ca;
  sincos_f64(d / 6371000.0, &sa, &ca);
  const double sl = ca * sl0 + sa * cl0 * cos_h;
  const double yy = sa * cl0 * sin_h, xx = ca - sl0 * sl;
  // d_lng = atan2(yy, xx): |d_lng| < 0.2 rad here, xx > 0 -> asin of the normalised sine
  const double d_lng = d_asin(yy / sqrt(xx * xx + yy * yy));
  *sin_lat = sl; *cos_lat = sqrt(1.0 - sl * sl);
  *lng_deg = lng0_deg + d_lng * (180.0 / kPiD);
}

constexpr float kSinMinSolarEl = -0.07396924496f;  // sin(-4.242 deg), solar.py:38

// The solar thresholds of the transition: day / night and the attenuation cut-off at
// MIN_SOLAR_EL_DEG = -4.242 deg (solar.py:38,199-200; balloon.py:524) and the two panel-shadow
// elevations (solar.py:212-236).  The fast path decides them on the fp32 sin(el) (floor ~1e-7,
// i.e. ~6e-6 deg); a stride whose sin(el) -- or whose uncorrected sin(el), for the 5 deg branch
// point of the refraction formula, solar.py:143-155 -- is within kSunBand of a threshold is
// re-decided on the reference's own fp64 chain (sun_exact), so a threshold never flips a
// stride early or late (it did, ~2 strides per 10^6 env-steps: 0.07 K, 1.3 Wh steps).
constexpr float kSinShadow33 = 0.61205375195f;   // sin(37.738149 deg): panels 3.3 m below the envelope
constexpr float kSinShadow27 = 0.56489306688f;   // sin(34.394865 deg): panels 2.7 m below
constexpr float kSin5 = 0.08715574443f;
struct SunState { float sin_el, cos_el; bool day, sh33, sh27; };
// Round 4: the four decisions are made on DISTANCES in w = 1 - sin(uncorrected elevation), the quantity the step interpolates.
// The refraction-corrected elevation is a monotone function of the uncorrected one, so "el_corrected > -4.242 deg" is
// "w < kOmsDay" with kOmsDay the (fp64, solved offline from solar.py:141-157) w of the uncorrected elevation whose corrected
// value is the threshold -- likewise the two shadow elevations; the 5 deg branch point is on the uncorrected elevation itself.
// A step carries d_j = (float)(w(0) - threshold_j), the difference taken in fp64 BEFORE rounding, and a stride adds the
// quadratic's increment q(k) = k (c2 k + c1): near a threshold both are small, the sum is exact (Sterbenz) and its error is
// the increment's own rounding, ~2e-9 -- against the 1.7e-7 of comparing an fp32 sin(el) that is O(1).  What is left is the
// quadratic interpolation itself (<= 1.8e-8, see sun_one_minus_sin_f64), so the band inside which a stride is re-decided on the
// reference's fp64 chain (sun_exact) shrinks from 1e-6 to 6e-8: 17x fewer of the 1 200-instruction cold chains (they were the
// tail of every one-step launch, r03_step_launch.md), and decisions that no longer depend on an fp32 rounding.
constexpr double kOmsDay = 1.0752991363576547;        // uncorrected -4.318410158082298 deg -> corrected -4.242 deg (solar.py:38)
constexpr double kOmsShadow33 = 0.38823376985560987;  // uncorrected 37.71732276171744 deg -> corrected 37.738149050524044 deg
constexpr double kOmsShadow27 = 0.4354459419249166;   // uncorrected 34.371330036126224 deg -> corrected 34.39486500086289 deg
constexpr double kOmsRefr5 = 0.9128442572523419;      // 1 - sin(5 deg)
constexpr float kSunBand = 6.0e-8f;                    // at the reference's 18 strides per step
// The quadratic's error is |S'''| h^3 0.064 with h the half step: 1.8e-8 at 18 strides, and it grows with the cube of the step
// length (6e-8 near 27 strides, 6.6e-7 at BLE_MAX_SUBSTEPS = 60), so the band does too: a fixed 6e-8 would let the four decisions
// differ from sun_exact's on long steps without the fp64 chain being taken.  Wave-uniform: `substeps` is a kernel argument.
BLE_FN float sun_band(int substeps) {
  const float r = (float)substeps * (1.0f / 18.0f);
  return substeps <= 18 ? kSunBand : kSunBand * (r * r * r);
}
struct SunThresholds { float d_day, d_s33, d_s27, d_r5, band; };
BLE_FN SunThresholds sun_thresholds(double w0, int substeps) {      // w0 = 1 - sin(uncorrected elevation) at the first node of the step, fp64
  SunThresholds t;
  t.d_day = (float)(w0 - kOmsDay); t.d_s33 = (float)(w0 - kOmsShadow33); t.d_s27 = (float)(w0 - kOmsShadow27); t.d_r5 = (float)(w0 - kOmsRefr5);
  t.band = sun_band(substeps);
  return t;
}
// oms = c0 + q: the interpolated 1 - sin(uncorrected elevation) of the stride, q its increment over the step's first node
BLE_FN SunState sun_fast(float oms, float q, const SunThresholds& t, bool* near) {
  const SunSC unc = sun_from_one_minus_sin(oms);
  const float d_day = t.d_day + q, d_s33 = t.d_s33 + q, d_s27 = t.d_s27 + q, d_r5 = t.d_r5 + q;
  const SunSC cor = sun_refract(unc, d_r5 < 0.0f);
  SunState r;
  r.sin_el = cor.sin_el; r.cos_el = cor.cos_el;
  r.day = d_day < 0.0f;          // el > -4.242  (sun_exact's comparisons, mapped)
  r.sh33 = d_s33 <= 0.0f;        // el >= 37.738...
  r.sh27 = d_s27 <= 0.0f;        // el >= 34.394...
  *near = f_minnum(f_minnum(fabsf(d_day), fabsf(d_s33)), f_minnum(fabsf(d_s27), fabsf(d_r5))) < t.band;
  return r;
}
// solar.solar_calculator on BalloonState.latlng in fp64 (the oracle's chain op for op), cold.
BLE_FN SunState sun_exact(double lat0_deg, double lng0_deg, double x, double y, int64_t unix_s) {
  double sl, cl, lng;
  latlng_f64(lat0_deg, lng0_deg, x, y, &sl, &cl, &lng);
  const double el = solar_elevation_f64(sl, cl, lng, unix_s, nullptr);
  double se, ce;
  sincos_f64(el * (kPiD / 180.0), &se, &ce);
  SunState r;
  r.sin_el = (float)se; r.cos_el = (float)ce;
  r.day = el > -4.242;
  r.sh33 = el >= 37.738149050524044;
  r.sh27 = el >= 34.39486500086289;
  return r;
}

// solar_atmospheric_attenuation (solar.py:177-209) from sin(el); 0 at night (`day` false).
// The reference's pressure range check (:194-197) is done by the caller once per agent step.
BLE_FN float solar_attenuation(float sin_el, float pressure, bool day) {
  const float t = 614.0f * sin_el;
  const float root = f_sqrt(f_fma(t, t, 1229.0f));
  // sqrt(1229 + t^2) - t, written without cancellation for t > 0
  const float diff = t > 0.0f ? 1229.0f * f_rcp(root + t) : root - t;
  const float airmass = (pressure * (0.34764f / 101325.0f)) * diff;
  const float att = 0.5f * (f_exp2(airmass * (-0.65f * kLog2e)) + f_exp2(airmass * (-0.95f * kLog2e)));   // constants folded: 3 multiplies fewer
  return day ? att : 0.0f;
}
// solar_power (solar.py:515-536) with balloon_shadow (:212-236) folded in.
// the part that depends on the sun alone: projected, shadowed panel area per unit (ble_step_split.h evaluates it a stride ahead)
BLE_FN float solar_panel_factor(const SunState& sun) {
  const float kCos35 = 0.81915204429f, kSin35 = 0.57357643635f;
  const float kCos65 = 0.42261826174f, kSin65 = 0.90630778704f;
  float sh33 = sun.sh33 ? 0.4392f : 1.0f;
  float sh27 = sun.sh27 ? 0.4392f : 1.0f;
  // 4 c35 and 2 c65 with the factors inside the constants (powers of two: the same bits as scaling the sums)
  float c35x4 = f_fma(sun.cos_el, 4.0f * kCos35, sun.sin_el * (4.0f * kSin35));
  float c65x2 = f_fma(sun.cos_el, 2.0f * kCos65, sun.sin_el * (2.0f * kSin65));
  return f_fma(c35x4, sh33, c65x2 * sh27);
}
BLE_FN float solar_power_from_factor(float panel_factor, float attenuation) { return 210.0f * attenuation * panel_factor; }
BLE_FN float solar_power(const SunState& sun, float attenuation) {
  return solar_power_from_factor(solar_panel_factor(sun), attenuation);
}

// ---------------------------------------------------------------- thermal
// thermal.py:52-230.
constexpr float kSolarAbsorptivityTotal =
    0.01435f * (1.0f + (1.0f - 0.01435f - 0.0291f) / (1.0f - 0.0291f));
// a^(-1/4) and a^(-1/10) for a > 0 (fp32-representable): fp32 hardware seed (~2e-7) + one
// Newton step of y <- y (n + 1 - a y^n) / n, quadratic: ~1e-13 relative.  No fp64 division,
// no libm.  Used by the fp64 convection model below.
BLE_FN double d_inv_root4(double a, double five = 5.0) {
  const double y = (double)f_sqrt(f_rsqrt((float)a));
  const double y2 = y * y;
  return y * d_fma(-a * y2, y2, five) * 0.25;
}
// (scale: 0.1, the Newton step's own factor, times whatever constant the caller multiplies the root by)
BLE_FN double d_inv_root10(double a, double eleven = 11.0, double scale = 0.1) {
  const double y = (double)f_exp2(-0.1f * f_log2((float)a));
  const double y2 = y * y, y4 = y2 * y2, y5 = y4 * y;
  return y * d_fma(-a * y5, y5, eleven) * scale;
}
// Increment of the internal temperature over one 10 s stride, fp64:
//   10 s * d_balloon_temperature_dt(V, 68.5, T_int, T_amb, p, el, flux, IR)     (thermal.py:175-230,
//   convective_heat_air_factor :150-172).
// The four heat flows (each ~10 W/m^2 x ~700 m^2) nearly cancel, and the result feeds the
// buoyancy difference rho V - m whose map amplifies errors (ble_step_core.h): an fp32 evaluation
// (~1e-6 W/m^2 absolute) put 1 env-step in 10^4 beyond the 1e-5 parity bar; evaluated in fp64
// none are left.  Formulated without pow / division:
//   yc = V^(-1/3) (caller; Newton-refined)  ->  V^(2/3) = V yc,  1/(2 r) = yc / (2 k1),  (2 r)^3 = 6 V / pi
//   rt = T_amb^(-1/2)                       ->  rho / mu = p (M/R) (T + 110.4) rt^5 / 1.458e-6,  1/T = rt^2
//   Ra^(1/4) = Ra (Ra^(-1/4))^3,  (T/273.15)^0.9 = T T^(-1/10) / 273.15^0.9
// Only (1 + 2.69e-8 Ra)^(1/12) -- a term of ~2 next to 0.457 Ra^(1/4) ~ 300 -- stays an fp32 pow.
// q_solar_area = flux * attenuation * 0.25 * absorptivity [W/m^2] (the transition passes an fp32 product: the
// solar geometry's own floor), q_earth_area = earth_heat_per_area(IR) (per-episode constant).
constexpr double kStefanBoltzmannD = 0.000000056704;
BLE_FN double total_absorptivity_d(double a) { return a * d_fma(-a, 1.0 / (1.0 - 0.0291), 2.0); }   // a (1 + (1-a-r)/(1-r))
BLE_FN double earth_heat_per_area_f64(double upwelling_ir, uint32_t* flags) {   // thermal.py:209-213
  const double t_bb = d_sqrt(d_sqrt(upwelling_ir * (1.0 / kStefanBoltzmannD)));
  const double f = total_absorptivity_d(d_fma(0.000232, t_bb - 210.0, 0.04587));
  *flags |= (f < 0.0 || f > 1.0) ? kFlagAbsorptivity : 0u;
  return upwelling_ir * 0.4605 * f;
}
template <bool kExactTwelfthRoot = false>
BLE_FN double thermal_increment_f64(double vol, double yc, double t_int, double t_amb, double p, double q_solar_area,
                                    double q_earth_area, const StrideK& K = stride_k_literal(), double thermal_scale = VehicleDefault::thermal_scale) {
  constexpr double kR2 = 0.38483473658887897;          // (3 / (4 pi))^(2/3)
  constexpr double kR1 = 0.62035049089940009;          // (3 / (4 pi))^(1/3)
  const double v23 = vol * yc;
  // emitted (thermal.py:214-217): sigma T^4 a (2 - a / (1 - r)), a = 0.04587 + 0.000232 (T - 210), expanded in T
  const double t2 = t_int * t_int;
  const double q_emit = (t2 * t2) * d_fma(d_fma(kEmitA, t_int, K.emit_b), t_int, K.emit_c);
  // convection.  Ra = g beta |dT| (2 r)^3 (rho / mu)^2 Pr with rho / mu = p (M/R) (T + 110.4) T^(-5/2) / 1.458e-6, beta = 1 / T,
  // (2 r)^3 = 6 V / pi:  Ra = [scale Pr(T)] (p (T + 110.4))^2 V |dT| T^(-6)
  const double rt = d_rcp(t_amb), rt2 = rt * rt, rt3 = rt2 * rt;
  const double pw = p * (t_amb + K.t110);
  const double dt = t_amb - t_int;
  double ra = (d_fma(kRayleighT, t_amb, K.ra_t0) * (pw * pw)) * ((vol * (rt3 * rt3)) * __builtin_fabs(dt));
  ra = d_max(ra, 1e-30);
  const double y4 = d_inv_root4(ra, K.five);
  const double ra14 = (ra * y4) * (y4 * y4);
  // (the cold-start Newton of the reset / observation kernels converges on differences of this function:
  // there the twelfth root is fp64 too)
  // kExactTwelfthRoot: the fp32 root (1e-7 relative) refined by one Newton step on y^12 = x in fp64 -- 5e-14, a quarter
  // of the instructions of exp(log(x) / 12)
  const double tw_arg = d_fma(2.69e-8, ra, 1.0);
  double tw = (double)f_pow((float)tw_arg, 1.0f / 12.0f);
  if (kExactTwelfthRoot) {
    const double y2 = tw * tw, y4r = y2 * y2, y8 = y4r * y4r;
    tw = d_fma(-tw * (1.0 / 12.0), d_fma(y8 * y4r, d_rcp(tw_arg), -1.0), tw);       // y - y (y^12 / x - 1) / 12
  }
  const double nusselt = d_fma(0.457, ra14, 2.0 + tw);
  // k(T) / (2 r) = 0.0241 (T / 273.15)^0.9 V^(-1/3) / (2 (3 / (4 pi))^(1/3)): the constants ride on the tenth root's Newton factor
  const double q_conv = ((nusselt * (t_amb * d_inv_root10(t_amb, K.eleven, K.cond_tenth))) * yc) * dt;
  const double q = (q_solar_area + q_earth_area) + (q_conv - q_emit);
  return (q * v23) * thermal_scale;                   // 10 s x 4 pi r^2 / (c_p m_envelope), r^2 = (3 / (4 pi))^(2/3) V^(2/3)  (thermal.py:221-230)
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
BLE_FN void superpressure_volume_f64(double mols_air, double t_int, double p, double rp, double* volume, double* sp,
                                     const StrideK& K = stride_k_literal(), double dvdp = VehicleDefault::dvdp,
                                     double four_dvdp = VehicleDefault::four_dvdp, double inv_dvdp = VehicleDefault::inv_dvdp) {
  // rp = 1/p.  Fully inflated branch: V from the quadratic (balloon.py:596-604); the
  // superpressure then follows from the envelope model V = V0 + dV/dp * sp, which is the
  // same root written without the division p Vu / V (relative difference ~1e-14).
  const double w = ((K.lift + mols_air) * kGasConstantD) * t_int;       // p Vu = n R T
  double vu = w * rp;
  double b = -(K.v0 - dvdp * p);
  double c4 = four_dvdp * w;                                            // 4 dV/dp (p Vu)
  double v = 0.5 * (d_sqrt_rs(d_fma(b, b, c4)) - b);
  bool slack = vu <= K.v0;
  *volume = slack ? vu : v;
  *sp = slack ? 0.0 : (v - K.v0) * inv_dvdp;
}

// ---------------------------------------------------------------- ACS
// acs.py:24-68.  prm1 = pressure_ratio - 1.
// Fan-efficiency table acs.py:31-41, rows = power 100/200/300/400 W, columns = pressure
// ratio 1.05 .. 1.35 step 0.025.  `tab` points at 4 x 13 floats (LDS copy in the kernel).
struct AcsEfficiencyTable { float v[4 * 13]; };
constexpr AcsEfficiencyTable kAcsEfficiencyValues = {{
    0.4f, 0.4f, 0.3f, 0.2f, 0.2f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
    0.4f, 0.3f, 0.3f, 0.30f, 0.25f, 0.23f, 0.20f, 0.15f, 0.12f, 0.10f, 0.0f, 0.0f, 0.0f,
    0.0f, 0.3f, 0.25f, 0.25f, 0.25f, 0.20f, 0.20f, 0.20f, 0.2f, 0.15f, 0.13f, 0.12f, 0.11f,
    0.0f, 0.23f, 0.23f, 0.23f, 0.23f, 0.23f, 0.20f, 0.20f, 0.20f, 0.18f, 0.16f, 0.15f, 0.13f}};
BLE_CONST_TABLE AcsEfficiencyTable kAcsEfficiencyTable = kAcsEfficiencyValues;
#define kAcsEfficiency (kAcsEfficiencyTable.v)
// fp64 versions for the transition (the mass flow feeds rho V - m); table entries are exact in fp32
BLE_FN double acs_power_f64(double prm1) {
  const double seg1 = d_fma(prm1 - 0.05, 200.0 / 0.15, 100.0);
  const double seg2 = d_fma(prm1 - 0.2, 100.0 / 0.05, 300.0);
  const double w = prm1 <= 0.2 ? seg1 : seg2;
  return d_max(d_min(w, 400.0), 100.0);
}
BLE_FN double acs_efficiency_f64(const float* tab, double prm1, double power) {
  const double fx = d_max(d_min((prm1 - 0.05) * 40.0, 12.0), 0.0);
  const double fy = d_max(d_min((power - 100.0) * 0.01, 3.0), 0.0);
  int ix = (int)fx; ix = ix > 11 ? 11 : ix;
  int iy = (int)fy; iy = iy > 2 ? 2 : iy;
  const double wx = fx - (double)ix, wy = fy - (double)iy;
  const float* r0 = tab + iy * 13 + ix;
  const double z00 = (double)r0[0], z01 = (double)r0[1], z10 = (double)r0[13], z11 = (double)r0[14];
  const double lo = d_fma(wx, z01 - z00, z00), hi = d_fma(wx, z11 - z10, z10);
  return d_fma(wy, hi - lo, lo);
}

// The ACS's DOWN branch as the transition evaluates it: power W(pr) = get_most_efficient_power and mass flow
// eff(pr, W(pr)) W(pr) / 3600 (balloon.py:500-510).  Along the curve W(pr) the bilinear efficiency table is a
// piecewise polynomial of pr alone: every power node (100 / 200 / 300 / 400 W) is reached exactly at a
// pressure-ratio node (1.05 / 1.125 / 1.2 / 1.25), so inside each of the 12 ratio intervals W is linear, both
// interpolation weights are linear and the mass flow is a cubic.  Entry i: c0..c3 of the mass flow [kg/s] and
// w0, w1 of the power [W] in t = prm1 - (0.05 + 0.025 i), t clamped to [0, 0.025] (flat outside the table,
// acs.py:44-68 `fill_value=None` / the flat end segments of the power curve).  Identical to the two-table
// form to 1e-16 (tests: probe vs oracle); 3 LDS reads + 5 FMAs instead of two dependent table walks.
constexpr int kAcsPolyDoubles = 12 * 6;
// The 12 x 6 coefficients are a COMPILE-TIME table (constexpr evaluation: plain IEEE double arithmetic, no FMA
// contraction, the same numbers on the device and in the host build of the test tooling); the transition copies it from
// constant memory into LDS at kernel entry instead of rebuilding it per launch.
struct AcsPolyTable { double c[kAcsPolyDoubles]; };
constexpr double acs_power_constexpr(double prm1) {        // acs.py:44-50 == acs_power_f64
  const double w = prm1 <= 0.2 ? (prm1 - 0.05) * (200.0 / 0.15) + 100.0 : (prm1 - 0.2) * (100.0 / 0.05) + 300.0;
  return w > 400.0 ? 400.0 : (w < 100.0 ? 100.0 : w);
}
constexpr AcsPolyTable make_acs_poly_table() {
  AcsPolyTable t = {};
  for (int i = 0; i < 12; ++i) {
    double* c = t.c + 6 * i;
    const double x0 = 0.05 + 0.025 * (double)i;
    const double w0 = acs_power_constexpr(x0), w1 = (acs_power_constexpr(x0 + 0.025) - w0) * 40.0;
    const double wm = acs_power_constexpr(x0 + 0.0125);
    double fy = (wm - 100.0) * 0.01;
    fy = fy > 3.0 ? 3.0 : (fy < 0.0 ? 0.0 : fy);
    int iy = (int)fy; iy = iy > 2 ? 2 : iy;
    const double a0 = (w0 - 100.0) * 0.01 - (double)iy, a1 = w1 * 0.01;
    const float* tab = kAcsEfficiencyValues.v;
    const double z00 = (double)tab[iy * 13 + i], z01 = (double)tab[iy * 13 + i + 1];
    const double z10 = (double)tab[iy * 13 + 13 + i], z11 = (double)tab[iy * 13 + 14 + i];
    const double l0 = z00, l1 = (z01 - z00) * 40.0, h0 = z10, h1 = (z11 - z10) * 40.0;
    const double e0 = l0 + a0 * (h0 - l0), e1 = l1 + a0 * (h1 - l1) + a1 * (h0 - l0), e2 = a1 * (h1 - l1);
    c[0] = e0 * w0 * (1.0 / 3600.0); c[1] = (e0 * w1 + e1 * w0) * (1.0 / 3600.0);
    c[2] = (e1 * w1 + e2 * w0) * (1.0 / 3600.0); c[3] = e2 * w1 * (1.0 / 3600.0);
    c[4] = w0; c[5] = w1;
  }
  return t;
}
constexpr AcsPolyTable kAcsPolyValues = make_acs_poly_table();
BLE_CONST_TABLE AcsPolyTable kAcsPoly = kAcsPolyValues;
BLE_FN void acs_down_poly(const double* poly, double prm1, double* power_w, double* mdot, const StrideK& K = stride_k_literal()) {
  int i = (int)((prm1 - K.acs_x0) * K.forty);      // truncation: [-1, 1) -> 0
  i = i < 0 ? 0 : (i > 11 ? 11 : i);
  const double t = d_max(d_min(prm1 - d_fma(0.025, (double)i, K.acs_x0), 0.025), 0.0);
  const double* c = poly + 6 * i;
  *mdot = d_fma(d_fma(d_fma(c[3], t, c[2]), t, c[1]), t, c[0]);
  *power_w = d_fma(c[5], t, c[4]);
}

// power_table.lookup (power_table.py:21-38)
BLE_FN float power_table_lookup_f64(double prd, double s, uint32_t* flags) {
  if (!(prd >= 0.99 && prd <= 5.0)) *flags |= kFlagPowerTable;
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
BLE_FN float power_table_lookup(float pr, float soc, uint32_t* flags) {
  // thresholds are compared in fp64 against the reference's double literals
  if (!(pr >= 0.99f && pr <= 5.0f)) *flags |= kFlagPowerTable;
  uint32_t ignored = 0;
  return power_table_lookup_f64((double)pr, (double)soc, &ignored);
}

// ---------------------------------------------------------------- reward
// perciatelli_reward_function (env/balloon_env.py:44-102), base term.
BLE_FN float reward_distance(float x, float y) {
  float d = f_sqrt(f_fma(x, x, y * y));
  if (d <= 50000.0f) return 1.0f;
  return 0.4f * f_exp((-0.69314718056f / 100.0f) * ((d - 50000.0f) * 0.001f));
}

}  // namespace ble
