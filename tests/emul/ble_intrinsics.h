// TEST TOOLING: libm stand-ins for csrc/ble_intrinsics.h so that the lane functions of ble_physics.h /
// ble_step_core.h build with g++ (numerics triage without a GPU).  tests/emul force-includes this file
// (g++ -include) and its include guard keeps csrc/ble_intrinsics.h out; the package never sees this file.  The fp32 "hardware" ops are correctly rounded libm calls and
// the fp64 reciprocal seeds are fp32-precision, like v_rcp_f64 / v_rsq_f64.
#ifndef BLE_INTRINSICS_H_
#define BLE_INTRINSICS_H_
#include <math.h>
#include <stdint.h>

#define BLE_FN static inline
#define BLE_CONST_TABLE static const
#define BLE_NO_CONTRACT

namespace ble {

BLE_FN float f_exp2(float x) { return exp2f(x); }
BLE_FN float f_log2(float x) { return log2f(x); }
BLE_FN float f_rcp(float x) { return 1.0f / x; }
BLE_FN float f_sqrt(float x) { return sqrtf(x); }
BLE_FN float f_rsqrt(float x) { return 1.0f / sqrtf(x); }
BLE_FN float f_fma(float a, float b, float c) { return fmaf(a, b, c); }
BLE_FN double d_fma(double a, double b, double c) { return fma(a, b, c); }
BLE_FN double d_vreg(double k) { return k; }
BLE_FN bool wave_any(bool c) { return c; }
BLE_FN int i_opaque(int v) { return v; }
BLE_FN double d_rint(double x) { return rint(x); }
BLE_FN double d_sqrt(double x) { return sqrt(x); }
BLE_FN float f_minnum(float a, float b) { return fminf(a, b); }
BLE_FN double d_min(double a, double b) { return fmin(a, b); }
BLE_FN double d_max(double a, double b) { return fmax(a, b); }
BLE_FN double d_rcp_seed(double x) { return (double)(1.0f / (float)x); }
BLE_FN double d_rsq_seed(double x) { return (double)(1.0f / sqrtf((float)x)); }
BLE_FN double d_frexp_mant(double x) { int e; return frexp(x, &e); }
BLE_FN int d_frexp_exp(double x) { int e; frexp(x, &e); return e; }
BLE_FN double d_ldexp(double x, int e) { return ldexp(x, e); }

}  // namespace ble
#endif  // BLE_INTRINSICS_H_
