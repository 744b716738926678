// ble_intrinsics.h -- the gfx950 instruction layer under the lane functions of ble_physics.h.
// Device only (hipcc --offload-arch=gfx950).  The include guard is shared with the libm stand-in that the test
// tooling force-includes first (g++ -include tests/emul/ble_intrinsics.h); the package builds only this one.
#ifndef BLE_INTRINSICS_H_
#define BLE_INTRINSICS_H_
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define BLE_FN __device__ __forceinline__
#define BLE_CONST_TABLE __device__ __constant__ const
#define BLE_NO_CONTRACT _Pragma("clang fp contract(off)")

namespace ble {

BLE_FN float f_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
BLE_FN float f_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
BLE_FN float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32
BLE_FN float f_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32
BLE_FN float f_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32
BLE_FN float f_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
BLE_FN double d_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// A 64-bit constant kept as ONE value in a vector-register pair.  hipcc materialises a non-inline fp64 literal as two 32-bit halves, shares
// equal halves between literals and re-assembles (and, for an fma's addend, copies) the pair in front of every use: 2-3 issue slots of the
// stride loop per constant.  An opaque pair made once per agent step is used in place (StrideK, ble_physics.h).
BLE_FN double d_vreg(double k) { asm("" : "+v"(k)); return k; }
// true if the predicate holds on any lane of the wavefront: a rare per-lane path guarded by `if (wave_any(c)) if (c) {...}` costs the common
// case a compare and ONE scalar branch (s_cbranch_vccnz) instead of an exec-mask save / branch / restore
#ifdef BLE_NO_VOTE      // (A/B knob: the compiler's own exec-mask form of the rare paths)
BLE_FN bool wave_any(bool c) { return c; }
#else
BLE_FN bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
#endif
// an integer the optimiser cannot see through: used inside rarely taken blocks so that the common path does not carry their
// induction variables (10 k, (double)k, (k + 1) << 8 ...)
BLE_FN int i_opaque(int v) { asm("" : "+v"(v)); return v; }
BLE_FN double d_rint(double x) { return __builtin_rint(x); }
BLE_FN double d_sqrt(double x) { return __builtin_sqrt(x); }
BLE_FN float f_minnum(float a, float b) { return __builtin_fminf(a, b); }   // v_min_f32 / v_min3_f32 (f_min in ble_physics.h is a compare + select)
BLE_FN double d_min(double a, double b) { return __builtin_fmin(a, b); }   // v_min_f64 (operands are never NaN here)
BLE_FN double d_max(double a, double b) { return __builtin_fmax(a, b); }
BLE_FN double d_rcp_seed(double x) { return __builtin_amdgcn_rcp(x); }   // v_rcp_f64: 4.3e-8 relative (measured)
BLE_FN double d_rsq_seed(double x) { return __builtin_amdgcn_rsq(x); }   // v_rsq_f64: 5.0e-8 relative (measured)
BLE_FN double d_frexp_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }   // [0.5, 1)
BLE_FN int d_frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
BLE_FN double d_ldexp(double x, int e) { return __builtin_amdgcn_ldexp(x, e); }

}  // namespace ble
#endif  // BLE_INTRINSICS_H_
