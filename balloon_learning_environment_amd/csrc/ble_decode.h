// Tail of the wind-field VAE decoder (generative/vae.py:149-186) for one output point; shared by
// the kernel (ble_kernels.hip::ble_decode_flow_kernel) and the host test build (tests/emul).
#pragma once
#include <math.h>

#include "ble_physics.h"

namespace ble {

// jax.image.resize(method='linear') 7 -> 23 along one axis: output index a samples the input at
// x = (a + 0.5) * 7 / 23 - 0.5 with a triangle kernel; renormalising the edge weights equals
// clamping the two taps.
BLE_FN void resize_tap(int a, int* tap0, float* w1) {
  const float x = ((float)a + 0.5f) * (7.0f / 23.0f) - 0.5f;
  const float fl = floorf(x);
  *tap0 = (int)fl;
  *w1 = x - fl;
}
// psi: the flow fields of one sample with the field index fastest ([7][7][90]), already offset to
// field f.  One point (a, b) of the resized 23 x 23 lattice.
BLE_FN float decode_resized(const float* psi, int a, int b, const int* tap0, const float* w1) {
  const int a0 = tap0[a], b0 = tap0[b];
  const int a_lo = a0 < 0 ? 0 : a0, a_hi = a0 + 1 > 6 ? 6 : a0 + 1;
  const int b_lo = b0 < 0 ? 0 : b0, b_hi = b0 + 1 > 6 ? 6 : b0 + 1;
  const float wa = w1[a], wb = w1[b];
  const float lo = f_fma(wb, psi[(a_lo * 7 + b_hi) * 90] - psi[(a_lo * 7 + b_lo) * 90], psi[(a_lo * 7 + b_lo) * 90]);
  const float hi = f_fma(wb, psi[(a_hi * 7 + b_hi) * 90] - psi[(a_hi * 7 + b_lo) * 90], psi[(a_hi * 7 + b_lo) * 90]);
  return f_fma(wa, hi - lo, lo);
}
// (i, j) in 0..20.  u = d psi / da, v = -d psi / db on the resized lattice: central differences of four of its points.
BLE_FN void decode_flow_from_lattice(float r_ip2_jp1, float r_i_jp1, float r_ip1_jp2, float r_ip1_j, float* u, float* v) {
  *u = 0.5f * (r_ip2_jp1 - r_i_jp1);
  *v = -0.5f * (r_ip1_jp2 - r_ip1_j);
}
BLE_FN void decode_flow_point(const float* psi, int i, int j, const int* tap0, const float* w1, float* u, float* v) {
  decode_flow_from_lattice(decode_resized(psi, i + 2, j + 1, tap0, w1), decode_resized(psi, i, j + 1, tap0, w1),
                           decode_resized(psi, i + 1, j + 2, tap0, w1), decode_resized(psi, i + 1, j, tap0, w1), u, v);
}

}  // namespace ble
