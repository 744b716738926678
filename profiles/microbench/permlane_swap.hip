#include <hip/hip_runtime.h>
#include <cstdio>
// sum over the four lanes jq, jq + 16, jq + 32, jq + 48 with v_permlane32_swap / v_permlane16_swap (gfx950)
__device__ inline double colsum(double v) {
  union { double d; int w[2]; } a, lo, hi;
  a.d = v;
  // halves: after the swap one result holds the upper half's values in every lane, the other the lower half's
  auto r0 = __builtin_amdgcn_permlane32_swap(a.w[0], a.w[0], false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(a.w[1], a.w[1], false, false);
  lo.w[0] = r0[0]; lo.w[1] = r1[0]; hi.w[0] = r0[1]; hi.w[1] = r1[1];
  const double s = lo.d + hi.d;
  a.d = s;
  auto q0 = __builtin_amdgcn_permlane16_swap(a.w[0], a.w[0], false, false);
  auto q1 = __builtin_amdgcn_permlane16_swap(a.w[1], a.w[1], false, false);
  lo.w[0] = q0[0]; lo.w[1] = q1[0]; hi.w[0] = q0[1]; hi.w[1] = q1[1];
  return lo.d + hi.d;
}
__global__ void k(double* out, int* raw) {
  const int l = threadIdx.x;
  const double v = (double)(1 << (l / 16)) * 1000.0 + (l % 16);   // g-dependent weight + jq
  out[l] = colsum(v);
  auto r = __builtin_amdgcn_permlane32_swap(l, l, false, false);
  auto q = __builtin_amdgcn_permlane16_swap(l, l, false, false);
  raw[l] = r[0]; raw[64 + l] = r[1]; raw[128 + l] = q[0]; raw[192 + l] = q[1];
}
int main() {
  double* d; int* r; hipMalloc(&d, 64 * 8); hipMalloc(&r, 256 * 4);
  k<<<1, 64>>>(d, r); double h[64]; int hr[256];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 1024, hipMemcpyDeviceToHost);
  // expected: (1+2+4+8)*1000 + 4*jq
  int bad = 0; for (int l = 0; l < 64; ++l) if (h[l] != 15000.0 + 4 * (l % 16)) ++bad;
  printf("colsum mismatches: %d (lane 5: %.1f, lane 37: %.1f)\n", bad, h[5], h[37]);
  for (int b = 0; b < 4; ++b) { printf("%s:", b == 0 ? "p32 vdst" : b == 1 ? "p32 src " : b == 2 ? "p16 vdst" : "p16 src "); for (int l = 0; l < 64; l += 8) printf(" %d", hr[64 * b + l]); printf("\n"); }
  return 0;
}
