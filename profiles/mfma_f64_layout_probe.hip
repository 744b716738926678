// Measures the register layout of v_mfma_f64_16x16x4_f64 on the device:
//   hipcc --offload-arch=gfx950 -O2 profiles/mfma_f64_layout_probe.hip -o mfma_probe && ./mfma_probe
// Result on gfx950 (this round): A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16],
// D lane l register v = D[4 v + l / 16][l % 16]  (the first hypothesis below, i = 4 (l / 16) + v, is
// reported as MISMATCH and the table that follows shows the actual mapping).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {
  // hypothesis: a = A[i = l%16][k = l/16], b = B[k = l/16][j = l%16], d[v] = D[i = 4*(l/16)+v][j = l%16]
  int l = threadIdx.x;
  double a = A[(l % 16) * 4 + l / 16];
  double b = B[(l / 16) * 16 + l % 16];
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (i * 7 % 13) - 6 + 0.25 * i; hB[i] = (i * 5 % 11) - 5 + 0.125 * i; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, dA, dB, dD); hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
  // find, for each (lane, v), which (i, j) it matches
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    int i = 4 * (l / 16) + v, j = l % 16;
    if (hD[l * 4 + v] != ref[i * 16 + j]) ok = 0;
  }
  printf("hypothesis D[i=4*(l/16)+v][j=l%%16]: %s\n", ok ? "OK" : "MISMATCH");
  if (!ok) for (int l = 0; l < 64; l += 5) for (int v = 0; v < 4; ++v) {
    for (int e = 0; e < 256; ++e) if (hD[l * 4 + v] == ref[e]) printf("lane %d v %d -> i %d j %d\n", l, v, e / 16, e % 16);
  }
  return 0;
}
