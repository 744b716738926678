#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NV, int KIND>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  d4 acc0 = {0,0,0,0}, acc1 = {0,0,0,0};
  double a = seed + threadIdx.x, b = seed * 0.5;
  double v[8]; float f[8]; int n[8];
  for (int i = 0; i < 8; ++i) { v[i] = seed + i; f[i] = (float)seed + i; n[i] = (int)seed + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == 0) v[j % 8] = __builtin_fma(v[j % 8], 1.0000001, 0.5);
        if (KIND == 1) f[j % 8] = __builtin_fmaf(f[j % 8], 1.0000001f, 0.5f);
        if (KIND == 2) n[j % 8] = n[j % 8] * 3 + 1;
      }
    }
  }
  double s = acc0[0] + acc1[1] + acc0[2] + acc1[3];
  for (int i = 0; i < 8; ++i) s += v[i] + f[i] + n[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NV, int KIND, bool MF>
__global__ __launch_bounds__(64) void kv(double* out, int iters, double seed) {   // VALU only
  double v[8]; float f[8]; int n[8];
  for (int i = 0; i < 8; ++i) { v[i] = seed + i; f[i] = (float)seed + i; n[i] = (int)seed + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == 0) v[j % 8] = __builtin_fma(v[j % 8], 1.0000001, 0.5);
        if (KIND == 1) f[j % 8] = __builtin_fmaf(f[j % 8], 1.0000001f, 0.5f);
        if (KIND == 2) n[j % 8] = n[j % 8] * 3 + 1;
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i] + f[i] + n[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; hipMalloc(&out, 1024 * 64 * 8 * 4);
  const int iters = 20000; const int grid = 1024;   // one wave per SIMD
  // cycles per iteration = ms * 2.4e6 / iters   (8 MFMAs per iteration)
#define RUN(name, expr) { float ms = timeit([&] { expr; }); printf("%-34s %8.3f ms  %7.1f cycles/iter (at 2.4 GHz)\n", name, ms, ms * 2.4e6 / iters); }
  RUN("8 mfma", (k<0, 0><<<grid, 64>>>(out, iters, 1.0)));
  RUN("8 mfma + 32 dp fma", (k<8, 0><<<grid, 64>>>(out, iters, 1.0)));
  RUN("8 mfma + 64 dp fma", (k<16, 0><<<grid, 64>>>(out, iters, 1.0)));
  RUN("8 mfma + 128 dp fma", (k<32, 0><<<grid, 64>>>(out, iters, 1.0)));
  RUN("8 mfma + 64 f32 fma", (k<16, 1><<<grid, 64>>>(out, iters, 1.0)));
  RUN("8 mfma + 128 f32 fma", (k<32, 1><<<grid, 64>>>(out, iters, 1.0)));
  RUN("8 mfma + 128 int mad", (k<32, 2><<<grid, 64>>>(out, iters, 1.0)));
  RUN("64 dp fma only", (kv<16, 0, false><<<grid, 64>>>(out, iters, 1.0)));
  RUN("128 dp fma only", (kv<32, 0, false><<<grid, 64>>>(out, iters, 1.0)));
  RUN("128 f32 fma only", (kv<32, 1, false><<<grid, 64>>>(out, iters, 1.0)));
  RUN("128 int mad only", (kv<32, 2, false><<<grid, 64>>>(out, iters, 1.0)));
  // two waves per SIMD: does a VALU-only wave slow an MFMA-only wave?
  RUN("8 mfma, 2 waves/SIMD", (k<0, 0><<<2 * grid, 64>>>(out, iters, 1.0)));
  RUN("128 dp only, 2 waves/SIMD", (kv<32, 0, false><<<2 * grid, 64>>>(out, iters, 1.0)));
  return 0;
}
