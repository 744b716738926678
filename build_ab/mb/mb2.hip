#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  d4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = seed + threadIdx.x, b = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 12 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; hipMalloc(&out, 4096 * 64 * 8);
  const int iters = 20000;
#define RUN(name, expr) { float ms = timeit([&] { expr; }); printf("%-40s %8.3f ms  %7.2f ns per MFMA per wave\n", name, ms, ms * 1e6 / iters / 12); }
  RUN("1 chain, 1 wave/SIMD", (k<1><<<1024, 64>>>(out, iters, 1.0)));
  RUN("2 chains, 1 wave/SIMD", (k<2><<<1024, 64>>>(out, iters, 1.0)));
  RUN("3 chains, 1 wave/SIMD", (k<3><<<1024, 64>>>(out, iters, 1.0)));
  RUN("4 chains, 1 wave/SIMD", (k<4><<<1024, 64>>>(out, iters, 1.0)));
  RUN("6 chains, 1 wave/SIMD", (k<6><<<1024, 64>>>(out, iters, 1.0)));
  RUN("1 chain, 2 waves/SIMD", (k<1><<<2048, 64>>>(out, iters, 1.0)));
  RUN("2 chains, 2 waves/SIMD", (k<2><<<2048, 64>>>(out, iters, 1.0)));
  RUN("4 chains, 2 waves/SIMD", (k<4><<<2048, 64>>>(out, iters, 1.0)));
  return 0;
}
