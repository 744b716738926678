// fp64 MFMA shapes on gfx950: issue time of v_mfma_f64_16x16x4_f64 (2 048 flop) against v_mfma_f64_4x4x4_4b_f64 (4 blocks x
// 128 flop = 512 flop) -- does a block-triangular 16 x 16 product on 4 x 4 blocks (10 of 16 block columns non-zero) pay?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_shapes mfma_f64_shapes.hip && ./mfma_f64_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(64) void k16(double* out, int iters, double seed) {
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
template <int CH>
__global__ __launch_bounds__(64) void k4(double* out, int iters, double seed) {
  double acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = 0.0;
  double a = seed + threadIdx.x, b = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 12 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c];
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
#define RUN(name, flop, expr) { float ms = timeit([&] { expr; }); double ns = ms * 1e6 / iters / 12; printf("%-44s %8.3f ms  %7.2f ns per MFMA per wave  %6.1f flop/ns/SIMD\n", name, ms, ns, flop / ns); }
  RUN("16x16x4, 1 chain, 1 wave/SIMD", 2048.0, (k16<1><<<1024, 64>>>(out, iters, 1.0)));
  RUN("16x16x4, 4 chains, 1 wave/SIMD", 2048.0, (k16<4><<<1024, 64>>>(out, iters, 1.0)));
  RUN("4x4x4 (4 blocks), 1 chain, 1 wave/SIMD", 512.0, (k4<1><<<1024, 64>>>(out, iters, 1.0)));
  RUN("4x4x4 (4 blocks), 4 chains, 1 wave/SIMD", 512.0, (k4<4><<<1024, 64>>>(out, iters, 1.0)));
  RUN("4x4x4 (4 blocks), 12 chains, 1 wave/SIMD", 512.0, (k4<12><<<1024, 64>>>(out, iters, 1.0)));
  RUN("4x4x4 (4 blocks), 4 chains, 2 waves/SIMD", 512.0, (k4<4><<<2048, 64>>>(out, iters, 1.0)));
  return 0;
}
