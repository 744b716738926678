// Latency of DEPENDENT VALU instructions for a lone wave (one wave per SIMD, grid 1024): C independent chains per lane,
// 64 instructions of one kind per inner iteration; C = 1 is a pure dependent chain.  ns per wave instruction.
// (valu_op_cost.hip measures the issue cost with 8 chains; this one asks how much instruction-level parallelism the
// transition's stride needs before the issue cost is what is paid.)
#include <hip/hip_runtime.h>
#include <cstdio>
enum { FMA64, MUL64, ADD64, FMA32, MUL32, CVT6432, CVT3264, RSQ64, EXP32, MIX, MULLO, MUL24, XOR32, CNDSG };
template <int KIND, int C>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  double v[8]; float f[8]; unsigned n[8];
  for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x * 1e-3; f[i] = (float)v[i]; n[i] = (unsigned)(seed * 1000) + i * 77u + threadIdx.x; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const int i = j % C;
      if (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[7]), "v"(v[6]));
      if (KIND == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(v[7]));
      if (KIND == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(v[7]));
      if (KIND == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[7]), "v"(f[6]));
      if (KIND == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[7]));
      if (KIND == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[i]));
      if (KIND == EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
      if (KIND == CVT6432) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(v[i])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[i]) : "v"(f[i])); }
      if (KIND == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[7]));
      if (KIND == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(n[7]));
      if (KIND == XOR32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[i]) : "v"(n[7]));
      if (KIND == CNDSG) { asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" :: "v"(n[i]), "v"(n[7]) : "s20", "s21"); asm volatile("s_nop 1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(n[i]) : "v"(n[6]) : "s20", "s21"); }
      if (KIND == MIX) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(v[7])); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[7])); }
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
  double* out; hipMalloc(&out, 2048 * 64 * 8);
  const int iters = 4000;
#define RUN1(name, KIND, C, per) { float m1 = timeit([&] { k<KIND, C><<<1024, 64>>>(out, iters, 1.0000001); }); printf("  C=%d %6.2f", C, m1 * 1e6 / (iters * 64.0 * per)); }
#define RUN(name, KIND, per) { printf("%-26s", name); RUN1(name, KIND, 1, per) RUN1(name, KIND, 2, per) RUN1(name, KIND, 3, per) RUN1(name, KIND, 4, per) RUN1(name, KIND, 6, per) printf("   ns per wave instruction\n"); }
  RUN("v_fma_f64", FMA64, 1) RUN("v_mul_f64", MUL64, 1) RUN("v_add_f64", ADD64, 1) RUN("v_fma_f32", FMA32, 1) RUN("v_mul_f32", MUL32, 1)
  RUN("v_rsq_f64", RSQ64, 1) RUN("v_exp_f32", EXP32, 1) RUN("cvt f64->f32->f64 (pair)", CVT6432, 2) RUN("v_mul_f64 + v_mul_f32", MIX, 2) RUN("v_mul_lo_u32", MULLO, 1) RUN("v_mul_u32_u24", MUL24, 1) RUN("v_xor_b32", XOR32, 1) RUN("v_cmp (sgpr) + nop + cndmask", CNDSG, 3)
  return 0;
}
