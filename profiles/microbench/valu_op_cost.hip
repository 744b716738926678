// Issue cost of the fp64 / conversion / integer instructions the observation and transition kernels are made of:
// 8 independent chains per lane, 64 instructions of one kind per inner iteration, one wave per SIMD (grid 1024)
// and two (grid 2048).  ns per wave instruction = ms * 1e6 / (iters * 64) [/ 2 for two waves].
#include <hip/hip_runtime.h>
#include <cstdio>
enum { FMA64, MUL64, ADD64, RSQ64, RCP64, RNDNE64, CVTI64, LDEXP64, MIN64, FMA32, EXP32, RSQ32, CVT6432, MOV64, CND32, LSHLADD, MOV32, SQRT64, MULLO32, MULU24, MADU24, XOR32, PKFMA32, FRACT64, FLOOR32, CVTI32F };
template <int KIND>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  double v[8]; float f[8]; int n[8];
  for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x * 1e-3; f[i] = (float)v[i]; n[i] = (int)seed + i + threadIdx.x; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const int i = j % 8;
      if (KIND == FMA64) v[i] = __builtin_fma(v[i], 1.0000001, 0.5);
      if (KIND == MUL64) v[i] = v[i] * 1.0000001;
      if (KIND == ADD64) v[i] = v[i] + 1.0000001;
      if (KIND == RSQ64) v[i] = __builtin_amdgcn_rsq(v[i]);
      if (KIND == RCP64) v[i] = __builtin_amdgcn_rcp(v[i]);
      if (KIND == SQRT64) v[i] = __builtin_amdgcn_sqrt(v[i]);
      if (KIND == RNDNE64) { asm volatile("v_rndne_f64 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) % 8])); }
      if (KIND == CVTI64) { n[i] = (int)v[i]; asm volatile("" : "+v"(n[i])); }
      if (KIND == LDEXP64) v[i] = __builtin_amdgcn_ldexp(v[i], n[i]);
      if (KIND == MIN64) v[i] = __builtin_fmin(v[i], v[(i + 1) % 8]);
      if (KIND == FMA32) { asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(f[i]) : "v"(f[i]), "v"(f[(i + 1) % 8])); }
      if (KIND == EXP32) f[i] = __builtin_amdgcn_exp2f(f[i]);
      if (KIND == RSQ32) f[i] = __builtin_amdgcn_rsqf(f[i]);
      if (KIND == CVT6432) { f[i] = (float)v[i]; asm volatile("" : "+v"(f[i])); }
      if (KIND == MOV64) { asm volatile("v_mov_b64 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) % 8])); }
      if (KIND == MOV32) { asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 1) % 8])); }
      if (KIND == CND32) n[i] = n[i] > j ? n[(i + 1) % 8] : n[i];
      if (KIND == LSHLADD) n[i] = (n[i] << 3) + n[(i + 1) % 8];
      if (KIND == MULLO32) { asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(n[i]) : "v"(n[i]), "v"(n[(i + 1) % 8])); }
      if (KIND == MULU24) { asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(n[i]) : "v"(n[i]), "v"(n[(i + 1) % 8])); }
      if (KIND == MADU24) { asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(n[i]) : "v"(n[i]), "v"(n[(i + 1) % 8]), "v"(n[(i + 2) % 8])); }
      if (KIND == XOR32) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(n[i]) : "v"(n[i]), "v"(n[(i + 1) % 8])); }
      if (KIND == PKFMA32) { typedef float f2 __attribute__((ext_vector_type(2))); f2 a = {f[i], f[(i + 4) % 8]}; asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(a) : "v"(a)); f[i] = a.x; }
      if (KIND == FRACT64) v[i] = __builtin_amdgcn_fract(v[i]);
      if (KIND == FLOOR32) { asm volatile("v_floor_f32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) % 8])); }
      if (KIND == CVTI32F) { n[i] = (int)f[i]; asm volatile("" : "+v"(n[i])); }
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
#define RUN(name, KIND) { float m1 = timeit([&] { k<KIND><<<1024, 64>>>(out, iters, 1.5); }); float m2 = timeit([&] { k<KIND><<<2048, 64>>>(out, iters, 1.5); }); \
    printf("%-18s %7.2f ns / wave instr (1 wave per SIMD)  %7.2f (2 waves per SIMD, per instr of either)\n", name, m1 * 1e6 / (iters * 64.0), m2 * 1e6 / (iters * 128.0)); }
  RUN("v_fma_f64", FMA64) RUN("v_mul_f64", MUL64) RUN("v_add_f64", ADD64) RUN("v_rsq_f64", RSQ64) RUN("v_rcp_f64", RCP64) RUN("v_sqrt_f64", SQRT64)
  RUN("v_rndne_f64", RNDNE64) RUN("v_cvt_i32_f64", CVTI64) RUN("v_ldexp_f64", LDEXP64) RUN("v_min_f64", MIN64)
  RUN("v_fma_f32", FMA32) RUN("v_exp_f32", EXP32) RUN("v_rsq_f32", RSQ32) RUN("v_cvt_f32_f64", CVT6432)
  RUN("v_mov_b64", MOV64) RUN("v_mov_b32", MOV32) RUN("v_cndmask (cmp+sel)", CND32) RUN("v_lshl_add_u32", LSHLADD)
  RUN("v_mul_lo_u32", MULLO32) RUN("v_mul_u32_u24", MULU24) RUN("v_mad_u32_u24", MADU24) RUN("v_xor_b32", XOR32) RUN("v_pk_fma_f32 (+mov)", PKFMA32)
  RUN("v_fract_f64", FRACT64) RUN("v_floor_f32", FLOOR32) RUN("v_cvt_i32_f32", CVTI32F)
  return 0;
}
