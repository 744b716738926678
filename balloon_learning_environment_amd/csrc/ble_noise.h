// Wind noise: the additive "truth minus forecast" field of env/simplex_wind_noise.py:82-211.
//
// Structure as in the reference: per wind component five harmonics, each a 4-D gradient noise
// sampled at (x km / x_spacing, y km / y_spacing, p / p_spacing, t h / t_spacing) + a random
// offset in [-1, 1)^4, scaled to a variance of 1.02 (m/s)^2, combined with the reference's
// weights and variance adjustment (:190-211).
//
// PARITY UNPINNED: the reference's primitive is `opensimplex==0.3`'s
// noise4d, whose source is not part of the checkout and which no reference test pins.  The
// primitive here is Gustavson's 4-D simplex noise with hashed gradients (no permutation tables:
// 10 generators x 256 B per environment would not fit LDS at one lane per environment), its
// empirical variance measured on the device (tests/test_gpu_noise.py).
//
// Every function below evaluates with `fp contract(off)` (a fused multiply-add only where f_fma is written): the generator runs in two
// kernels -- ble_wind_noise_kernel and,
// inlined, ble_step_kernel<noise> (ABI 3) -- and the two must produce the same bits whatever the surrounding code lets the
// compiler fuse (tests/test_gpu_parity.py: fused rollout in the ground-truth wind == noise kernel + single steps).
#pragma once
#include <math.h>

#include "ble_reset.h"

namespace ble {

struct Harmonic { float weight, x_spacing, y_spacing, p_spacing, t_spacing; };
// simplex_wind_noise.py:50-64 (weight, x km, y km, pressure Pa, time h); harmonic k = 5 comp + h, comp 0 = u, 1 = v.
// A table in constant memory: as a function-local array indexed at run time it lived on the stack (48 B of scratch per lane
// in every kernel that evaluates the noise).
struct HarmonicTable { Harmonic h[10]; };
BLE_CONST_TABLE HarmonicTable kHarmonics = {{{0.1445f, 702.269f, 2116.987f, 2587.802f, 245.0f},
                                             {0.2766f, 1483.570f, 752.124f, 646.208f, 16.39f},
                                             {0.2627f, 276.810f, 147.040f, 587.702f, 3.836f},
                                             {0.2137f, 10214.525f, 1512.216f, 965.629f, 41.780f},
                                             {0.1025f, 181.286f, 420.942f, 8500.0f, 245.0f},
                                             {0.2716f, 1974.228f, 2028.814f, 713.697f, 26.435f},
                                             {0.2684f, 699.738f, 541.845f, 632.116f, 9.530f},
                                             {0.2348f, 217.750f, 196.522f, 686.825f, 3.546f},
                                             {0.1186f, 47.500f, 43.048f, 66.553f, 8.424f},
                                             {0.1066f, 3663.291f, 232.023f, 7499.741f, 225.0f}}};
BLE_FN Harmonic harmonic_params(int comp, int h) { return kHarmonics.h[5 * comp + h]; }
// RN(1 / spacing) of the same table: coordinate / spacing is evaluated as f_div_const (ble_physics.h: reciprocal product + one exact-remainder
// correction = the bits of the division, 3 issue slots instead of 12; four divisions per harmonic, forty per env-step)
struct HarmonicRcp { float x, y, p, t; };
struct HarmonicRcpTable { HarmonicRcp h[10]; };
constexpr HarmonicRcp harmonic_rcp_of(float xs, float ys, float ps, float ts) { return HarmonicRcp{1.0f / xs, 1.0f / ys, 1.0f / ps, 1.0f / ts}; }
BLE_CONST_TABLE HarmonicRcpTable kHarmonicRcp = {{harmonic_rcp_of(702.269f, 2116.987f, 2587.802f, 245.0f),
                                                  harmonic_rcp_of(1483.570f, 752.124f, 646.208f, 16.39f),
                                                  harmonic_rcp_of(276.810f, 147.040f, 587.702f, 3.836f),
                                                  harmonic_rcp_of(10214.525f, 1512.216f, 965.629f, 41.780f),
                                                  harmonic_rcp_of(181.286f, 420.942f, 8500.0f, 245.0f),
                                                  harmonic_rcp_of(1974.228f, 2028.814f, 713.697f, 26.435f),
                                                  harmonic_rcp_of(699.738f, 541.845f, 632.116f, 9.530f),
                                                  harmonic_rcp_of(217.750f, 196.522f, 686.825f, 3.546f),
                                                  harmonic_rcp_of(47.500f, 43.048f, 66.553f, 8.424f),
                                                  harmonic_rcp_of(3663.291f, 232.023f, 7499.741f, 225.0f)}};
BLE_FN HarmonicRcp harmonic_rcp(int comp, int h) { return kHarmonicRcp.h[5 * comp + h]; }
constexpr float kSimplex4Variance = 0.088392f;   // of simplex4() below: measured 0.0889 (tests/test_gpu_noise.py) == the reference's SIMPLEX_VARIANCE (:70)
constexpr float kNoiseVariance = 1.02f;          // simplex_wind_noise.py:73

BLE_FN uint32_t float_bits_u32(float v) { uint32_t b; __builtin_memcpy(&b, &v, 4); return b; }
BLE_FN float u32_bits_float(uint32_t b) { float v; __builtin_memcpy(&v, &b, 4); return v; }

// The lattice point's hash, one multiply-xorshift round per coordinate.  The first round is a function of (seed, i) alone and the five
// corners of a simplex have i or i + 1 there: simplex4 evaluates the two once and picks (lattice_hash_first / lattice_hash_rest).
BLE_FN uint32_t lattice_hash_first(int i, uint32_t seed) {
  uint32_t h = (seed ^ (uint32_t)i) * 0x9E3779B1u;
  return h ^ (h >> 15);
}
BLE_FN uint32_t lattice_hash_rest(uint32_t h, int j, int k, int l) {
  h = (h ^ (uint32_t)j) * 0x85EBCA77u; h ^= h >> 13;
  h = (h ^ (uint32_t)k) * 0xC2B2AE3Du; h ^= h >> 16;
  h = (h ^ (uint32_t)l) * 0x27D4EB2Fu; h ^= h >> 15;
  return h;
}
BLE_FN uint32_t lattice_hash(int i, int j, int k, int l, uint32_t seed) { return lattice_hash_rest(lattice_hash_first(i, seed), j, k, l); }
// The 32 gradients -- the midpoints of the edges of the 4-cube: one zero component (g >> 3 says which), the other three +-1
// (bits 0, 1, 2 of g are the signs of the first, second, third non-zero component in x, y, z, w order) -- as a table of
// (wx, wy, wz, ww) weights: 512 B that every kernel which evaluates the noise keeps in LDS (grad_lut_fill).  Round 5: the dot
// product through the table is one 16-byte LDS read + a multiplication and three fused multiply-adds; picking the three components
// and their signs with compares, selects and sign-bit arithmetic was 16 vector instructions per corner, 80 per harmonic (of 333).
// A weight of +-1 is an exact product and the zero weight adds a zero: the value is the sum of the three signed components in
// x, y, z, w order.
constexpr int kGradLutFloats = 32 * 4;
BLE_FN void grad_lut_entry(int g, float* w4) {
  const int zero = g >> 3;
  int bit = 0;
  for (int comp = 0; comp < 4; ++comp) {
    if (comp == zero) { w4[comp] = 0.0f; continue; }
    w4[comp] = ((g >> bit) & 1) ? -1.0f : 1.0f;
    ++bit;
  }
}
BLE_FN void grad_lut_fill(float* lut, int tid, int n_threads) {      // followed by a barrier of the filling threads
  for (int g = tid; g < 32; g += n_threads) grad_lut_entry(g, lut + 4 * g);
}
// One corner: (0.6 - |d|^2)^4 * (gradient . d).  `lut`: the table above, 16-byte aligned.  The two sums are chains of EXPLICIT fused
// multiply-adds (4 + 4 instructions; as products and sums they were 8 + 7): written out, so that both kernels that inline this -- and
// the host build the tests compare with -- round alike.
BLE_FN float simplex_corner(float x, float y, float z, float w, uint32_t h, const float* lut) {
  BLE_NO_CONTRACT
  float t = f_fma(-w, w, f_fma(-z, z, f_fma(-y, y, f_fma(-x, x, 0.6f))));
  // a corner farther than sqrt(0.6) contributes nothing: t clamped to 0 makes its term (+-)0, which leaves the sum's bits alone -- no
  // branch (with 64 environments in a wave some lane always takes the other side)
  t = t < 0.0f ? 0.0f : t;
  const float* gw = static_cast<const float*>(__builtin_assume_aligned(lut + ((h >> 27) << 2), 16));
  const float dot = f_fma(gw[3], w, f_fma(gw[2], z, f_fma(gw[1], y, gw[0] * x)));
  t *= t;
  return t * t * dot;
}
BLE_FN float simplex4(float x, float y, float z, float w, uint32_t seed, const float* lut) {
  BLE_NO_CONTRACT
  const float F4 = 0.30901699437494745f, G4 = 0.1381966011250105f;
  const float s = (x + y + z + w) * F4;
  const int i = (int)floorf(x + s), j = (int)floorf(y + s), k = (int)floorf(z + s), l = (int)floorf(w + s);
  const float t = (float)(i + j + k + l) * G4;
  const float x0 = x - ((float)i - t), y0 = y - ((float)j - t), z0 = z - ((float)k - t), w0 = w - ((float)l - t);
  int rx = 0, ry = 0, rz = 0, rw = 0;
  if (x0 > y0) rx++; else ry++;
  if (x0 > z0) rx++; else rz++;
  if (x0 > w0) rx++; else rw++;
  if (y0 > z0) ry++; else rz++;
  if (y0 > w0) ry++; else rw++;
  if (z0 > w0) rz++; else rw++;
  const uint32_t h_i0 = lattice_hash_first(i, seed), h_i1 = lattice_hash_first(i + 1, seed);
  float n = simplex_corner(x0, y0, z0, w0, lattice_hash_rest(h_i0, j, k, l), lut);
#pragma unroll
  for (int c = 1; c <= 3; ++c) {
    const int th = 4 - c;                          // corner c adds 1 to the axes of rank >= 4 - c
    const bool di = rx >= th, dj = ry >= th, dk = rz >= th, dl = rw >= th;
    // the corner's offset from the cell's origin is c G4 - d per axis: one of two constants, picked (one select + one addition per axis)
    const float stay = (float)c * G4, move = (float)c * G4 - 1.0f;
    n += simplex_corner(x0 + (di ? move : stay), y0 + (dj ? move : stay), z0 + (dk ? move : stay), w0 + (dl ? move : stay),
                        lattice_hash_rest(di ? h_i1 : h_i0, j + (int)dj, k + (int)dk, l + (int)dl), lut);
  }
  const float last = 4.0f * G4 - 1.0f;
  n += simplex_corner(x0 + last, y0 + last, z0 + last, w0 + last, lattice_hash_rest(h_i1, j + 1, k + 1, l + 1), lut);
  return 27.0f * n;
}

// NoisyWindHarmonic.reset (:97-114): the generator seed and the random offset of one harmonic, from the environment's
// Philox stream (10 harmonics per environment, in the order u0..u4, v0..v4).
struct HarmonicDraw { uint32_t hseed; float ox, oy, op, ot; };
BLE_FN HarmonicDraw harmonic_draw(Philox& g) {
  HarmonicDraw d;
  d.hseed = philox_u32(g);
  d.ox = (float)(2.0 * philox_uniform(g) - 1.0); d.oy = (float)(2.0 * philox_uniform(g) - 1.0);
  d.op = (float)(2.0 * philox_uniform(g) - 1.0); d.ot = (float)(2.0 * philox_uniform(g) - 1.0);
  return d;
}
// NoisyWindComponent.get_noise (:190-211): the weighted harmonics of one component, variance-adjusted.
struct NoiseAccumulator { float acc = 0.0f, wsum = 0.0f, w2sum = 0.0f; };
// NoisyWindHarmonic.get_noise (:116-146): one harmonic's value at a point -- the expensive part (one 4-D simplex evaluation);
// the four-wave form of the transition evaluates the ten of them on different wavefronts
BLE_FN float noise_harmonic_value(int comp, int h, const HarmonicDraw& d, float x_km, float y_km, float pressure, float t_h,
                                  const float* lut) {
  BLE_NO_CONTRACT
  const float magnitude = sqrtf(kNoiseVariance / kSimplex4Variance);
  const Harmonic hp = harmonic_params(comp, h);
  const HarmonicRcp hr = harmonic_rcp(comp, h);
  return magnitude * simplex4(f_div_const(x_km, hp.x_spacing, hr.x) + d.ox, f_div_const(y_km, hp.y_spacing, hr.y) + d.oy,
                              f_div_const(pressure, hp.p_spacing, hr.p) + d.op, f_div_const(t_h, hp.t_spacing, hr.t) + d.ot, d.hseed, lut);
}
BLE_FN void noise_accumulate(NoiseAccumulator& a, int comp, int h, float nz) {
  BLE_NO_CONTRACT
  const Harmonic hp = harmonic_params(comp, h);
  a.acc = f_fma(nz, hp.weight, a.acc); a.wsum += hp.weight; a.w2sum = f_fma(hp.weight, hp.weight, a.w2sum);
}
BLE_FN void noise_add_harmonic(NoiseAccumulator& a, int comp, int h, const HarmonicDraw& d, float x_km, float y_km, float pressure,
                               float t_h, const float* lut) {
  noise_accumulate(a, comp, h, noise_harmonic_value(comp, h, d, x_km, y_km, pressure, t_h, lut));
}
// the coordinates the harmonics are sampled at (units.Distance.km, timedelta_to_hours)
BLE_FN void noise_coords(float x_m, float y_m, int32_t elapsed_s, float* x_km, float* y_km, float* t_h) {
  BLE_NO_CONTRACT
  *x_km = x_m * 1e-3f; *y_km = y_m * 1e-3f; *t_h = (float)elapsed_s * (1.0f / 3600.0f);
}
BLE_FN HarmonicDraw harmonic_draw_from_rows(const uint32_t* rows, int64_t stride, int k) {     // rows 5 k .. 5 k + 4, `stride` words apart
  const uint32_t* row = rows + (int64_t)(5 * k) * stride;
  HarmonicDraw d;
  d.hseed = row[0]; d.ox = u32_bits_float(row[stride]); d.oy = u32_bits_float(row[2 * stride]); d.op = u32_bits_float(row[3 * stride]);
  d.ot = u32_bits_float(row[4 * stride]);
  return d;
}
BLE_FN float noise_finish(const NoiseAccumulator& a) {
  BLE_NO_CONTRACT
  return a.acc / a.wsum * sqrtf(a.wsum / a.w2sum);
}

// Both components at one point, the harmonics' seeds and offsets drawn on the spot (50 Philox draws).
BLE_FN void wind_noise(float x_m, float y_m, float pressure, int32_t elapsed_s, uint64_t seed, uint64_t env,
                                  uint32_t episode, const float* lut, float* u, float* v) {
  BLE_NO_CONTRACT
  Philox g = philox_init(seed ^ 0x5EEDF00Dull, env, episode);
  const float x_km = x_m * 1e-3f, y_km = y_m * 1e-3f, t_h = (float)elapsed_s * (1.0f / 3600.0f);
  float out[2];
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    NoiseAccumulator a;
#pragma unroll 1
    for (int h = 0; h < 5; ++h) {
      const HarmonicDraw d = harmonic_draw(g);
      noise_add_harmonic(a, comp, h, d, x_km, y_km, pressure, t_h, lut);
    }
    out[comp] = noise_finish(a);
  }
  *u = out[0]; *v = out[1];
}

// Both components from draws that lie in memory as 50 words `stride` apart (rows 5 k .. 5 k + 4 = (seed, ox, oy, op, ot) of harmonic
// k = 5 comp + h): the HBM cache below, or the copy ble_step_kernel<noise> keeps in LDS for the steps of a launch.
BLE_FN void wind_noise_from_rows(float x_m, float y_m, float pressure, int32_t elapsed_s, const uint32_t* rows, int64_t stride,
                                 const float* lut, float* u, float* v) {
  BLE_NO_CONTRACT
  float x_km, y_km, t_h;
  noise_coords(x_m, y_m, elapsed_s, &x_km, &y_km, &t_h);
  float out[2];
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    NoiseAccumulator a;
#pragma unroll 1
    for (int h = 0; h < 5; ++h)
      noise_add_harmonic(a, comp, h, harmonic_draw_from_rows(rows, stride, 5 * comp + h), x_km, y_km, pressure, t_h, lut);
    out[comp] = noise_finish(a);
  }
  *u = out[0]; *v = out[1];
}
// ... and from the ten harmonic values, evaluated elsewhere (`nz`: 10 floats `stride` apart, harmonic k = 5 comp + h)
BLE_FN void wind_noise_from_values(const float* nz, int64_t stride, float* u, float* v) {
  float out[2];
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    NoiseAccumulator a;
#pragma unroll 1
    for (int h = 0; h < 5; ++h) noise_accumulate(a, comp, h, nz[(int64_t)(5 * comp + h) * stride]);
    out[comp] = noise_finish(a);
  }
  *u = out[0]; *v = out[1];
}
// The draws of one environment into `dst` (50 words, `dst_stride` apart): from the HBM cache when it holds this (seed, episode)'s
// -- redrawn and stored there otherwise -- or straight from the Philox stream when there is no cache.
constexpr int kNoiseCacheRows = 53;
// `env`: the environment's row in this call's arrays (and in `cache`); `env_key` = env + the shard's offset: the index the Philox
// stream is keyed by, so that a sharded batch draws what the unsharded one does
BLE_FN void noise_draws_fetch(uint64_t seed, uint64_t env, uint64_t env_key, uint32_t episode, uint32_t* cache, int64_t n, uint32_t* dst,
                              int64_t dst_stride) {
  if (cache != nullptr) {
    uint32_t* mine = cache + env;
    const uint32_t k0 = episode + 1u, k1 = (uint32_t)seed, k2 = (uint32_t)(seed >> 32);
    if (mine[50 * n] == k0 && mine[51 * n] == k1 && mine[52 * n] == k2) {
#pragma unroll 1
      for (int r = 0; r < 50; ++r) dst[r * dst_stride] = mine[(int64_t)r * n];
      return;
    }
  }
  Philox g = philox_init(seed ^ 0x5EEDF00Dull, env_key, episode);
#pragma unroll 1
  for (int k = 0; k < 10; ++k) {
    const HarmonicDraw d = harmonic_draw(g);
    const uint32_t w0 = d.hseed, w1 = float_bits_u32(d.ox), w2 = float_bits_u32(d.oy), w3 = float_bits_u32(d.op), w4 = float_bits_u32(d.ot);
    uint32_t* o = dst + (int64_t)(5 * k) * dst_stride;
    o[0] = w0; o[dst_stride] = w1; o[2 * dst_stride] = w2; o[3 * dst_stride] = w3; o[4 * dst_stride] = w4;
    if (cache != nullptr) {
      uint32_t* cch = cache + env + (int64_t)(5 * k) * n;
      cch[0] = w0; cch[n] = w1; cch[2 * n] = w2; cch[3 * n] = w3; cch[4 * n] = w4;
    }
  }
  if (cache != nullptr) {
    cache[env + 50 * n] = episode + 1u; cache[env + 51 * n] = (uint32_t)seed; cache[env + 52 * n] = (uint32_t)(seed >> 32);
  }
}

// the generator of a fused rollout (struct ble_noise_gen of the ABI)
struct StepNoise { unsigned long long seed; const uint32_t* episode; uint32_t* harmonic_cache; long long env_offset; };

// The same with the draws kept in HBM between calls, as the reference keeps them in its NoisyWindHarmonic objects between
// resets: `cache` is [kNoiseCacheRows][n] 32-bit words (coalesced), rows 5 k .. 5 k + 4 = (seed, ox, oy, op, ot) of harmonic k
// = 5 comp + h, rows 50 .. 52 the key (episode + 1, seed lo, seed hi) the entry was drawn for; an entry drawn for another
// (seed, episode) -- or an all-zero, fresh one -- is redrawn and stored.  Same values as wind_noise(), bit for bit.
BLE_FN void wind_noise_cached(float x_m, float y_m, float pressure, int32_t elapsed_s, uint64_t seed, uint64_t env, uint64_t env_key,
                              uint32_t episode, uint32_t* cache, int64_t n, const float* lut, float* u, float* v) {
  BLE_NO_CONTRACT
  uint32_t* mine = cache + env;
  const uint32_t k0 = episode + 1u, k1 = (uint32_t)seed, k2 = (uint32_t)(seed >> 32);
  if (!(mine[50 * n] == k0 && mine[51 * n] == k1 && mine[52 * n] == k2)) {
    Philox g = philox_init(seed ^ 0x5EEDF00Dull, env_key, episode);
#pragma unroll 1
    for (int k = 0; k < 10; ++k) {
      const HarmonicDraw d = harmonic_draw(g);
      uint32_t* row = mine + (int64_t)(5 * k) * n;
      row[0] = d.hseed; row[n] = float_bits_u32(d.ox); row[2 * n] = float_bits_u32(d.oy); row[3 * n] = float_bits_u32(d.op);
      row[4 * n] = float_bits_u32(d.ot);
    }
    mine[50 * n] = k0; mine[51 * n] = k1; mine[52 * n] = k2;
  }
  wind_noise_from_rows(x_m, y_m, pressure, elapsed_s, mine, n, lut, u, v);
}

}  // namespace ble
