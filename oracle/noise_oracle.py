"""CPU restatement of the wind-noise term (csrc/ble_noise.h), NumPy.  TEST INFRASTRUCTURE ONLY.

Two layers, pinned separately:

* The COMPOSITION -- five harmonics per wind component, their weights and spacings, the random offset of the simplex
  grid, NOISE_MAGNITUDE, the variance adjustment (reference env/simplex_wind_noise.py:50-76,116-146,180-211;
  env/wind_field.py:187-218).  This is the reference's own Python and IS pinned: tests/golden/f14_wind_noise.npz holds the
  outputs of the reference's `SimplexWindNoise.get_wind_noise` / `WindField.get_ground_truth`, imported unmodified, with its
  `opensimplex.OpenSimplex` dependency replaced by a stand-in whose `noise4d` is `simplex4` below and with recorded
  generator seeds / offsets (tests/golden/make_golden.py::f14_wind_noise; tests/test_oracle_golden.py holds
  `wind_noise` to it at 1e-12).

* The PRIMITIVE `simplex4` -- Gustavson 4-D simplex noise with hashed gradients, the kernel's own (csrc/ble_noise.h).  It is
  NOT `opensimplex==0.3`'s `noise4d` (reference requirements.txt:43; source absent from the image, no reference test pins
  its values): PARITY UNPINNED for the primitive, by construction.  What follows from that and is stated, not hidden: the
  reference normalises with OPENSIMPLEX_VARIANCE = 0.0569 (simplex_wind_noise.py:71,76), the variance of ITS primitive;
  this primitive's variance is the reference's other constant, SIMPLEX_VARIANCE = 0.088392 (:70; measured 0.0889), so the
  magnitude that yields the reference's target variance of 1.02 (m/s)^2 per harmonic is sqrt(1.02 / 0.088392).
  `MAGNITUDE_RATIO` converts between the two normalisations.
"""
import math

import numpy as np

# simplex_wind_noise.py:50-64 (weight, x km, y km, pressure Pa, time h)
U_HARMONICS = ((0.1445, 702.269, 2116.987, 2587.802, 245.0), (0.2766, 1483.570, 752.124, 646.208, 16.39),
               (0.2627, 276.810, 147.040, 587.702, 3.836), (0.2137, 10214.525, 1512.216, 965.629, 41.780),
               (0.1025, 181.286, 420.942, 8500.0, 245.0))
V_HARMONICS = ((0.2716, 1974.228, 2028.814, 713.697, 26.435), (0.2684, 699.738, 541.845, 632.116, 9.530),
               (0.2348, 217.750, 196.522, 686.825, 3.546), (0.1186, 47.500, 43.048, 66.553, 8.424),
               (0.1066, 3663.291, 232.023, 7499.741, 225.0))
# The version of simplex4's BIT PATTERN (== include/ble_abi.h::BLE_NOISE_PRIMITIVE_VERSION): bumped with every change of the primitive's
# values, here and in csrc/ble_noise.h together; tests/golden/f14_wind_noise.npz records the version it was generated with
# (tests/test_host_api.py holds the four to one number, tests/test_golden_reproducible.py the fixture to its generator).
PRIMITIVE_VERSION = 2
SIMPLEX4_VARIANCE = 0.088392          # of simplex4 below == simplex_wind_noise.py:70 SIMPLEX_VARIANCE
OPENSIMPLEX_VARIANCE = 0.0569         # simplex_wind_noise.py:71: of the reference's (absent) primitive
NOISE_VARIANCE = 1.02                 # simplex_wind_noise.py:76
MAGNITUDE = math.sqrt(NOISE_VARIANCE / SIMPLEX4_VARIANCE)                   # what the kernel multiplies simplex4 by
REFERENCE_MAGNITUDE = math.sqrt(NOISE_VARIANCE / OPENSIMPLEX_VARIANCE)      # simplex_wind_noise.py:76 NOISE_MAGNITUDE
MAGNITUDE_RATIO = MAGNITUDE / REFERENCE_MAGNITUDE                           # kernel output / reference output, same primitive


def _u32(a):
  return np.asarray(a).astype(np.int64).astype(np.uint32)      # two's complement wrap of negative lattice indices


def lattice_hash(i, j, k, l, seed):
  """csrc/ble_noise.h::lattice_hash, uint32 arithmetic."""
  with np.errstate(over='ignore'):
    h = np.broadcast_to(np.uint32(seed), np.shape(i)).astype(np.uint32)
    for v, mul, sh in ((i, 0x9E3779B1, 15), (j, 0x85EBCA77, 13), (k, 0xC2B2AE3D, 16), (l, 0x27D4EB2F, 15)):
      h = (h ^ _u32(v)) * np.uint32(mul)
      h = h ^ (h >> np.uint32(sh))
  return h


def _corner(x, y, z, w, h, dt):
  t = dt(0.6) - x * x - y * y - z * z - w * w            # (the kernel: four fused multiply-adds in this order)
  g = (h >> np.uint32(27)).astype(np.int64)
  zero = g >> 3
  a = np.where(zero == 0, y, x); b = np.where(zero <= 1, z, y); c = np.where(zero <= 2, w, z)
  a = np.where(g & 1, -a, a); b = np.where(g & 2, -b, b); c = np.where(g & 4, -c, c)
  t2 = t * t
  return np.where(t < 0, dt(0.0), t2 * t2 * (a + b + c))


def simplex4(x, y, z, w, seed, dtype=np.float64):
  """csrc/ble_noise.h::simplex4 (the kernel evaluates it in float32: dtype=np.float32 mirrors that operation for
  operation, up to the compiler's fused multiply-adds)."""
  dt = np.dtype(dtype).type
  x, y, z, w = (np.asarray(v, dtype) for v in (x, y, z, w))
  f4, g4 = dt(0.30901699437494745), dt(0.1381966011250105)
  s = (x + y + z + w) * f4
  i, j, k, l = (np.floor(v + s).astype(np.int64) for v in (x, y, z, w))
  t = (i + j + k + l).astype(dtype) * g4
  x0, y0, z0, w0 = x - (i.astype(dtype) - t), y - (j.astype(dtype) - t), z - (k.astype(dtype) - t), w - (l.astype(dtype) - t)
  rx = (x0 > y0).astype(np.int64) + (x0 > z0) + (x0 > w0)
  ry = (~(x0 > y0)).astype(np.int64) + (y0 > z0) + (y0 > w0)
  rz = (~(x0 > z0)).astype(np.int64) + ~(y0 > z0) + (z0 > w0)
  rw = (~(x0 > w0)).astype(np.int64) + ~(y0 > w0) + ~(z0 > w0)
  n = _corner(x0, y0, z0, w0, lattice_hash(i, j, k, l, seed), dt)
  for c in (1, 2, 3):
    th = 4 - c
    di, dj, dk, dl = (r >= th for r in (rx, ry, rz, rw))
    stay = dt(c) * g4; move = dt(c) * g4 - dt(1.0)      # the corner's offset per axis: one of two constants (ble_noise.h)
    n = n + _corner(x0 + np.where(di, move, stay), y0 + np.where(dj, move, stay), z0 + np.where(dk, move, stay),
                    w0 + np.where(dl, move, stay), lattice_hash(i + di, j + dj, k + dk, l + dl, seed), dt)
  last = dt(4.0) * g4 - dt(1.0)
  n = n + _corner(x0 + last, y0 + last, z0 + last, w0 + last, lattice_hash(i + 1, j + 1, k + 1, l + 1, seed), dt)
  return dt(27.0) * n


def wind_noise(x_m, y_m, pressure, elapsed_s, seeds, offsets, dtype=np.float64, magnitude=MAGNITUDE):
  """(u, v) noise [m/s] at the points, float64 result.

  seeds [2][5] (generator seed per component and harmonic), offsets [2][5][4] (x, y, pressure, time).  dtype is the
  precision of the COORDINATES and of the primitive (np.float64: the reference's arithmetic; np.float32: the kernel's).
  NoisyWindHarmonic.get_noise (simplex_wind_noise.py:116-146) and NoisyWindComponent.get_noise (:180-211)."""
  dt = np.dtype(dtype).type
  if dtype == np.float32:       # the kernel's own coordinate arithmetic (csrc/ble_noise.h::wind_noise_cached)
    x_km = np.asarray(x_m, np.float32) * np.float32(1e-3); y_km = np.asarray(y_m, np.float32) * np.float32(1e-3)
    t_h = np.asarray(elapsed_s).astype(np.float32) * np.float32(1.0 / 3600.0)
  else:                         # units.Distance.km, units.timedelta_to_hours
    x_km = np.asarray(x_m, np.float64) / 1000.0; y_km = np.asarray(y_m, np.float64) / 1000.0
    t_h = np.asarray(elapsed_s, np.float64) / 3600.0
  p = np.asarray(pressure, dtype)
  out = []
  for comp, table in enumerate((U_HARMONICS, V_HARMONICS)):
    acc = np.zeros(np.shape(x_km), np.float64); wsum = 0.0; w2sum = 0.0
    for h, (weight, xs, ys, ps, ts) in enumerate(table):
      ox, oy, op, ot = (dt(v) for v in offsets[comp][h])
      nz = simplex4(x_km / dt(xs) + ox, y_km / dt(ys) + oy, p / dt(ps) + op, t_h / dt(ts) + ot, int(seeds[comp][h]), dtype)
      acc = acc + magnitude * nz.astype(np.float64) * weight
      wsum += weight; w2sum += weight ** 2
    out.append(acc / wsum * math.sqrt(wsum / w2sum))
  return np.stack(out, axis=-1)
