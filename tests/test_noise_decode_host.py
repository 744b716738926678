"""CPU-side checks of the wind-noise and decoder-tail arithmetic through a host (g++) build of the
kernel headers (tests/emul, test tooling): the decoder tail against oracle/vae_oracle.py, the noise
against the statistics the reference prescribes (simplex_wind_noise.py:66-79: variance 1.02 per
component; SIMPLEX_VARIANCE 0.088392 for the raw simplex primitive).  The GPU twin of these
tests (tests/test_gpu_noise.py, tests/test_gpu_generative.py) checks device == host build."""
import ctypes

import numpy as np

import vae_oracle
from emul import emul


def _f(a): return np.ascontiguousarray(a, np.float32)
def _p(a): return a.ctypes.data_as(ctypes.c_void_p)


def simplex4(pts, seed):
  out = np.empty(len(pts), np.float32)
  cols = [_f(pts[:, k]) for k in range(4)]
  emul.lib().emul_simplex4(ctypes.c_int64(len(pts)), *[_p(c) for c in cols], ctypes.c_uint32(seed), _p(out))
  return out.astype(np.float64)


def wind_noise(x, y, p, t, seed, episode=None):
  n = len(x)
  out = np.empty((n, 2), np.float32)
  ep = None if episode is None else np.ascontiguousarray(episode, np.uint32)
  xs, ys, ps, ts = _f(x), _f(y), _f(p), np.ascontiguousarray(t, np.int32)
  emul.lib().emul_wind_noise(ctypes.c_int64(n), _p(xs), _p(ys), _p(ps), _p(ts), ctypes.c_uint64(seed),
                             None if ep is None else _p(ep), _p(out))
  return out.astype(np.float64)


def test_decoder_tail_host_build_matches_oracle():
  rng = np.random.default_rng(0)
  flow = _f(rng.standard_normal((3, 4410)) * 25)
  grid = np.empty((3, 21, 21, 10, 9, 2), np.float32)
  emul.lib().emul_decode_flow(ctypes.c_int64(3), _p(flow), _p(grid))
  want = vae_oracle.decode_flow(flow)
  np.testing.assert_allclose(grid, want, rtol=0, atol=1e-5 * np.abs(want).max())
  # resize weights of the oracle == the kernel's two-tap form
  w = vae_oracle.resize_weights()
  assert np.allclose(w.sum(1), 1.0) and (np.count_nonzero(w, axis=1) <= 2).all()
  assert w[0, 0] == 1.0 and w[-1, -1] == 1.0                  # edges: renormalised == clamped


def test_simplex_primitive_statistics():
  rng = np.random.default_rng(1)
  pts = rng.uniform(-30, 30, (200000, 4))
  v = simplex4(pts, 4242)
  assert np.abs(v).max() <= 1.05 and abs(v.mean()) < 6e-3
  assert abs(v.var() - 0.088392) < 0.03 * 0.088392            # SIMPLEX_VARIANCE, simplex_wind_noise.py:70
  moved = pts.copy(); moved[:, 2] += 1e-3
  assert np.abs(simplex4(moved, 4242) - v).max() < 2e-2       # continuous
  assert abs(np.corrcoef(v, simplex4(pts, 4243))[0, 1]) < 0.03


def test_wind_noise_structure():
  rng = np.random.default_rng(2)
  n = 100000
  x, y = rng.uniform(-2e5, 2e5, n), rng.uniform(-2e5, 2e5, n)
  p, t = rng.uniform(5000, 14000, n), rng.integers(0, 48 * 3600, n)
  uv = wind_noise(x, y, p, t, seed=7)
  for c in range(2):
    assert abs(uv[:, c].mean()) < 0.03 and 0.85 < uv[:, c].var() < 1.25        # target 1.02 (m/s)^2
  assert abs(np.corrcoef(uv[:, 0], uv[:, 1])[0, 1]) < 0.03
  # one field per (seed, env index, episode): same inputs -> same value; another episode -> another field
  again = wind_noise(x, y, p, t, seed=7)
  assert np.array_equal(uv, again)
  other = wind_noise(x, y, p, t, seed=7, episode=np.ones(n, np.uint32))
  assert abs(np.corrcoef(uv[:, 0], other[:, 0])[0, 1]) < 0.03
  # along one balloon's path the field is smooth: same env index 0, positions 100 m apart
  m = 2000
  path = wind_noise(np.full(1, 1000.0), np.zeros(1), np.full(1, 9000.0), np.zeros(1, np.int32), seed=3)
  xs = 1000.0 + 100.0 * np.arange(m)
  vals = np.array([wind_noise([xx], [0.0], [9000.0], [0], seed=3)[0] for xx in xs[:200]])
  assert np.array_equal(vals[0], path[0])
  assert np.abs(np.diff(vals[:, 0])).max() < 0.2


def test_decoder_resize_matches_pytorch_half_pixel_bilinear():
  """The decoder's 7 x 7 -> 23 x 23 resize (generative/vae.py:157-160: jax.image.resize(..., 'linear')) is half-pixel
  linear upsampling with edge-clamped taps.  jax is absent, so the oracle's restatement is anchored on an INDEPENDENT
  implementation of the same published operator: torch.nn.functional.interpolate(mode='bilinear',
  align_corners=False) (no antialiasing; for upsampling jax's default antialias has no effect).  This does not pin jax
  itself (row f3 stays "parity unpinned"): it shows that the restatement is the operator it claims to be."""
  import torch
  import vae_oracle
  rng = np.random.default_rng(11)
  psi = rng.standard_normal((3, 7, 7, 90))
  w = vae_oracle.resize_weights()
  ours = np.einsum('ai,bj,nijf->nabf', w, w, psi)
  t = torch.from_numpy(psi).permute(0, 3, 1, 2)                       # [n, f, 7, 7], float64
  theirs = torch.nn.functional.interpolate(t, size=(23, 23), mode='bilinear', align_corners=False).permute(0, 2, 3, 1).numpy()
  assert np.abs(ours - theirs).max() < 1e-12
  # and the separable weights are a partition of unity with at most two taps per output pixel
  assert np.allclose(w.sum(1), 1.0) and ((w > 0).sum(1) <= 2).all()
