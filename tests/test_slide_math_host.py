"""The algebra of the observation kernel's window slide (csrc/ble_observe.h, "GP slide"), restated in NumPy and
checked against fresh factorisations -- no GPU: dropping the oldest observation from K + noise = Lt D Lt^T is the rank-1
update K22 = L22 D2 L22^T + d1 l21 l21^T (Gill-Golub-Murray-Saunders C1) with p = L22^-1 l21 known from the previous
sweep; every row is two independent recurrences (forwards from l21, backwards from the diagonal because l21 = L22 p);
zeta = Lt^-1 y slides with one prefix sum; the newest observation is a bordering row."""
import numpy as np


def _ldl(K):
  L = np.linalg.cholesky(K)
  d = np.diag(L) ** 2
  return L / np.diag(L)[None, :], d          # unit lower, diagonal


def _kernel(x):
  ls = np.array([357000.0, 357000.0, 326.0, 34560.0])
  dd = (x[:, None, :] - x[None, :, :]) / ls
  return 3.6 ** 2 * np.exp(-np.sqrt((dd * dd).sum(-1))) + 0.05 * np.eye(len(x))


def _window(rng, n):
  t = np.arange(n) * 180.0
  return np.stack([np.cumsum(rng.normal(0, 900, n)), np.cumsum(rng.normal(0, 900, n)), 9000 + np.cumsum(rng.normal(0, 30, n)), t], 1)


def slide(Lt, d, p, zeta_over_d):
  """One drop, as the kernel does it.  Lt, d: factor of the (n + 1)-window; p = L22^-1 l21; returns the factor of the
  trailing n-window and its zeta / d."""
  n = len(d) - 1
  L22, d2, l21, d1 = Lt[1:, 1:], d[1:], Lt[1:, 0], d[0]
  t = p * p / d2
  gamma = 1.0 / d1 + np.cumsum(t)
  gamma_prev = gamma - t
  beta = p / (d2 * gamma)
  d_new = d2 * gamma / gamma_prev
  L_new = np.eye(n)
  for r in range(n):                      # lane = row
    m = (r + 1) // 2
    w = l21[r]                            # forwards: w^(0) = l21_r
    for k in range(m):
      w = w - p[k] * L22[r, k]
      L_new[r, k] = L22[r, k] + beta[k] * w
    w = p[r]                              # backwards: w^(r) = p_r
    for k in range(r - 1, m - 1, -1):
      L_new[r, k] = L22[r, k] + beta[k] * w
      w = w + p[k] * L22[r, k]
  # zeta' = T^-1 (zeta[1:] + y_0 p):  zeta'_i = b_i - p_i u_i / gamma_{i-1},  u = exclusive prefix sum of p b / d
  zeta = zeta_over_d * d
  b = zeta[1:] + zeta[0] * p
  c = p * b / d2
  u = np.cumsum(c) - c
  zeta_new = b - p * u / gamma_prev
  return L_new, d_new, zeta_new / d_new


def test_slide_equals_fresh_factorisation_over_many_steps():
  rng = np.random.default_rng(5)
  n = 40
  x = _window(rng, n + 60)
  y = rng.normal(0, 1.5, (n + 60, 2))
  Lt, d = _ldl(_kernel(x[:n]))
  zod = np.linalg.solve(Lt, y[:n]) / d[:, None]
  p = -np.linalg.solve(Lt, np.eye(n)[:, 0])[1:]                   # minus the first column of Lt^-1, below its head
  for s in range(60):
    # drop the oldest observation ...
    L2, d2, zu = slide(Lt, d, p, zod[:, 0])
    _, _, zv = slide(Lt, d, p, zod[:, 1])
    # ... and border the newest: omega = L^-1 k_new, r = omega / d, d_new = k_nn - omega . r, zeta_last = y - r . zeta
    xs = x[s + 1:s + n + 1]
    K = _kernel(xs)
    omega = np.linalg.solve(L2, K[:-1, -1])
    r = omega / d2
    Lt = np.eye(n); Lt[:-1, :-1] = L2; Lt[-1, :-1] = r
    d = np.concatenate([d2, [K[-1, -1] - omega @ r]])
    z_prev = np.stack([zu, zv], 1) * d2[:, None]
    z_last = y[s + n] - r @ z_prev
    zod = np.concatenate([z_prev, z_last[None, :]], 0) / d[:, None]
    p = -np.linalg.solve(Lt, np.eye(n)[:, 0])[1:]
    Lf, df = _ldl(K)
    assert np.abs(Lt - Lf).max() < 1e-10 and np.abs(d - df).max() / df.max() < 1e-11
    zf = np.linalg.solve(Lf, y[s + 1:s + n + 1]) / df[:, None]
    assert np.abs(zod - zf).max() / np.abs(zf).max() < 1e-9


def test_top_padding_changes_nothing():
  """The sweep pads the factor to 16-row tiles with identity rows at the TOP: solving against the padded unit-lower
  matrix returns the unpadded solution below the padding (and zeros in it)."""
  rng = np.random.default_rng(2)
  n, pad = 23, 9
  Lt, _ = _ldl(_kernel(_window(rng, n)))
  B = rng.normal(size=(n, 5))
  Lp = np.eye(n + pad); Lp[pad:, pad:] = Lt
  Bp = np.zeros((n + pad, 5)); Bp[pad:] = B
  Vp = np.linalg.solve(Lp, Bp)
  assert np.abs(Vp[:pad]).max() == 0.0 and np.abs(Vp[pad:] - np.linalg.solve(Lt, B)).max() < 1e-12
