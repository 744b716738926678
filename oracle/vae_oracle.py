"""CPU restatement of the wind-field VAE decoder (generative/vae.py:140-186), NumPy float64.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED against the reference: jax / flax are absent and the
reference's own test (generative/vae_test.py:43-54) asserts only the output shape.  The resize
follows the published semantics of jax.image.resize(method='linear') for upsampling: sample
position x = (i + 0.5) * in / out - 0.5, triangle-kernel weights over the existing input pixels,
renormalised (which equals clamping the two taps at the border).
"""
import numpy as np


def resize_weights(n_in=7, n_out=23):
  w = np.zeros((n_out, n_in))
  for i in range(n_out):
    x = (i + 0.5) * n_in / n_out - 0.5
    for k in range(n_in):
      w[i, k] = max(0.0, 1.0 - abs(x - k))
    w[i] /= w[i].sum()
  return w


def decode_flow(flow):
  """flow [n, 4410] -> [n, 21, 21, 10, 9, 2]  (vae.py:149-186)."""
  n = flow.shape[0]
  psi = np.asarray(flow, np.float64).reshape(n, 7, 7, 90)
  w = resize_weights()
  big = np.einsum('ai,bj,nijf->nabf', w, w, psi)                       # (n, 23, 23, 90)
  dy = (np.roll(big, -1, axis=1) - np.roll(big, 1, axis=1)) / 2.0
  dx = (np.roll(big, -1, axis=2) - np.roll(big, 1, axis=2)) / 2.0
  u = dy[:, 1:-1, 1:-1, :].reshape(n, 21, 21, 10, 9)
  v = -dx[:, 1:-1, 1:-1, :].reshape(n, 21, 21, 10, 9)
  return np.stack([u, v], axis=-1)


def mlp(latents, params):
  z = np.asarray(latents, np.float64)
  for k, (w, b) in enumerate(params):
    z = z @ np.asarray(w, np.float64) + np.asarray(b, np.float64)
    if k < 3:
      z = np.maximum(z, 0.0)
  return z
