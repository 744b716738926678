"""GenerativeWindFieldSampler (env/generative_wind_field.py:40-62) and the VAE decoder
(generative/vae.py:140-186) on the device.

The trained parameters (`models.load_offlineskies22`) are not part of the reference checkout
(/root/reference/.MISSING_LARGE_BLOBS), so the sampler carries *synthetic* weights: same
architecture (64 -> 1000 -> 1000 -> 1000 -> 4410, ReLU), seeded LeCun-normal kernels (flax's Dense
default) scaled so that the winds come out at a few m/s.  Output parity with the reference
decoder is therefore unpinned beyond shape and algorithm; the arithmetic after the
last Dense layer is checked against oracle/vae_oracle.py.

The four Dense layers are plain GEMMs (torch.matmul -> rocBLAS/hipBLASLt); resize + central
differences + layout is `ble_decode_flow_fields_f32`.  `decode(latents)` produces one grid per
latent directly in HBM: per-env forecasts for BASELINE config 5 (32 768 envs x 317.5 KB = 10.4 GB).
"""
import datetime as dt
import math
from typing import Optional

import numpy as np
import torch

from balloon_learning_environment_amd import _lib
from balloon_learning_environment_amd import device as dev
from balloon_learning_environment_amd.env import grid_wind_field_sampler

NUM_LATENTS = 64
HIDDEN = 1000
FLOW_UNITS = 7 * 7 * 90
GRID_FLOATS = 21 * 21 * 10 * 9 * 2


def synthetic_decoder_params(seed: int = 0, output_gain: float = 40.0):
  """[(kernel [in, out], bias [out])] x 4, float32 NumPy: LeCun-normal kernels, zero biases."""
  rng = np.random.default_rng(seed)
  dims = [NUM_LATENTS, HIDDEN, HIDDEN, HIDDEN, FLOW_UNITS]
  params = []
  for k, (a, b) in enumerate(zip(dims, dims[1:])):
    gain = output_gain if k == 3 else math.sqrt(2.0)      # keep activations O(1) through the ReLUs
    params.append(((rng.standard_normal((a, b)) * gain / math.sqrt(a)).astype(np.float32), np.zeros(b, np.float32)))
  return params


class GenerativeWindFieldSampler(grid_wind_field_sampler.GridWindFieldSampler):
  def __init__(self, params=None, device='cuda:0', seed: int = 0):
    self.device = dev.require_gpu(device)
    self._lib = _lib.lib()
    params = synthetic_decoder_params(seed) if params is None else params
    self.params = [(torch.from_numpy(np.ascontiguousarray(w, np.float32)).to(self.device),
                    torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(self.device)) for w, b in params]
    self._shape = grid_wind_field_sampler.FieldShape()

  @property
  def field_shape(self):
    return self._shape

  @dev.on_own_device
  def flow_fields(self, latents: torch.Tensor) -> torch.Tensor:
    """[n, 64] -> [n, 4410]: the MLP part of Decoder.__call__ (vae.py:145-148)."""
    z = latents.to(self.device, torch.float32)
    for k, (w, b) in enumerate(self.params):
      z = torch.addmm(b, z, w)
      if k < 3:
        z = torch.relu_(z)
    return z.contiguous()

  @dev.on_own_device
  def decode(self, latents: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[n, 64] latents -> [n, 21, 21, 10, 9, 2] float32 wind grids on the device."""
    n = latents.shape[0]
    if out is None:
      out = torch.empty((n, 21, 21, 10, 9, 2), dtype=torch.float32, device=self.device)
    assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == n * GRID_FLOATS
    flat = out.view(n, GRID_FLOATS)
    for lo in range(0, n, 32768):                 # slices bound the [n, 1000] activations and gridDim.y
      hi = min(n, lo + 32768)
      flow = self.flow_fields(latents[lo:hi])
      _lib.check(self._lib.ble_decode_flow_fields_f32(flow.data_ptr(), flat[lo:hi].data_ptr(), hi - lo,
                                                      dev.stream_ptr(self.device)), 'ble_decode_flow_fields_f32')
      flow.record_stream(torch.cuda.current_stream(self.device))
    return out

  @dev.on_own_device
  def sample_latents(self, n: int, seed: int) -> torch.Tensor:
    gen = torch.Generator(device=self.device); gen.manual_seed(int(seed))
    return torch.randn((n, NUM_LATENTS), dtype=torch.float32, device=self.device, generator=gen)

  @dev.on_own_device
  def sample_latents_keyed(self, global_index: torch.Tensor, episode: torch.Tensor, seed: int) -> torch.Tensor:
    """[n, 64] standard-normal latents, row i a function of (seed, global_index[i], episode[i]) ALONE -- a counter-based stream
    (splitmix64 of the counter, Box-Muller) instead of one generator walked in batch order: the wind field of environment g in its
    e-th episode is the same whatever the batch it is decoded in, so the shards of a job (VecBalloonArena(env_offset=...)) fly in
    the fields the unsharded batch flies in, and a masked refresh draws what a full one would."""
    g = global_index.to(self.device, torch.int64).reshape(-1, 1)
    e = episode.to(self.device, torch.int64).reshape(-1, 1)

    def mix(z):                                   # splitmix64's finaliser on int64 tensors (two's-complement wrap = mod 2^64)
      z = (z ^ ((z >> 30) & 0x3FFFFFFFF)) * (-4658895280553007687)        # 0xBF58476D1CE4E5B9
      z = (z ^ ((z >> 27) & 0x1FFFFFFFFF)) * (-7723592293110705685)       # 0x94D049BB133111EB
      return z ^ ((z >> 31) & 0x1FFFFFFFF)
    k = torch.arange(NUM_LATENTS, dtype=torch.int64, device=self.device).reshape(1, -1)
    base = mix(mix(torch.full_like(g, int(seed) & 0x7FFFFFFFFFFFFFFF) + e * (-7046029254386353131)) + g)   # (0x9E3779B97F4A7C15)
    h1, h2 = mix(base + (2 * k + 1) * (-7046029254386353131)), mix(base + (2 * k + 2) * (-7046029254386353131))
    u1 = (((h1 >> 11) & ((1 << 53) - 1)).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)           # (0, 1)
    u2 = (((h2 >> 11) & ((1 << 53) - 1)).to(torch.float64)) * (1.0 / 9007199254740992.0)                 # [0, 1)
    return (torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)).to(torch.float32).contiguous()

  def sample_field(self, key, date_time: Optional[dt.datetime] = None) -> np.ndarray:
    seed = int(np.asarray(key).ravel()[-1]) if key is not None else 0
    return self.decode(self.sample_latents(1, seed))[0].cpu().numpy()


def generative_wind_field_factory(device='cuda:0'):
  """The reference's factory (env/generative_wind_field.py:35-37): a GridBasedWindField over the generative sampler."""
  from balloon_learning_environment_amd.env import grid_based_wind_field
  return grid_based_wind_field.GridBasedWindField(GenerativeWindFieldSampler(device=device), device)


def GenerativeWindField(device='cuda:0'):     # noqa: N802  (the name the reference's earlier releases exported)
  return generative_wind_field_factory(device)
