"""Container-type shims that let the reference's own arithmetic import in THIS container.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` to produce the
committed golden vectors.  Nothing in the product package, ``bench.py`` or the
``-m gpu`` tests imports this file, and it is inert wherever ``/root/reference``
is absent (the GPU box): ``available()`` returns False and ``install()`` raises.

What is shimmed: only *container / decorator / value-type*
third-party modules that are not installed here (s2sphere.LatLng, absl.logging,
gin.configurable, jax name stubs, transitions.Machine first-match dispatch) and SciPy's
removed ``interp2d`` (regular-grid linear case, the migration recipe SciPy documents).
Every reference ``.py`` file is imported unmodified from ``/root/reference``; no reference
source is copied.

Round 4, for fixtures F14 / F15 (the reference's own Python AROUND two absent third-party pieces):
  * ``opensimplex.OpenSimplex(seed).noise4d`` -> the kernel's own primitive restated in NumPy
    (oracle/noise_oracle.py::simplex4).  NOT opensimplex 0.3's values: the fixture pins what the
    reference does with the primitive, not the primitive.
  * ``jax.random.split / choice / uniform / normal`` -> recorded draws (NumPy SeedSequence of the key):
    the JAX threefry streams are not reproduced; the drawn seeds / offsets / latents are stored in
    the fixture.
  * ``flax.linen`` Module / Dense / relu / compact -> a 40-line functional stand-in (parameters looked up as
    flax names them: Dense_0 ... in call order; float64 accumulation).
  * ``jax.image.resize(method='linear')`` -> torch.nn.functional.interpolate(mode='bilinear',
    align_corners=False), an independent implementation of half-pixel linear resampling
    (upsampling: jax's antialias has no effect).  This operator is the stated ASSUMPTION of F15.
"""
import math
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = '/root/reference'
_PKG = 'balloon_learning_environment'


def available() -> bool:
  return os.path.isdir(os.path.join(REFERENCE_ROOT, _PKG))


class _Angle:
  __slots__ = ('radians',)

  def __init__(self, radians):
    self.radians = radians

  @property
  def degrees(self):
    return math.degrees(self.radians)


class _LatLng:
  """Value type standing in for s2sphere.LatLng (pinned s2sphere==0.2.5)."""

  def __init__(self, lat_rad, lng_rad):
    self._lat = lat_rad
    self._lng = lng_rad

  @classmethod
  def from_degrees(cls, lat, lng):
    return cls(math.radians(lat), math.radians(lng))

  @classmethod
  def from_radians(cls, lat, lng):
    return cls(lat, lng)

  def lat(self):
    return _Angle(self._lat)

  def lng(self):
    return _Angle(self._lng)

  @property
  def is_valid(self):
    return abs(self._lat) <= math.pi / 2 and abs(self._lng) <= math.pi

  def normalized(self):
    lat = max(-math.pi / 2, min(math.pi / 2, self._lat))
    return _LatLng(lat, math.remainder(self._lng, 2 * math.pi))

  def __repr__(self):
    return f'LatLng({math.degrees(self._lat)}, {math.degrees(self._lng)})'


class _Machine:
  """transitions.Machine stand-in: first matching transition in table order."""

  def __init__(self, states, transitions, initial):
    del states
    self.state = initial
    self._table = {}
    for t in transitions:
      self._table.setdefault(t['trigger'], []).append(t)
    for name in self._table:
      setattr(self, name, self._make_trigger(name))

  def _make_trigger(self, name):
    def fire():
      for t in self._table[name]:
        src = t['source']
        if src == '*' or src == self.state or (
            isinstance(src, (tuple, list)) and self.state in src):
          self.state = t['dest']
          return True
      raise RuntimeError(f'no transition for {name} from {self.state}')
    return fire


def _make_interp2d():
  import scipy.interpolate as si

  class interp2d:  # pylint: disable=invalid-name
    """Regular-grid, kind='linear', fill_value=None (nearest outside)."""

    def __init__(self, x, y, z, fill_value=None):
      del fill_value
      self.x = np.asarray(x, dtype=float)
      self.y = np.asarray(y, dtype=float)
      zz = np.asarray(z, dtype=float).reshape(len(self.y), len(self.x)).T
      self._spline = si.RectBivariateSpline(self.x, self.y, zz, kx=1, ky=1)

    def __call__(self, x, y):
      xc = np.clip(x, self.x[0], self.x[-1])
      yc = np.clip(y, self.y[0], self.y[-1])
      return np.array([self._spline(xc, yc)[0, 0]])

  return interp2d


class _StandInSimplex:
  """opensimplex.OpenSimplex stand-in for fixture F14: noise4d = oracle/noise_oracle.py::simplex4 (the kernel's primitive)."""

  def __init__(self, seed=0):
    self.seed = int(seed)

  def noise4d(self, x, y, z, w):
    import noise_oracle
    return float(noise_oracle.simplex4(np.float64(x), np.float64(y), np.float64(z), np.float64(w), self.seed & 0xFFFFFFFF))


def key_rng(key, salt):
  """The recorded stream behind a stand-in PRNG key (any integer array)."""
  words = [int(v) & 0xFFFFFFFF for v in np.asarray(key).ravel()]
  return np.random.default_rng(np.random.SeedSequence(words + [int(salt)]))


def _install_jax_random(jrandom):
  def split(key, num=2):
    return key_rng(key, 1).integers(0, 2 ** 32, size=(num, 2), dtype=np.uint32)

  def choice(key, a):
    return np.int64(key_rng(key, 2).integers(0, int(a)))

  def uniform(key, shape=()):
    # float32 VALUES (jax's default dtype) in a float64 container: under the reference's pinned NumPy 1.x a float32
    # scalar combined with a Python float is promoted to float64 (value-based casting), which NumPy 2 no longer does
    return key_rng(key, 3).random(shape).astype(np.float32).astype(np.float64)

  def normal(key, shape=()):
    return key_rng(key, 4).standard_normal(shape).astype(np.float32)
  jrandom.split, jrandom.choice, jrandom.uniform, jrandom.normal = split, choice, uniform, normal
  jrandom.PRNGKey = lambda seed: np.array([0, int(seed) & 0xFFFFFFFF], np.uint32)


def _resize_linear(image, shape, method='linear'):
  """jax.image.resize(method='linear') of an (h, w, c) array: see the module docstring (the assumption of F15)."""
  import torch
  assert method == 'linear' and len(shape) == 3 and shape[2] == np.shape(image)[2]
  t = torch.from_numpy(np.ascontiguousarray(np.asarray(image, np.float64))).permute(2, 0, 1)[None]
  r = torch.nn.functional.interpolate(t, size=(int(shape[0]), int(shape[1])), mode='bilinear', align_corners=False)
  return r[0].permute(1, 2, 0).numpy()


class _LinenScope:
  params = None
  counters = None


class _LinenDense:
  """flax.linen.Dense stand-in: y = x @ kernel + bias with the parameters flax would bind (Dense_<k> in call order)."""

  def __init__(self, features, name=None):
    self.features, self.name = int(features), name

  def __call__(self, x):
    name = self.name
    if name is None:
      k = _LinenScope.counters.get('Dense', 0)
      _LinenScope.counters['Dense'] = k + 1
      name = f'Dense_{k}'
    p = _LinenScope.params[name]
    kernel, bias = np.asarray(p['kernel'], np.float64), np.asarray(p['bias'], np.float64)
    assert kernel.shape[1] == self.features
    return np.asarray(x, np.float64) @ kernel + bias


class _LinenModule:
  """flax.linen.Module stand-in: dataclass-style fields stay class attributes; apply() binds {'params': {...}}."""

  def apply(self, variables, *args, **kwargs):
    prev = (_LinenScope.params, _LinenScope.counters)
    _LinenScope.params, _LinenScope.counters = variables['params'], {}
    try:
      return self(*args, **kwargs)
    finally:
      _LinenScope.params, _LinenScope.counters = prev


def install() -> None:
  """Installs the shims.  Call before importing any reference module."""
  if not available():
    raise RuntimeError('reference tree not present; golden vectors can only be '
                       'regenerated in the build container')
  if _PKG in sys.modules and getattr(sys.modules[_PKG], '_ble_shimmed', False):
    return
  pkg = types.ModuleType(_PKG)
  pkg.__path__ = [os.path.join(REFERENCE_ROOT, _PKG)]  # skips the gym-registering __init__
  pkg._ble_shimmed = True
  sys.modules[_PKG] = pkg

  s2 = types.ModuleType('s2sphere')
  s2.LatLng = _LatLng
  sys.modules['s2sphere'] = s2

  absl = types.ModuleType('absl')
  absl_logging = types.ModuleType('absl.logging')
  for name in ('warning', 'info', 'error', 'debug'):
    setattr(absl_logging, name, lambda *a, **k: None)
  absl.logging = absl_logging
  sys.modules['absl'] = absl
  sys.modules['absl.logging'] = absl_logging

  gin = types.ModuleType('gin')

  def configurable(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
      return args[0]
    return lambda f: f
  gin.configurable = configurable
  gin.REQUIRED = object()
  sys.modules['gin'] = gin

  jax = types.ModuleType('jax')
  jnp = types.ModuleType('jax.numpy')
  jrandom = types.ModuleType('jax.random')
  jnp.ndarray = np.ndarray
  jnp.int32 = np.int32
  jnp.float32 = np.float32
  jnp.pi = np.pi
  jnp.asarray = np.asarray

  def linspace(start, stop, num, dtype=np.float32):  # jnp default dtype is float32
    return np.linspace(start, stop, num).astype(dtype)
  jnp.linspace = linspace
  jax.numpy = jnp
  jax.random = jrandom
  jax.jit = lambda f, *a, **k: f
  sys.modules.update({'jax': jax, 'jax.numpy': jnp, 'jax.random': jrandom})

  osx = types.ModuleType('opensimplex')
  osx.OpenSimplex = _StandInSimplex   # NOT opensimplex 0.3 (absent, unpinned): the kernel's primitive, see the module docstring
  sys.modules['opensimplex'] = osx
  _install_jax_random(jrandom)
  jnp.roll, jnp.stack, jnp.sign, jnp.abs = np.roll, np.stack, np.sign, np.abs
  jimage = types.ModuleType('jax.image')
  jimage.resize = _resize_linear
  jax.image = jimage
  sys.modules['jax.image'] = jimage

  tr = types.ModuleType('transitions')
  tr.Machine = _Machine
  sys.modules['transitions'] = tr

  import scipy.interpolate as si
  if not hasattr(si, '_ble_orig_interp2d'):
    si._ble_orig_interp2d = getattr(si, 'interp2d', None)
    si.interp2d = _make_interp2d()

  # gym / flax are only needed for vae.FieldShape (dataclass) and features.py.
  gym = types.ModuleType('gym')
  spaces = types.ModuleType('gym.spaces')

  class Box:
    def __init__(self, low, high, dtype=np.float32, shape=None):
      self.low = np.asarray(low)
      self.high = np.asarray(high)
      self.shape = self.low.shape if shape is None else shape
      self.dtype = dtype

  class Discrete:
    def __init__(self, n):
      self.n = n
  spaces.Box = Box
  spaces.Discrete = Discrete
  gym.spaces = spaces
  gym.Env = object
  gym.Space = object
  sys.modules['gym'] = gym
  sys.modules['gym.spaces'] = spaces

  flax = types.ModuleType('flax')
  linen = types.ModuleType('flax.linen')

  linen.Module = _LinenModule
  linen.compact = lambda f: f
  linen.Dense = _LinenDense
  linen.relu = lambda x: np.maximum(x, 0.0)
  flax.linen = linen
  sys.modules['flax'] = flax
  sys.modules['flax.linen'] = linen
  # Import-only placeholders (never called by the golden generator): needed so that
  # env/balloon_env.py (reward function) and env/balloon_arena.py import.
  flax_metrics = types.ModuleType('flax.metrics')
  flax_tb = types.ModuleType('flax.metrics.tensorboard')
  flax_tb.SummaryWriter = object
  flax_metrics.tensorboard = flax_tb
  flax.metrics = flax_metrics
  flax.serialization = types.ModuleType('flax.serialization')
  sys.modules['flax.metrics'] = flax_metrics
  sys.modules['flax.metrics.tensorboard'] = flax_tb
  sys.modules['flax.serialization'] = flax.serialization
  tfp_root = types.ModuleType('tensorflow_probability')
  tfp_sub = types.ModuleType('tensorflow_probability.substrates')
  tfp_jax = types.ModuleType('tensorflow_probability.substrates.jax')
  tfp_sub.jax = tfp_jax
  tfp_root.substrates = tfp_sub
  sys.modules['tensorflow_probability'] = tfp_root
  sys.modules['tensorflow_probability.substrates'] = tfp_sub
  sys.modules['tensorflow_probability.substrates.jax'] = tfp_jax
  tf = types.ModuleType('tensorflow')
  sys.modules['tensorflow'] = tf


def make_atmosphere(alpha: float):
  """Atmosphere with a chosen alpha (bypasses jax.random.uniform in reset)."""
  from balloon_learning_environment.env.balloon import standard_atmosphere as sa
  atm = sa.Atmosphere.__new__(sa.Atmosphere)
  atm._lapse_rates = ((1 - alpha) * atm._LAPSE_RATES_LOW +
                      alpha * atm._LAPSE_RATES_HIGH)
  atm._initialize_temperature_transitions()
  atm._initialize_pressure_transitions()
  return atm


def make_grid_wind_field(field: np.ndarray):
  """GridBasedWindField over `field` without constructing SimplexWindNoise."""
  from balloon_learning_environment.env import grid_based_wind_field as gbwf
  from balloon_learning_environment.generative import vae
  wf = gbwf.GridBasedWindField.__new__(gbwf.GridBasedWindField)
  wf._wind_field_sampler = None
  wf.field_shape = vae.FieldShape()
  wf.field = field
  wf._grid = (
      np.asarray(wf.field_shape.latlng_grid_points()),
      np.asarray(wf.field_shape.latlng_grid_points()),
      np.asarray(wf.field_shape.pressure_grid_points()),
      np.asarray(wf.field_shape.time_grid_points()))
  return wf
