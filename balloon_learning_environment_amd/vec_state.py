"""Batched simulator state held as device tensors + the stepping call into libble_hip.so.

`VecSimulator` is the device-side counterpart of N reference `Balloon` objects
(env/balloon/balloon.py:253-328) with their `Atmosphere` alphas and one shared (or
per-env) wind grid.  It owns tensors only; all arithmetic happens in the HIP library.
"""
import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from balloon_learning_environment_amd import _abi
from balloon_learning_environment_amd import _lib
from balloon_learning_environment_amd import device as dev

GRID_SHAPE = (21, 21, 10, 9, 2)  # generative/vae.py:30-38 FieldShape.grid_shape()
SUBSTEPS = 18                    # constants.AGENT_TIME_STEP (180 s) / 10 s stride
COUNT_SLOTS = 64                 # BLE_COUNT_SLOTS in include/ble_abi.h


class ReferenceError_(Exception):
  """Base for conditions on which the reference raises inside the transition."""


def raise_for_flags(flags: int) -> None:
  """Turns BLE_FLAG_* bits back into the exceptions the reference raises."""
  if flags & _lib.FLAG_PRESSURE_RANGE:
    raise AssertionError('Atmosphere.at_pressure: pressure out of range '
                         '(standard_atmosphere.py:126-127)')
  if flags & _lib.FLAG_ABSORPTIVITY:
    raise ValueError('total_absorptivity: Computed total absorptivity factor out of expected range [0, 1].')
  if flags & _lib.FLAG_SOLAR_RANGE:
    raise ValueError('solar_atmospheric_attenuation: Pressure altitude out of expected range [0, 101325] Pa.')
  if flags & _lib.FLAG_POWER_TABLE:
    raise AssertionError('power_table.lookup: pressure_ratio out of [0.99, 5]')
  if flags & _lib.FLAG_NONFINITE:
    raise FloatingPointError('non-finite balloon state')
  if flags & _lib.FLAG_PRESSURE_SEARCH:
    raise ValueError('Unable to find safe pressure for balloon.')       # pressure_range_builder.py:180-182
  if flags & _lib.FLAG_DAY_CYCLE:
    raise ZeroDivisionError('float division by zero')                     # features.py:432-437 at a station in polar night
  if flags & _lib.FLAG_GP_WINDOW:
    raise OverflowError('WindGP window holds more than 120 observations (agent steps shorter than 180 s)')


_on_own_device = dev.on_own_device


class VecSimulator:
  """N balloons on one GPU.  State tensors are exposed as attributes of `.state`."""

  def __init__(self, n: int, device='cuda:0', env_offset: int = 0):
    """env_offset: index of this simulator's environment 0 in the GLOBAL batch (a rank of a sharded run passes its shard's
    start): the device reset and the wind noise key their Philox streams by (seed, env_offset + i, episode), so the shards of
    a batch draw exactly what the unsharded batch draws -- one seed for the whole job, whatever the sharding."""
    self.device = dev.require_gpu(device)
    self.lib = _lib.lib()
    self.n = int(n)
    self.env_offset = int(env_offset)
    assert self.env_offset >= 0
    with torch.cuda.device(self.device):
      self.state: Dict[str, torch.Tensor] = {
          name: torch.zeros(self.n, dtype=dev.torch_dtype(_abi.FIELD_DTYPES[name]), device=self.device)
          for name in _abi.FIELD_NAMES}
      self.reward = torch.zeros(self.n, dtype=torch.float32, device=self.device)
      self.terminal = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
      self.effective_action = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
      self.err_flags = torch.zeros(1, dtype=torch.int32, device=self.device)
      self.active_slots = torch.zeros(COUNT_SLOTS, dtype=torch.int64, device=self.device)
      self.episode = torch.zeros(self.n, dtype=torch.int32, device=self.device)   # per-env episode counter
    self.grid: Optional[torch.Tensor] = None
    self.grid_env_stride = 0
    # per-episode derived constants (atmosphere transition pressures, station sin / cos, earth-IR heat): filled by the reset
    # kernel, re-derived by the step kernel itself wherever an entry does not match the constants above -- never stale
    with torch.cuda.device(self.device):
      self.episode_cache = torch.zeros(_abi.EPISODE_CACHE_ROWS, self.n, dtype=torch.float64, device=self.device)
    self._struct = dev.state_struct(self.state, self.episode_cache)
    self.vehicle: Dict[str, float] = {}     # the BalloonState vehicle fields that differ from the reference's defaults (set_vehicle)
    self._noise_cache = None        # per-episode draws of the wind noise's harmonics (allocated by the first wind_noise())
    self._noise_gens = []           # the ble_noise_gen structs handed out (prepared launches hold them): load_state_dict re-keys them
    self._gp = None                 # WindGP history ring (allocated by the first observe())
    self._obs_reset = None          # envs whose history must restart at the next observe()

  # ------------------------------------------------------------------ data in / out
  def set_state(self, arrays: Dict[str, np.ndarray]) -> None:
    """Copies host arrays (any float/int dtype) into the device state."""
    for name in _abi.FIELD_NAMES:
      if name in arrays:
        a = np.ascontiguousarray(np.asarray(arrays[name]).astype(_abi.FIELD_DTYPES[name]))
        assert a.shape == (self.n,), (name, a.shape)
        self.state[name].copy_(torch.from_numpy(a))

  def get_state(self) -> Dict[str, np.ndarray]:
    return {name: t.cpu().numpy() for name, t in self.state.items()}

  @_on_own_device
  def rows(self, first: int = 0, count: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Environments first .. first + count - 1 as records: a [count, 26] float64 DEVICE tensor, the per-environment members of
    ble_state_f32 in the struct's order (_abi.FIELD_NAMES), every value exact (`ble_state_rows_f64`) -- one transfer for a host
    consumer instead of one per member.  Asynchronous on the current stream."""
    count = self.n - first if count is None else int(count)
    if out is None:
      out = torch.empty(count, _lib.ROW_DOUBLES, dtype=torch.float64, device=self.device)
    assert out.dtype == torch.float64 and out.is_contiguous() and out.numel() == count * _lib.ROW_DOUBLES
    _lib.check(self.lib.ble_state_rows_f64(ctypes.byref(self._struct), int(first), count, out.data_ptr(), self.n, dev.stream_ptr(self.device)),
               'ble_state_rows_f64')
    return out

  @staticmethod
  def row_dict(record) -> dict:
    """One record of rows() (26 numbers, host) -> {field: python scalar} with the members' own types."""
    out = {}
    for name, v in zip(_abi.FIELD_NAMES, record):
      out[name] = float(v) if _abi.FIELD_DTYPES[name] == np.float32 else int(v)
    return out

  def set_grid(self, grid, per_env: bool = False) -> None:
    """`grid`: (21,21,10,9,2) float32 shared by all envs, or (n,21,21,10,9,2) per env."""
    g = grid if isinstance(grid, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(grid, np.float32))
    g = g.to(device=self.device, dtype=torch.float32).contiguous()
    if per_env:
      assert tuple(g.shape) == (self.n,) + GRID_SHAPE, g.shape
      self.grid_env_stride = int(np.prod(GRID_SHAPE))
    else:
      assert tuple(g.shape) == GRID_SHAPE, g.shape
      self.grid_env_stride = 0
    self.grid = g

  def set_vehicle(self, **fields) -> None:
    """The flight vehicle every balloon of this simulator flies: BalloonState's vehicle constants (reference
    env/balloon/balloon.py:156-173: envelope_volume_base, envelope_volume_dv_pressure, envelope_mass,
    envelope_max_superpressure, envelope_cod, payload_mass, nighttime_power_load_w, daytime_power_load_w,
    acs_valve_hole_diameter_m, battery_capacity_wh), mols_lift_gas (:183) and power_safety_layer_enabled (:200), by keyword;
    what is not named keeps the reference's default.  No argument (or all defaults): the kernels with compile-time constants
    (ble_state_f32.vehicle == NULL).  Takes effect with the next call of an entry point, launches prepared by prepare_step_n
    included (they call the entry point, which reads the struct this updates).  A captured HIP graph does NOT see it: the
    entry point derived the vehicle's constants and chose the kernel when the graph was recorded -- capture again after
    set_vehicle (BalloonArena does so itself; VecBalloonEnv.capture_graph is the caller's)."""
    veh = _abi.vehicle_struct(**fields)
    _abi.set_vehicle(self._struct, veh)
    self.vehicle = {} if veh is None else {k: getattr(veh, k) for k in _abi.VEHICLE_DEFAULTS if getattr(veh, k) != _abi.VEHICLE_DEFAULTS[k]}

  # ------------------------------------------------------------------ reset on the device
  @_on_own_device
  def reset_device(self, seed: int, mask: Optional[torch.Tensor] = None, sample: bool = True) -> None:
    """BalloonArena.reset's balloon part for the envs with mask != 0 (all if None), on the GPU:
    draws (if `sample`), Newton cold start, sunrise/sunset search, fresh clocks and FSMs."""
    if mask is not None:
      assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == self.n
    code = self.lib.ble_reset_at_f32(ctypes.byref(self._struct), dev.ptr(mask), int(seed) & (2 ** 64 - 1),
                                     self.episode.data_ptr(), 1 if sample else 0, self.err_flags.data_ptr(), self.env_offset, self.n,
                                     dev.stream_ptr(self.device))
    _lib.check(code, 'ble_reset_at_f32')
    if self._gp is not None:        # a new episode gets a new feature constructor (balloon_arena.py:171-177)
      if mask is None:
        self._obs_reset.fill_(1)
      else:
        torch.maximum(self._obs_reset, mask, out=self._obs_reset)

  # ------------------------------------------------------------------ observation
  @_on_own_device
  def observe(self, noise_uv: Optional[torch.Tensor] = None, append: bool = True,
              out: Optional[torch.Tensor] = None, carry_factor: bool = True,
              forecast_levels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PerciatelliFeatureConstructor.observe + get_features for every env: [n, 1099] float32
    device tensor.  `noise_uv` [n, 2]: measured wind minus forecast at the balloons (None = 0).
    carry_factor (fixed by the first call): keep each env's WindGP Cholesky factor in HBM (61 KB per
    env) and slide it from step to step instead of refactoring the whole window every call.
    forecast_levels [n, 181, 2] float32: the forecast (u, v) at the 181 levels 5 000 .. 14 000 Pa above every balloon as the
    CALLER's WindField gives it (a forecast that is not a grid: `ble_observe_forecast_f32`); None: from the grid."""
    assert self.grid is not None, 'Must call set_grid (reset) before observe.'
    if self._gp is None:
      self._allocate_history(carry_factor)
    if noise_uv is not None:
      assert noise_uv.dtype == torch.float32 and noise_uv.is_contiguous() and tuple(noise_uv.shape) == (self.n, 2)
    if out is None:
      out = torch.empty(self.n, _lib.OBS_DIM, dtype=torch.float32, device=self.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (self.n, _lib.OBS_DIM)
    if forecast_levels is not None:
      assert forecast_levels.dtype == torch.float32 and forecast_levels.is_contiguous() and tuple(forecast_levels.shape) == (self.n, 181, 2)
    code = self.lib.ble_observe_forecast_f32(ctypes.byref(self._struct), self.grid.data_ptr(), self.grid_env_stride, dev.ptr(forecast_levels),
                                             dev.ptr(noise_uv), self._obs_reset.data_ptr(), ctypes.byref(self._gp_struct),
                                             1 if append else 0, out.data_ptr(), self.err_flags.data_ptr(), self.n,
                                             dev.stream_ptr(self.device))
    _lib.check(code, 'ble_observe_forecast_f32')
    self._obs_reset.zero_()         # stream-ordered after the kernel
    return out


  def _allocate_history(self, carry_factor: bool) -> None:
    """The WindGP ring of every environment (and, with carry_factor, the HBM-resident factor slab)."""
    with torch.cuda.device(self.device):
      cap = _lib.GP_CAPACITY
      self._gp = dict(xyp=torch.zeros(self.n, cap, 3, dtype=torch.float32, device=self.device),
                      elapsed_s=torch.zeros(self.n, cap, dtype=torch.int32, device=self.device),
                      err_uv=torch.zeros(self.n, cap, 2, dtype=torch.float32, device=self.device),
                      count=torch.zeros(self.n, dtype=torch.int32, device=self.device))
      if carry_factor:
        self._gp['chol'] = torch.zeros(self.n, _lib.GP_CHOL_STRIDE, dtype=torch.float64, device=self.device)
        self._gp['n_chol'] = torch.zeros(self.n, dtype=torch.int32, device=self.device)
      self._obs_reset = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
      self._gp_struct = _abi.BleGpHistoryF32()
      for name, ct in (('xyp', ctypes.c_float), ('elapsed_s', ctypes.c_int32), ('err_uv', ctypes.c_float),
                       ('count', ctypes.c_int32), ('chol', ctypes.c_double), ('n_chol', ctypes.c_int32)):
        if name not in self._gp:
          continue
        setattr(self._gp_struct, name, ctypes.cast(ctypes.c_void_p(self._gp[name].data_ptr()), ctypes.POINTER(ct)))
      self._gp_struct.chol_stride = _lib.GP_CHOL_STRIDE if carry_factor else 0

  # ------------------------------------------------------------------ checkpoint / resume
  def state_dict(self) -> dict:
    """Everything a resumed run needs to continue bit for bit: the balloon state, the per-environment episode counters,
    the wind grid(s), the WindGP history (ring, carried factor, pending resets) and the live-environment counter --
    clones, on the simulator's device.  The derived caches (per-episode constants, noise draws) are not part of it: they
    are keyed by what they were derived from and refill themselves."""
    d = {'n': self.n, 'env_offset': self.env_offset, 'vehicle': dict(self.vehicle), 'noise_primitive_version': _lib.NOISE_PRIMITIVE_VERSION,
         'state': {k: t.clone() for k, t in self.state.items()}, 'episode': self.episode.clone(),
         'active_slots': self.active_slots.clone(), 'err_flags': self.err_flags.clone(),
         'grid': None if self.grid is None else self.grid.clone(), 'grid_env_stride': self.grid_env_stride, 'gp': None}
    if self._gp is not None:
      d['gp'] = {k: t.clone() for k, t in self._gp.items()}
      d['obs_reset'] = self._obs_reset.clone()
    return d

  @_on_own_device
  def load_state_dict(self, d: dict) -> None:
    """Restores a state_dict() of a simulator of the same size IN PLACE: every tensor, the wind grid included, keeps its
    address, so launches prepared by prepare_step_n and captured HIP graphs stay valid.  Only a checkpoint whose grid
    has another layout (shared vs per-environment) replaces the grid tensor; launches prepared before must then be
    prepared again."""
    assert int(d['n']) == self.n, f"checkpoint of {d['n']} environments, simulator of {self.n}"
    made_with = int(d.get('noise_primitive_version', _lib.NOISE_PRIMITIVE_VERSION))
    if made_with != _lib.NOISE_PRIMITIVE_VERSION:      # (include/ble_abi.h::BLE_NOISE_PRIMITIVE_VERSION)
      raise ValueError(f'checkpoint flown with wind-noise primitive version {made_with}, this library evaluates version '
                       f'{_lib.NOISE_PRIMITIVE_VERSION}: its noise seeds would fly another wind')
    offset = int(d.get('env_offset', self.env_offset))
    if offset != self.env_offset:
      # another shard's checkpoint: the harmonic draws cached for (seed, episode) belong to the OLD global indices -- forget them
      # -- and the generators already handed out (prepared launches hold them by reference) are re-keyed in place
      self.env_offset = offset
      if self._noise_cache is not None:
        self._noise_cache.zero_()
      for gen in self._noise_gens:
        gen.env_offset = offset
    self.set_vehicle(**d.get('vehicle', {}))
    for k, t in self.state.items():
      t.copy_(d['state'][k])
    self.episode.copy_(d['episode']); self.active_slots.copy_(d['active_slots']); self.err_flags.copy_(d['err_flags'])
    if d['grid'] is not None:
      # in place whenever the layout matches: launch closures (prepare_step_n) and captured HIP graphs hold the grid's
      # ADDRESS, so a replaced tensor would leave them reading freed memory; and no second transient copy of a
      # per-environment grid set (10 GB at 32 768 environments)
      if (self.grid is not None and self.grid.shape == d['grid'].shape and
          self.grid_env_stride == int(d['grid_env_stride'])):
        self.grid.copy_(d['grid'])
      else:     # another layout: a new tensor -- prepared launches and graphs of the old one must be rebuilt
        self.set_grid(d['grid'].clone(), per_env=int(d['grid_env_stride']) != 0)
    if d['gp'] is None:
      self._gp, self._obs_reset = None, None
    else:
      carried = 'chol' in d['gp']
      if self._gp is None or ('chol' in self._gp) != carried:
        self._allocate_history(carried)
      for k, t in self._gp.items():
        t.copy_(d['gp'][k])
      self._obs_reset.copy_(d['obs_reset'])

  @_on_own_device
  def wind_noise(self, seed: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SimplexWindNoise at every env's current position and time: [n, 2] float32 (m/s), the
    `noise_uv` input of step() / observe().  One noise field per (seed, env, episode)."""
    if out is None:
      out = torch.empty(self.n, 2, dtype=torch.float32, device=self.device)
    if self._noise_cache is None:      # the harmonics' seeds and offsets, drawn once per (seed, episode) like the reference's
      self._noise_cache = torch.zeros(_lib.NOISE_CACHE_ROWS, self.n, dtype=torch.int32, device=self.device)
    s = self.state
    code = self.lib.ble_wind_noise_at_f32(s['x'].data_ptr(), s['y'].data_ptr(), s['pressure'].data_ptr(),
                                          s['time_elapsed_s'].data_ptr(), int(seed) & (2 ** 64 - 1), self.episode.data_ptr(),
                                          0, self._noise_cache.data_ptr(), out.data_ptr(), self.env_offset, self.n,
                                          dev.stream_ptr(self.device))
    _lib.check(code, 'ble_wind_noise_at_f32')
    return out

  def reset_observation_history(self, mask: Optional[torch.Tensor] = None) -> None:
    """Forget the WindGP observations of the selected envs (all if None)."""
    if self._gp is not None:
      if mask is None:
        self._obs_reset.fill_(1)
      else:
        torch.maximum(self._obs_reset, mask, out=self._obs_reset)

  # ------------------------------------------------------------------ stepping
  @_on_own_device
  def step(self, action: torch.Tensor, noise_uv: Optional[torch.Tensor] = None, substeps: int = SUBSTEPS):
    """One agent step for all envs (asynchronous on the current stream).

    Returns (reward, terminal) device tensors (views of internal buffers).
    """
    assert self.grid is not None, 'Must call set_grid (reset) before step.'   # grid_based_wind_field.py:86-87
    assert action.dtype == torch.uint8 and action.is_contiguous() and action.numel() == self.n
    assert action.device == self.device
    if noise_uv is not None:
      assert noise_uv.dtype == torch.float32 and noise_uv.is_contiguous() and tuple(noise_uv.shape) == (self.n, 2)
    code = self.lib.ble_step_f32(ctypes.byref(self._struct), action.data_ptr(), self.grid.data_ptr(),
                                 self.grid_env_stride, dev.ptr(noise_uv), self.reward.data_ptr(),
                                 self.terminal.data_ptr(), self.effective_action.data_ptr(),
                                 self.err_flags.data_ptr(), self.active_slots.data_ptr(), self.n, substeps,
                                 dev.stream_ptr(self.device))
    _lib.check(code, 'ble_step_f32')
    return self.reward, self.terminal

  def _noise_gen(self, noise_seed: Optional[int], prepared: bool = False):
    """The ble_noise_gen of a fused rollout that flies in the ground-truth wind (forecast + SimplexWindNoise evaluated
    inside the kernel), or None for the forecast alone.  Same generator as wind_noise(seed): same (seed, env, episode)."""
    if noise_seed is None:
      return None
    if self._noise_cache is None:
      with torch.cuda.device(self.device):
        self._noise_cache = torch.zeros(_lib.NOISE_CACHE_ROWS, self.n, dtype=torch.int32, device=self.device)
    gen = _abi.BleNoiseGen(int(noise_seed) & (2 ** 64 - 1), self.episode.data_ptr(), self._noise_cache.data_ptr(), self.env_offset)
    if prepared:          # a prepared launch keeps its generator: load_state_dict re-keys it when the shard offset changes
      self._noise_gens.append(gen)
      del self._noise_gens[:-4096]          # (bounded: a long-lived simulator may prepare launches again and again)
    return gen

  @_on_own_device
  def step_n(self, actions: torch.Tensor, rewards: torch.Tensor, terminals: torch.Tensor,
             active_counts: Optional[torch.Tensor] = None, substeps: int = SUBSTEPS, noise_seed: Optional[int] = None) -> None:
    """`actions` [K, n] uint8 -> K agent steps enqueued by one library call.  noise_seed: fly in the ground-truth wind
    (WindField.get_ground_truth: the noise of wind_noise(noise_seed) evaluated in the kernel before every step)."""
    k = actions.shape[0]
    assert actions.dtype == torch.uint8 and actions.is_contiguous() and tuple(actions.shape) == (k, self.n)
    assert rewards.dtype == torch.float32 and tuple(rewards.shape) == (k, self.n) and rewards.is_contiguous()
    assert terminals.dtype == torch.uint8 and tuple(terminals.shape) == (k, self.n) and terminals.is_contiguous()
    if active_counts is not None:
      assert active_counts.dtype == torch.int64 and tuple(active_counts.shape) == (k, COUNT_SLOTS)
      assert active_counts.is_contiguous()
    gen = self._noise_gen(noise_seed)
    code = self.lib.ble_step_n_f32(ctypes.byref(self._struct), actions.data_ptr(), self.grid.data_ptr(),
                                   self.grid_env_stride, None if gen is None else ctypes.byref(gen), rewards.data_ptr(),
                                   terminals.data_ptr(), self.err_flags.data_ptr(), dev.ptr(active_counts), self.n, substeps, k,
                                   dev.stream_ptr(self.device))
    _lib.check(code, 'ble_step_n_f32')

  def prepare_step_n(self, actions: torch.Tensor, rewards: torch.Tensor, terminals: torch.Tensor,
                     active_counts: Optional[torch.Tensor] = None, substeps: int = SUBSTEPS, noise_seed: Optional[int] = None):
    """step_n with everything but the launch done NOW: the checks run once and the arguments are marshalled once;
    the returned callable enqueues the K agent steps on the stream that is current when IT is called (~3 us of host
    time instead of ~10).  For loops that launch the same buffers again and again (rollouts, the benchmark); the
    caller keeps the tensors, the grid and the simulator alive and unchanged in shape."""
    k = actions.shape[0]
    assert actions.dtype == torch.uint8 and actions.is_contiguous() and tuple(actions.shape) == (k, self.n)
    assert rewards.dtype == torch.float32 and tuple(rewards.shape) == (k, self.n) and rewards.is_contiguous()
    assert terminals.dtype == torch.uint8 and tuple(terminals.shape) == (k, self.n) and terminals.is_contiguous()
    if active_counts is not None:
      assert active_counts.dtype == torch.int64 and tuple(active_counts.shape) == (k, COUNT_SLOTS)
      assert active_counts.is_contiguous()
    assert self.grid is not None, 'Must call set_grid (reset) before step.'
    fn, struct = self.lib.ble_step_n_f32, ctypes.byref(self._struct)
    gen = self._noise_gen(noise_seed, prepared=True)            # (kept alive by the closure)
    grid = self.grid                             # the closure reads THIS tensor: load_state_dict restores it in place
    args = (struct, actions.data_ptr(), grid.data_ptr(), self.grid_env_stride, None if gen is None else ctypes.byref(gen),
            rewards.data_ptr(), terminals.data_ptr(), self.err_flags.data_ptr(), dev.ptr(active_counts), self.n, substeps, k)
    device, index = self.device, self.device.index

    def launch():
      if torch.cuda.current_device() != index:
        with torch.cuda.device(device):
          code = fn(*args, dev.stream_ptr(device))
      else:
        code = fn(*args, dev.stream_ptr(device))
      if code != 0:
        _lib.check(code, 'ble_step_n_f32')
    return launch

  @property
  def active_count(self) -> torch.Tensor:
    """Total number of envs stepped so far (sum of the counter slots), a 0-d device tensor."""
    return self.active_slots.sum()

  def check_errors(self) -> None:
    """Synchronises and raises what the reference would have raised (see raise_for_flags)."""
    flags = int(self.err_flags.item())
    if flags:
      self.err_flags.zero_()
      raise_for_flags(flags)
