"""BalloonArena (env/balloon_arena.py:42-275) on the HIP transition.

`VecBalloonArena` is the native object: N environments, device tensors, one kernel launch
per agent step.  `BalloonArena` is the reference-shaped single-environment facade over a
VecBalloonArena of size 1, so that BalloonEnv(arena=...) and code written against
BalloonArenaInterface keep working.
"""
import abc
import time
from typing import Callable, Optional, Union

import numpy as np
import torch

from balloon_learning_environment_amd import _abi
from balloon_learning_environment_amd import vec_state
from balloon_learning_environment_amd.env import features
from balloon_learning_environment_amd.env import grid_based_wind_field
from balloon_learning_environment_amd.env import grid_wind_field_sampler
from balloon_learning_environment_amd.env import simulator_data
from balloon_learning_environment_amd.env import wind_field as wind_field_lib
from balloon_learning_environment_amd.env.balloon import balloon
from balloon_learning_environment_amd.env.balloon import control
from balloon_learning_environment_amd.utils import constants
from balloon_learning_environment_amd.utils import units


def _mix_seed(seed: int, count: int) -> int:
  """Seed of the `count`-th un-seeded reset after reset(seed) (splitmix64 finaliser; count 0 -> seed)."""
  if count == 0:
    return int(seed)
  z = (int(seed) + 0x9E3779B97F4A7C15 * count) & (2 ** 64 - 1)
  z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
  z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
  return int((z ^ (z >> 31)) % (2 ** 31))


class VecBalloonArena:
  """N independent arenas advanced together (one lane per arena).

  Wind fields.  Default: the generative sampler (env/generative_wind_field.py; synthetic decoder
  weights) like the reference's BalloonEnv.  `per_env_fields=False`: all N environments fly in ONE
  field, resampled by every reset() -- reset(seed) uses `seed`, each later reset() without a seed
  derives a new one from (seed, reset count); environments auto-reset in between (reset_terminated)
  keep flying in that field.  `per_env_fields=True`: every environment has its own decoded grid
  (N x 317.5 KB of HBM) and gets a NEW one for each new episode: reset() decodes all of them,
  reset_terminated() marks the lanes and refresh_fields() re-decodes the marked ones (one host
  synchronisation, so VecBalloonEnv calls it every `field_refresh_every` steps; until then a re-started
  lane flies in its previous field)."""

  def __init__(self, num_envs: int, wind_field_instance: Optional[grid_based_wind_field.GridBasedWindField] = None,
               seed: Optional[int] = None, device='cuda:0', per_env_fields: bool = False, env_offset: int = 0):
    """env_offset: this arena's environment 0 in the global batch of a sharded job (vec_state.VecSimulator): with one seed for
    the job the shards reset and fly in the noise exactly as the unsharded batch would."""
    self.num_envs = int(num_envs)
    self.sim = vec_state.VecSimulator(self.num_envs, device, env_offset=env_offset)
    self.device = self.sim.device
    if wind_field_instance is None:
      from balloon_learning_environment_amd.env import generative_wind_field
      wind_field_instance = grid_based_wind_field.GridBasedWindField(
          generative_wind_field.GenerativeWindFieldSampler(device=self.device), self.device)
    self.wind_field = wind_field_instance
    self.per_env_fields = bool(per_env_fields)
    self._grids = None
    self._stale = None
    self._field_epoch = 0
    self._step_duration = constants.AGENT_TIME_STEP
    self._seed0, self._reset_count = None, 0
    self.reset(seed)

  def reset(self, seed: Optional[int] = None, on_device: bool = True) -> None:
    """New episodes for every env, by ble_reset_f32 (Philox streams on the GPU).  There is no host path: on_device=False
    (rounds 1-4: a NumPy sampler, now test tooling in tests/reset_host.py) raises; a host-made state goes in through
    `sim.set_state(...)` followed by `sim.reset_observation_history()`.
    reset(seed) is reproducible: the same seed gives the same episodes and wind field(s), whatever
    happened before (the reference's env.seed(s); env.reset() contract, eval/eval_lib.py)."""
    if seed is not None:
      self._seed0, self._reset_count = int(np.asarray(seed).ravel()[-1]), 0
    elif self._seed0 is None:
      self._seed0, self._reset_count = int(time.time() * 1e6) % (2 ** 31), 0
    else:
      self._reset_count += 1
    seed = _mix_seed(self._seed0, self._reset_count)
    self._seed = seed
    self.sim.episode.zero_()            # Philox streams are keyed by (seed, env, episode): restart the count
    if not on_device:
      raise NotImplementedError('VecBalloonArena.reset: the episode reset runs on the device only (ble_reset_f32)')
    self.sim.reset_device(seed)
    self.wind_field.reset(np.array([seed], np.uint32), None)
    self._field_epoch = 0
    if self.per_env_fields:
      self._decode_fields(None)
    else:
      self.sim.set_grid(self.wind_field.grid)

  def _decode_fields(self, lanes: Optional[torch.Tensor]) -> None:
    sampler = getattr(self.wind_field, '_wind_field_sampler', None)
    assert hasattr(sampler, 'decode'), 'per_env_fields needs a sampler with a batched device decode()'
    if self._grids is None:
      self._grids = torch.empty((self.num_envs,) + tuple(vec_state.GRID_SHAPE), dtype=torch.float32, device=self.device)
      self._stale = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
    # one latent per (seed, GLOBAL environment index, that environment's episode count): a shard (env_offset) decodes exactly
    # the fields the unsharded batch decodes for its environments, and a masked refresh what a full one would
    index = torch.arange(self.num_envs, dtype=torch.int64, device=self.device) if lanes is None else lanes.to(torch.int64)
    latents = sampler.sample_latents_keyed(index + self.sim.env_offset, self.sim.episode[index], self._seed)
    self._field_epoch += 1
    if lanes is None:
      sampler.decode(latents, self._grids)
    else:
      self._grids[lanes] = sampler.decode(latents)
    self.sim.set_grid(self._grids, per_env=True)
    self._stale.zero_()

  def refresh_fields(self) -> int:
    """per_env_fields: new wind fields for the lanes re-started since the last call.  Returns their number."""
    if not self.per_env_fields or self._stale is None:
      return 0
    lanes = torch.nonzero(self._stale, as_tuple=False).flatten()      # host synchronisation
    if lanes.numel():
      self._decode_fields(lanes)
    return int(lanes.numel())

  def reset_terminated(self) -> int:
    """Auto-reset: starts a new episode in every env whose status != OK (shared field: the same
    wind field; per_env_fields: a new one at the next refresh_fields()).  Returns the number reset."""
    mask = (self.sim.state['status'] != 0).to(torch.uint8)
    self.reset_lanes(mask)
    return int(mask.sum().item())

  def reset_lanes(self, mask: torch.Tensor) -> None:
    """Masked auto-reset without a host round trip (mask: uint8 [N], != 0 -> new episode)."""
    self.sim.reset_device(self._seed, mask=mask)
    if self.per_env_fields and self._stale is not None:
      torch.maximum(self._stale, mask, out=self._stale)

  # ---- checkpoint / resume ------------------------------------------------------------
  def state_dict(self) -> dict:
    """The simulator's state_dict plus what this arena adds: the seed bookkeeping of reset(), the wind field (host copy)
    and its noise seed, the per-environment grids' refresh state.  torch.save-able."""
    wf = self.wind_field
    noise = getattr(wf, 'noise_model', None)
    return {'sim': self.sim.state_dict(), 'seed0': self._seed0, 'reset_count': self._reset_count, 'seed': getattr(self, '_seed', None),
            'field_epoch': self._field_epoch, 'per_env_fields': self.per_env_fields,
            # (torch tensors and plain Python scalars only: loadable with torch.load's default weights_only=True)
            'field': None if getattr(wf, 'field', None) is None else torch.from_numpy(np.array(wf.field, copy=True)),
            'noise_seed': None if noise is None else noise._seed,
            'stale': None if self._stale is None else self._stale.clone()}

  def load_state_dict(self, d: dict) -> None:
    assert bool(d['per_env_fields']) == self.per_env_fields, 'checkpoint and arena differ in per_env_fields'
    self._seed0, self._reset_count, self._seed, self._field_epoch = d['seed0'], d['reset_count'], d['seed'], d['field_epoch']
    if d['field'] is not None:
      self.wind_field.set_field(d['field'].cpu().numpy())
    noise = getattr(self.wind_field, 'noise_model', None)
    if noise is not None and d['noise_seed'] is not None:
      noise._seed = d['noise_seed']
    self.sim.load_state_dict(d['sim'])
    if self.per_env_fields:
      self._grids = self.sim.grid                          # the restored per-environment grids
      self._stale = d['stale'].clone() if d['stale'] is not None else torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
    else:
      self.sim.set_grid(self.wind_field.grid)

  def step(self, actions: torch.Tensor, noise_uv: Optional[torch.Tensor] = None):
    """actions: uint8 device tensor [N] -> (reward [N] f32, terminal [N] u8) device tensors."""
    return self.sim.step(actions, noise_uv)

  def observe(self, noise_uv: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Perciatelli observation of every env: [N, 1099] float32 device tensor (features.py:301-330 on
    the device, WindGP history kept per env).  `noise_uv`: measured minus forecast wind at the
    balloons' current positions."""
    return self.sim.observe(noise_uv, out=out)

  # ---- per-env views -----------------------------------------------------------------
  def row(self, i: int) -> dict:
    """State of env i as Python scalars: ONE kernel (`ble_state_rows_f64`) and one device->host copy."""
    return self.sim.row_dict(self.sim.rows(i, 1)[0].cpu().tolist())

  def get_balloon_state(self, i: int = 0) -> balloon.BalloonState:
    return balloon.state_from_row(self.row(i), self.sim.vehicle)

  def set_balloon_state(self, new_state: balloon.BalloonState, i: int = 0) -> None:
    # the state's flight-vehicle constants (balloon.py:156-173,183,200): one vehicle per batch (ble_state_f32.vehicle)
    veh = _abi.vehicle_struct(**balloon.vehicle_of(new_state))
    overrides = {} if veh is None else {k: getattr(veh, k) for k in _abi.VEHICLE_DEFAULTS if getattr(veh, k) != _abi.VEHICLE_DEFAULTS[k]}
    if overrides != self.sim.vehicle:
      if self.num_envs != 1:
        raise ValueError('all environments of a batch fly one vehicle: VecSimulator.set_vehicle(...) sets it for the batch')
      self.sim.set_vehicle(**overrides)
    alpha = float(self.sim.state['alpha'][i].item())
    row = balloon.row_from_state(new_state, alpha)
    for name, value in row.items():
      self.sim.state[name][i] = torch.tensor(value).to(self.sim.state[name].dtype)

  def get_atmosphere(self, i: int = 0) -> simulator_data.Atmosphere:
    return simulator_data.Atmosphere(float(self.sim.state['alpha'][i].item()))


class BalloonArenaInterface(abc.ABC):
  @abc.abstractmethod
  def reset(self, seed: Optional[int] = None) -> np.ndarray: ...
  @abc.abstractmethod
  def step(self, action: control.AltitudeControlCommand) -> np.ndarray: ...
  @abc.abstractmethod
  def get_simulator_state(self) -> simulator_data.SimulatorState: ...
  @abc.abstractmethod
  def set_simulator_state(self, new_state: simulator_data.SimulatorState) -> None: ...
  @abc.abstractmethod
  def get_balloon_state(self) -> balloon.BalloonState: ...
  @abc.abstractmethod
  def set_balloon_state(self, new_state: balloon.BalloonState) -> None: ...
  @abc.abstractmethod
  def get_measurements(self) -> simulator_data.SimulatorObservation: ...


class BalloonArena(BalloonArenaInterface):
  """One balloon in one wind field (reference constructor signature)."""

  def __init__(self, feature_constructor_factory: Callable = features.perciatelli_feature_constructor,
               wind_field_instance: Optional[grid_based_wind_field.GridBasedWindField] = None,
               seed: Optional[int] = None, device='cuda:0'):
    self._feature_constructor_factory = feature_constructor_factory
    self._vec = VecBalloonArena.__new__(VecBalloonArena)
    self._vec.num_envs = 1
    self._vec.sim = vec_state.VecSimulator(1, device)
    self._vec.device = self._vec.sim.device
    self._vec.wind_field = wind_field_instance or grid_based_wind_field.GridBasedWindField(
        grid_wind_field_sampler.GaussianFieldSampler(), device)
    self._vec.per_env_fields, self._vec._grids, self._vec._stale = False, None, None
    self._vec._step_duration = constants.AGENT_TIME_STEP
    self._wind_field = self._vec.wind_field
    self.last_reward = None
    self._row = None          # host copy of the balloon's row, valid until the next device-side change
    self._fast, self._noise_valid = None, False      # the one-transfer step (see _step_fast)
    self.reset(seed)

  def _bind_wind_field(self) -> None:
    """A grid-based field is interpolated inside the step kernel.  Any other WindField is
    looked up on the host (`get_ground_truth`, env/balloon_arena.py:270-275) and handed to the
    kernel through its additive wind input over an all-zero grid."""
    self._grid_based = getattr(self._wind_field, 'grid', None) is not None
    if self._grid_based:
      self._vec.sim.set_grid(self._wind_field.grid)
    else:
      self._vec.sim.set_grid(torch.zeros(grid_wind_field_sampler.FieldShape().grid_shape(), dtype=torch.float32, device=self._vec.device))

  def _host_wind(self) -> Optional[torch.Tensor]:
    b = self.get_balloon_state()
    if self._grid_based:
      w = self._wind_field.get_wind_noise(b.x, b.y, b.pressure, b.time_elapsed)
      if w.u.mps == 0.0 and w.v.mps == 0.0:
        return None
    else:
      w = self._wind_field.get_ground_truth(b.x, b.y, b.pressure, b.time_elapsed)
    return torch.tensor([[w.u.mps, w.v.mps]], dtype=torch.float32, device=self._vec.device)

  def reset(self, seed: Union[int, np.ndarray, None] = None) -> np.ndarray:
    self._vec.wind_field = self._wind_field
    seed_ = int(time.time() * 1e6) % (2 ** 31) if seed is None else int(np.asarray(seed).ravel()[-1])
    self._vec._seed = seed_
    self._row = None
    self._fast, self._noise_valid = None, False      # a new feature constructor (new WindGP buffers): buffers and graph are rebuilt
    self._vec.sim.episode.zero_()         # reset(seed) reproduces the same episode whatever happened before
    self._vec.sim.reset_device(seed_)
    self._wind_field.reset(np.array([seed_], np.uint32), self.get_balloon_state().date_time)
    self._bind_wind_field()
    self.feature_constructor = self._feature_constructor_factory(self._wind_field, self._vec.get_atmosphere())
    # the device constructor reads this arena's state in place -- unless a subclass edits what the balloon "measures"
    # (get_measurements overridden, e.g. sensor noise): then the observation object is the only truth
    if hasattr(self.feature_constructor, 'bind_state') and type(self).get_measurements is BalloonArena.get_measurements:
      self.feature_constructor.bind_state(self._vec.sim)
    self._observe()
    return self.feature_constructor.get_features()

  def _observe(self) -> None:
    fc = self.feature_constructor
    (fc.observe_bound if getattr(fc, '_bound', False) else fc.observe)(self.get_measurements())

  # ---- the one-transfer step ---------------------------------------------------------------------------------------------
  # BalloonArena.step as the reference's loop drives it (`observation, reward, done, info = env.step(action)`,
  # eval/eval_lib.py:158-171) is host-synchronous: what it costs is round trips, not arithmetic.  With a grid-based wind field
  # and the device feature constructor everything a step needs is already on the device, so the step is ONE HIP graph --
  # action in, transition (ble_step_f32) in the ground-truth wind, the wind noise at the new position (ble_wind_noise_f32: the
  # measured-minus-forecast term of this observation AND the next transition's ground-truth term), the observation
  # (ble_observe_f32), the balloon's row (ble_state_rows_f64) -- and ONE pinned device-to-host copy of 4.6 KB carrying the
  # observation, the row, the reward and both error words.  Anything the path cannot express (a host-only wind field, a
  # subclass that edits the measurements, another feature constructor) takes the general path below.
  def _fast_path_ok(self) -> bool:
    fc = getattr(self, 'feature_constructor', None)
    model = getattr(self._wind_field, 'noise_model', None)
    return (getattr(self, '_grid_based', False) and type(self._wind_field) is grid_based_wind_field.GridBasedWindField and
            type(fc) is features.PerciatelliFeatureConstructor and getattr(fc, '_bound', False) and
            type(self).get_measurements is BalloonArena.get_measurements and type(self)._host_wind is BalloonArena._host_wind and
            (model is None or (type(model).__name__ == 'SimplexWindNoise' and model._seed is not None)))

  def _launch_noise(self, out: torch.Tensor) -> None:
    """SimplexWindNoise.get_wind_noise at the balloon's CURRENT position and time, device to device: the launch
    env/simplex_wind_noise.py::SimplexWindNoise.get_wind_noise makes, on the state where it lives."""
    from balloon_learning_environment_amd import _lib, device as dev
    s, sim = self._vec.sim.state, self._vec.sim
    _lib.check(sim.lib.ble_wind_noise_f32(s['x'].data_ptr(), s['y'].data_ptr(), s['pressure'].data_ptr(), s['time_elapsed_s'].data_ptr(),
                                          int(self._wind_field.noise_model._seed), 0, 0, 0, out.data_ptr(), 1, dev.stream_ptr(sim.device)),
               'ble_wind_noise_f32')

  def _fast_setup(self) -> dict:
    d = self._vec.device
    stage = torch.zeros(240 + 4 * 1099, dtype=torch.uint8, device=d)          # [26 row doubles | reward, step flags, observe flags, pad | 1099 floats]
    host = torch.zeros(240 + 4 * 1099, dtype=torch.uint8).pin_memory()
    host_np = host.numpy()
    f = dict(stage=stage, host=host, row=stage[:208].view(torch.float64).view(1, 26), extra=stage[208:240].view(torch.float64),
             obs=stage[240:].view(torch.float32).view(1, 1099), host_head=host_np[:240].view(np.float64), host_obs=host_np[240:].view(np.float32),
             action=torch.ones(1, dtype=torch.uint8, device=d), action_host=torch.ones(1, dtype=torch.uint8).pin_memory(),
             noise=(torch.zeros(1, 2, dtype=torch.float32, device=d) if getattr(self._wind_field, 'noise_model', None) is not None else None),
             graph=None, eager_steps=0)
    f['action_np'] = f['action_host'].numpy()
    return f

  def _fast_body(self, f: dict) -> None:
    sim, fsim = self._vec.sim, self.feature_constructor._sim
    f['action'].copy_(f['action_host'], non_blocking=True)
    sim.step(f['action'], f['noise'])                 # the ground-truth wind: forecast (in-kernel) + the noise evaluated after the previous step
    if f['noise'] is not None:
      self._launch_noise(f['noise'])                  # at the NEW position: this observation's error term and the next step's noise
    fsim.observe(f['noise'], out=f['obs'])
    sim.rows(0, 1, out=f['row'])
    f['extra'][0:1].copy_(sim.reward); f['extra'][1:2].copy_(sim.err_flags); f['extra'][2:3].copy_(fsim.err_flags)
    f['host'].copy_(f['stage'], non_blocking=True)

  def _step_fast(self, action: control.AltitudeControlCommand) -> np.ndarray:
    sim = self._vec.sim
    if self._fast is None:
      self._fast = self._fast_setup()
    f = self._fast
    stream = torch.cuda.current_stream(sim.device)
    if not self._noise_valid and f['noise'] is not None:     # after reset() / set_balloon_state(): the noise at where the balloon is now
      self._launch_noise(f['noise'])
    self._noise_valid = True
    if f['graph'] is None and f['eager_steps'] >= 2:
      # buffers exist and every lazy allocation has happened: record the step once (recording does not execute it), replay from now on
      stream.synchronize()
      side = torch.cuda.Stream(device=sim.device)
      side.wait_stream(stream)
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
          self._fast_body(f)
      stream.wait_stream(side)
      f['graph'] = graph
    f['action_np'][0] = int(action)
    if f['graph'] is not None:
      f['graph'].replay()
    else:
      self._fast_body(f)
      f['eager_steps'] += 1
    stream.synchronize()
    return self._fast_finish(f['host_head'], f['host_obs'])

  def _fast_finish(self, head, obs) -> np.ndarray:
    sim, fsim = self._vec.sim, self.feature_constructor._sim
    step_flags, obs_flags = int(head[27]), int(head[28])
    if step_flags or obs_flags:        # what the reference would have raised inside the transition / the feature constructor
      sim.err_flags.zero_(); fsim.err_flags.zero_()
      vec_state.raise_for_flags(step_flags | obs_flags)
    self._row = sim.row_dict(head[:26])
    self.last_reward = float(np.float32(head[26]))
    self.feature_constructor._features = obs.copy()        # (the pinned buffer is overwritten by the next step)
    return self.feature_constructor._features.copy()

  def step(self, action: control.AltitudeControlCommand) -> np.ndarray:
    # balloon.py:288-290: stepping a terminal balloon is an error in the single-env API
    if self._row is None:
      self._row = self._vec.row(0)
    status = balloon.BalloonStatus(int(self._row['status']))
    assert status == balloon.BalloonStatus.OK, (
        f'Stepping balloon after a terminal event occured. ({status.name})')
    if self._fast_path_ok():
      return self._step_fast(control.AltitudeControlCommand(action))
    self._noise_valid = False
    a = torch.tensor([int(action)], dtype=torch.uint8, device=self._vec.device)
    reward, _ = self._vec.step(a, self._host_wind())
    self._row = None
    self._vec.sim.check_errors()
    self.last_reward = float(reward[0].item())
    self._observe()
    return self.feature_constructor.get_features()

  def get_simulator_state(self) -> simulator_data.SimulatorState:
    return simulator_data.SimulatorState(self.get_balloon_state(), self._wind_field, self._vec.get_atmosphere())

  def set_simulator_state(self, new_state: simulator_data.SimulatorState) -> None:
    self._row = None
    self._fast, self._noise_valid = None, False       # (another wind field may come with it)
    self._vec.sim.state['alpha'][0] = float(new_state.atmosphere.alpha)
    self.set_balloon_state(new_state.balloon_state)
    self._wind_field = new_state.wind_field
    self._vec.wind_field = new_state.wind_field
    self._bind_wind_field()

  def get_balloon_state(self) -> balloon.BalloonState:
    """A fresh BalloonState built from ONE read-back of the device row per transition (the reference hands out
    its live state object; callers that mutate the result must pass it to set_balloon_state)."""
    if self._row is None:
      self._row = self._vec.row(0)
    return balloon.state_from_row(self._row, self._vec.sim.vehicle)

  def set_balloon_state(self, new_state: balloon.BalloonState) -> None:
    self._row = None
    self._noise_valid = False                          # the balloon is somewhere else now
    before = dict(self._vec.sim.vehicle)
    self._vec.set_balloon_state(new_state, 0)
    if self._vec.sim.vehicle != before:
      self._fast = None                                # (a captured graph holds the KERNEL it captured: default vs run-time vehicle)

  def get_measurements(self) -> simulator_data.SimulatorObservation:
    b = self.get_balloon_state()
    return simulator_data.SimulatorObservation(
        balloon_observation=b, wind_at_balloon=self._wind_field.get_ground_truth(b.x, b.y, b.pressure, b.time_elapsed))
