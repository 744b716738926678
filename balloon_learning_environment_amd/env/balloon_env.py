"""BalloonEnv (env/balloon_env.py:106-300): the gym-style RL surface over the HIP arena."""
import datetime as dt
import math
import time
from typing import Any, Callable, Dict, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from balloon_learning_environment_amd.env import balloon_arena
from balloon_learning_environment_amd.env import features
from balloon_learning_environment_amd.env import grid_based_wind_field
from balloon_learning_environment_amd.env import grid_wind_field_sampler
from balloon_learning_environment_amd.env import simulator_data
from balloon_learning_environment_amd.env.balloon import balloon
from balloon_learning_environment_amd.env.balloon import control
from balloon_learning_environment_amd.utils import units


def perciatelli_reward_function(simulator_state: simulator_data.SimulatorState, *, station_keeping_radius_km: float = 50.0,
                                reward_dropoff: float = 0.4, reward_halflife: float = 100.0) -> float:
  """Host restatement of env/balloon_env.py:44-102 for callers that pass their own
  reward_function / non-default parameters; the default path uses the kernel's reward."""
  b = simulator_state.balloon_state
  radius = units.Distance(km=station_keeping_radius_km)
  distance = units.relative_distance(b.x, b.y)
  if distance <= radius:
    reward = 1.0
  else:
    reward = reward_dropoff * math.exp(-0.69314718056 / reward_halflife * (distance - radius).kilometers)
  if b.last_command == control.AltitudeControlCommand.DOWN and not b.excess_energy:
    scale = min(max((b.acs_power.watts - 100.0) / (300.0 - 100.0), 0.0), 1.0)
    reward *= 0.95 - 0.3 * scale
  return reward


def generative_wind_field_factory(device='cuda:0'):
  """GridBasedWindField over the VAE sampler, as in the reference (env/generative_wind_field.py:35-37).
  The decoder runs on the device with synthetic weights (the trained blob is not in the checkout)."""
  from balloon_learning_environment_amd.env import generative_wind_field
  return grid_based_wind_field.GridBasedWindField(generative_wind_field.GenerativeWindFieldSampler(device=device), device)


def gaussian_wind_field_factory(device='cuda:0'):
  """White-noise grid (tests, worst case for the interpolation)."""
  return grid_based_wind_field.GridBasedWindField(grid_wind_field_sampler.GaussianFieldSampler(), device)


class BalloonEnv:
  """Old-gym (0.21) 4-tuple API like the reference."""
  metadata: Dict[str, Any] = {'render.modes': []}

  def __init__(self, *, station_keeping_radius_km: float = 50.0,
               arena: Optional[balloon_arena.BalloonArenaInterface] = None,
               reward_function: Callable[[simulator_data.SimulatorState], float] = perciatelli_reward_function,
               feature_constructor_factory: Callable = features.perciatelli_feature_constructor,
               wind_field_factory: Callable = generative_wind_field_factory, seed: Optional[int] = None,
               renderer=None):
    self.radius = units.Distance(km=station_keeping_radius_km)
    self._reward_fn = reward_function
    self._use_kernel_reward = (reward_function is perciatelli_reward_function and station_keeping_radius_km == 50.0)
    self._global_iteration = 0
    self.arena = arena if arena is not None else balloon_arena.BalloonArena(feature_constructor_factory, wind_field_factory())
    self._renderer = renderer
    if renderer is not None:
      self.metadata = {'render.modes': renderer.render_modes}       # env/balloon_env.py:150-152
    self.reset(seed=seed if seed is not None else int(time.time() * 1e6) % (2 ** 31))

  def step(self, action: int) -> Tuple[np.ndarray, float, bool, Mapping[str, Any]]:
    command = control.AltitudeControlCommand(action)
    observation = self.arena.step(command)
    assert isinstance(observation, np.ndarray)
    kernel_reward = getattr(self.arena, 'last_reward', None)
    row = getattr(self.arena, '_row', None)
    if row is not None and self._renderer is None and self._use_kernel_reward and kernel_reward is not None:
      # the step brought the balloon's row back with the observation (BalloonArena._step_fast): status and clock straight from it;
      # the BalloonState object is built when somebody asks for it (get_simulator_state)
      reward, status = kernel_reward, int(row['status'])
      info = {'out_of_power': status == balloon.BalloonStatus.OUT_OF_POWER.value, 'envelope_burst': status == balloon.BalloonStatus.BURST.value,
              'zeropressure': status == balloon.BalloonStatus.ZEROPRESSURE.value, 'time_elapsed': dt.timedelta(seconds=int(row['time_elapsed_s']))}
    else:
      simulator_state = self.arena.get_simulator_state()
      if self._renderer is not None:
        self._renderer.step(simulator_state)
      reward = kernel_reward if (self._use_kernel_reward and kernel_reward is not None) else self._reward_fn(simulator_state)
      info = self._get_info(simulator_state.balloon_state)
    is_terminal = info['out_of_power'] or info['envelope_burst'] or info['zeropressure']
    self._global_iteration += 1
    return observation, reward, is_terminal, info

  def reset(self, *, seed: Optional[int] = None, return_info: bool = False):
    if seed is not None:
      self.seed(seed)
    self._rng = np.random.Generator(np.random.Philox(self._rng.integers(0, 2 ** 31)))
    observation = self.arena.reset(int(self._rng.integers(0, 2 ** 31)))
    if self._renderer is not None:                                  # env/balloon_env.py:216-218
      self._renderer.reset()
      self._renderer.step(self.arena.get_simulator_state())
    if return_info:
      return observation, self._get_info(self.get_simulator_state().balloon_state)
    return observation

  def render(self, mode: str = 'human'):
    return None if self._renderer is None else self._renderer.render(mode)

  def close(self) -> None:
    pass

  def seed(self, seed: int) -> None:
    self._rng = np.random.Generator(np.random.Philox(int(seed)))

  @property
  def unwrapped(self): return self

  @property
  def action_space(self): return features.Discrete(3)

  @property
  def observation_space(self): return self.arena.feature_constructor.observation_space

  @property
  def reward_range(self): return (0.0, 1.0)

  def get_simulator_state(self) -> simulator_data.SimulatorState:
    return self.arena.get_simulator_state()

  def _get_info(self, balloon_state: balloon.BalloonState) -> Dict[str, Any]:
    return {'out_of_power': balloon_state.status == balloon.BalloonStatus.OUT_OF_POWER,
            'envelope_burst': balloon_state.status == balloon.BalloonStatus.BURST,
            'zeropressure': balloon_state.status == balloon.BalloonStatus.ZEROPRESSURE,
            'time_elapsed': balloon_state.time_elapsed}

  def __str__(self): return 'BalloonEnv'
  def __enter__(self): return self

  def __exit__(self, *args):
    self.close()
    return False


class VecBalloonEnv:
  """N BalloonEnvs advanced together on one GPU (no reference counterpart: the reference steps one
  environment per Python call).  Same semantics per environment as BalloonEnv -- Perciatelli
  observation, perciatelli_reward_function, terminal on out-of-power / burst / zero-pressure --
  with device tensors in and out and optional auto-reset of terminated environments.

    env = VecBalloonEnv(65536, seed=0)
    obs = env.reset()                                  # [N, 1099] float32 (device)
    obs, reward, terminal = env.step(actions_u8)       # device tensors
  """

  def __init__(self, num_envs: int, *, seed: int = 0, wind_field=None, wind_noise: bool = True,
               auto_reset: bool = True, per_env_fields: bool = False, field_refresh_every: int = 32, device='cuda:0'):
    """wind_noise=True (default, as the reference): ground truth = forecast + SimplexWindNoise, so the WindGP
    has an error signal to model; False is the opt-out (forecast == truth, every GP error exactly 0).
    Wind fields: see VecBalloonArena (default: generative sampler, one shared field per reset();
    per_env_fields=True: one decoded field per environment and per episode)."""
    self.arena = balloon_arena.VecBalloonArena(num_envs, wind_field, seed=seed, device=device, per_env_fields=per_env_fields)
    self.num_envs, self.device = self.arena.num_envs, self.arena.device
    self._seed, self._wind_noise, self._auto_reset = int(seed), bool(wind_noise), bool(auto_reset)
    self._field_refresh_every, self._steps_since_refresh = int(field_refresh_every), 0
    self._noise = None
    self._graph = None
    self._reseed = True               # the first reset() replays the constructor's seed

  def _noise_now(self):
    if not self._wind_noise:
      return None
    self._noise = self.arena.sim.wind_noise(self.arena._seed, out=self._noise)
    return self._noise

  def seed(self, seed: int) -> None:
    self._seed = int(seed)
    self._reseed = True

  def reset(self, seed: Optional[int] = None):
    """reset(seed) / seed(s); reset(): reproducible new episodes and wind field(s).  reset() without a
    seed: the next episodes of the current seed (new initial conditions, new wind field(s))."""
    if seed is not None:
      self.seed(seed)
    if getattr(self, '_reseed', False):
      self.arena.reset(self._seed); self._reseed = False
    else:
      self.arena.reset()
    self.check_errors()
    self._graph = None
    self._steps_since_refresh = 0
    return self.arena.observe(self._noise_now())

  def state_dict(self) -> dict:
    """Checkpoint of the whole batch (balloons, episode counters, wind field(s), WindGP histories, seeds): a run restored
    with load_state_dict() continues bit for bit -- same observations, rewards, terminals, auto-resets."""
    return {'arena': self.arena.state_dict(), 'seed': self._seed, 'reseed': bool(getattr(self, '_reseed', False)),
            'steps_since_refresh': self._steps_since_refresh,
            'noise': None if self._noise is None else self._noise.clone()}

  def load_state_dict(self, d: dict) -> None:
    self.arena.load_state_dict(d['arena'])
    self._seed, self._reseed, self._steps_since_refresh = int(d['seed']), bool(d['reseed']), int(d['steps_since_refresh'])
    self._noise = None if d['noise'] is None else d['noise'].clone()
    self._graph = None                       # a captured graph holds the old buffers: capture again after a resume

  def check_errors(self) -> None:
    """Synchronises and raises what the reference would have raised inside step / observe since the last
    call (non-finite state, pressure out of range, WindGP window overflow, failed pressure-range search...)."""
    self.arena.sim.check_errors()

  def _step_eager(self, actions, obs_out=None):
    noise = None
    if self._wind_noise:
      noise = self._noise if self._noise is not None else self._noise_now()     # step() before reset()
    reward, terminal = self.arena.step(actions, noise)
    if self._auto_reset:
      self._terminal_buf.copy_(terminal)
      self.arena.reset_lanes(self._terminal_buf)
    return self.arena.observe(self._noise_now(), out=obs_out), reward, terminal

  def step(self, actions):
    """actions: uint8 device tensor [N] in {0, 1, 2}.  Returns (obs [N, 1099], reward [N], terminal [N] u8);
    `terminal` refers to the transition just made; with auto_reset the returned observation of a
    terminated environment is the first one of its next episode.  With `capture_graph()` the three
    tensors are static buffers that the next step overwrites.  Error conditions are latched on the device:
    call check_errors() (reset() does) to have them raised."""
    if not hasattr(self, '_terminal_buf'):
      self._terminal_buf = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
    if self._graph is not None:
      self._g_actions.copy_(actions)
      self._graph.replay()
      out = self._g_obs, self._g_reward, self._g_terminal
    else:
      obs, reward, terminal = self._step_eager(actions)
      out = obs, reward.clone(), terminal.clone()
    if self.arena.per_env_fields and self._auto_reset:
      self._steps_since_refresh += 1
      if self._steps_since_refresh >= self._field_refresh_every:
        self._steps_since_refresh = 0
        self.arena.refresh_fields()
    return out

  def capture_graph(self):
    """Records one step (wind noise, transition, masked reset, observation: four kernels plus a few
    fills) into a HIP graph and replays it from then on -- for small batches the loop is launch-bound.
    Call after reset() and at least one step() (lazy allocations must have happened)."""
    n, dev_ = self.num_envs, self.device
    self._g_actions = torch.ones(n, dtype=torch.uint8, device=dev_)
    self._g_obs = torch.empty(n, 1099, dtype=torch.float32, device=dev_)
    if not hasattr(self, '_terminal_buf'):
      self._terminal_buf = torch.zeros(n, dtype=torch.uint8, device=dev_)
    side = torch.cuda.Stream(device=dev_)
    side.wait_stream(torch.cuda.current_stream(dev_))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
      with torch.cuda.graph(graph, stream=side):
        _, reward, terminal = self._step_eager(self._g_actions, obs_out=self._g_obs)
        self._g_reward, self._g_terminal = reward.clone(), terminal.clone()
    torch.cuda.current_stream(dev_).wait_stream(side)
    self._graph = graph

  @property
  def observation_space(self):
    return features.Box(np.concatenate([[0, 0, 0, -1, -1, -1, -1], np.zeros(8), [1.0], np.zeros(1083)]).astype(np.float32),
                        np.concatenate([np.ones(15), [np.inf], np.ones(1083)]).astype(np.float32))

  @property
  def action_space(self):
    return features.Discrete(3)
