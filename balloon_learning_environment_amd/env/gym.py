"""Gym registration (env/gym.py:20-40 of the reference): `BalloonLearningEnvironment-v0`.

gym is optional here (it is not installed in the build image): `register_env()` raises ImportError
without it, and nothing else in the package needs it -- BalloonEnv is duck-typed to the gym 0.21 API
(step / reset / seed / render / close, action_space, observation_space, reward_range, metadata).
"""
import contextlib

ENV_ID = 'BalloonLearningEnvironment-v0'
ENTRY_POINT = 'balloon_learning_environment_amd.env.balloon_env:BalloonEnv'


def register_env() -> None:
  """Registers this package's BalloonEnv under the reference's environment id."""
  from gym.envs import registration  # inline like the reference: avoids a circular import inside gym's plugin loader

  specs = getattr(registration.registry, 'env_specs', registration.registry)
  if ENV_ID in specs:
    return
  with contextlib.ExitStack() as stack:
    if hasattr(registration, 'namespace'):      # gym 0.21 workaround kept from the reference
      stack.enter_context(registration.namespace(None))
    registration.register(id=ENV_ID, entry_point=ENTRY_POINT)
