"""Optional gym registration of this package's BalloonEnv under the reference's environment id
(the reference does the same for its own class in env/gym.py:20-40).

gym is not a dependency (it is absent from the build image): `register_env()` raises ImportError without
it, and nothing else in the package imports this module.  BalloonEnv itself is duck-typed to the gym 0.21
API (step / reset / seed / render / close, action_space, observation_space, reward_range, metadata).
"""

ENV_ID = 'BalloonLearningEnvironment-v0'
ENTRY_POINT = 'balloon_learning_environment_amd.env.balloon_env:BalloonEnv'


def _already_registered(registry) -> bool:
  specs = getattr(registry, 'env_specs', None)          # gym <= 0.21 keeps a dict there; later versions are dict-like
  return ENV_ID in (specs if specs is not None else registry)


def register_env() -> None:
  """Idempotent: a second call (gym's plugin loader imports entry points more than once) is a no-op."""
  import gym.envs.registration as reg                   # imported here so that importing this module never needs gym

  if _already_registered(reg.registry):
    return
  root_namespace = getattr(reg, 'namespace', None)      # gym 0.21 needs it to register an id without a namespace
  if root_namespace is None:
    reg.register(id=ENV_ID, entry_point=ENTRY_POINT)
  else:
    with root_namespace(None):
      reg.register(id=ENV_ID, entry_point=ENTRY_POINT)
