"""steps/s of the single-environment gym facade (BalloonEnv.step: configs[0]'s counterpart).
  python profiles/facade_rate.py
Three paths: the default (generative grid field + device feature constructor: one HIP graph and one pinned copy per step,
BalloonArena._step_fast), the same with the one-transfer step disabled (the general path: a handful of round trips per step, what
rounds 1-5 measured), and a forecast that is not a grid (SimpleStaticWindField: host wind lookups, the column handed to the kernel)."""
import sys
import time

sys.path.insert(0, '.')
from balloon_learning_environment_amd.env import balloon_arena, balloon_env, wind_field  # noqa: E402


def rate(env, n=300):
  for i in range(20):
    env.step(i % 3)
  t = time.perf_counter()
  for i in range(n):
    _, _, terminal, _ = env.step(i % 3)
    if terminal:
      env.reset()
  dt = time.perf_counter() - t
  return n / dt, dt / n * 1e3


print('BalloonEnv.step, default (one graph + one pinned copy per step): %.0f steps/s (%.3f ms/step)' % rate(balloon_env.BalloonEnv(seed=3)))
slow = balloon_env.BalloonEnv(seed=3)
slow.arena._fast_path_ok = lambda: False
print('BalloonEnv.step, general path (round trips per step):            %.0f steps/s (%.3f ms/step)' % rate(slow))
print('BalloonEnv.step, SimpleStaticWindField (forecast not a grid):     %.0f steps/s (%.3f ms/step)' %
      rate(balloon_env.BalloonEnv(seed=3, wind_field_factory=wind_field.SimpleStaticWindField)))
