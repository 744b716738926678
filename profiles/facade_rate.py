import time, sys
sys.path.insert(0, '.')
from balloon_learning_environment_amd.env import balloon_env, features
for name, fac in (('device observation', features.perciatelli_feature_constructor), ('host observation', features.PerciatelliFeatureConstructor)):
  env = balloon_env.BalloonEnv(seed=3, feature_constructor_factory=fac)
  for i in range(20): env.step(i % 3)
  t = time.perf_counter(); n = 150
  for i in range(n): env.step(i % 3)
  dt = time.perf_counter() - t
  print('BalloonEnv.step with %s: %.1f steps/s (%.2f ms/step)' % (name, n / dt, dt / n * 1e3))
