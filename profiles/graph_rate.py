"""Steps/s of VecBalloonEnv (noise + transition + masked reset + observation per step) with eager
launches and with the step captured in a HIP graph, for small batches (launch-bound)."""
import sys, time
import torch
sys.path.insert(0, '.')
from balloon_learning_environment_amd.env import balloon_env
for n in (64, 1024, 8192):
  for graph in (False, True):
    env = balloon_env.VecBalloonEnv(n, seed=1, wind_noise=True)
    env.reset()
    a = torch.ones(n, dtype=torch.uint8, device='cuda')
    for _ in range(125):
      env.step(a)
    if graph:
      env.capture_graph()
    torch.cuda.synchronize(); t = time.perf_counter(); k = 200
    for _ in range(k):
      env.step(a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('N=%5d %s: %.3f ms/step, %.3g env-steps/s' % (n, 'graph' if graph else 'eager', dt / k * 1e3, n * k / dt))
