"""Long closed-loop soak of VecBalloonEnv with its defaults (generative wind field with synthetic weights, wind noise,
auto-reset): `steps` agent steps of `n` environments under a random policy, HIP-graph replay after the first steps.
Checks at every `every`-th step: observations finite and inside the observation space, rewards in [0, 1], everybody flying
after the auto-reset, no error flag latched.  Prints the rate and the episode statistics.
  python profiles/soak_env.py [n_envs] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balloon_learning_environment_amd.env import balloon_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
every = 500
env = balloon_env.VecBalloonEnv(n, seed=3)
obs = env.reset()
space = env.observation_space
low, high = torch.from_numpy(space.low).cuda() - 1e-6, torch.from_numpy(space.high).cuda() + 1e-6
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
terminals = torch.zeros((), dtype=torch.int64, device='cuda'); reward_sum = torch.zeros((), dtype=torch.float64, device='cuda')
for k in range(4):
  env.step(torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda', generator=gen))
env.capture_graph()
t0 = time.perf_counter()
for k in range(steps):
  obs, reward, terminal = env.step(torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda', generator=gen))
  terminals += terminal.sum(); reward_sum += reward.sum(dtype=torch.float64)
  if k % every == 0 or k == steps - 1:
    assert torch.isfinite(obs).all(), k
    assert ((obs >= low) & (obs <= high)).all(), k
    assert ((reward >= 0) & (reward <= 1)).all(), k
    assert (env.arena.sim.state['status'] == 0).all(), k
    env.check_errors()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'{n} envs x {steps} steps in {dt:.1f} s = {n * steps / dt:.3g} env-steps/s closed loop (HIP graph); episodes ended {int(terminals)} '
      f'({int(terminals) / (n * steps) * 960:.2f} per 960 env-steps); mean reward {float(reward_sum) / (n * steps):.3f}; no error flag, every checked observation finite and in range')
