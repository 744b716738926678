"""Per-launch fixed cost of ble_step_kernel: launches of K = 1 .. 32 agent steps (4 back to back per event pair).
   python profiles/step_launch_cost.py"""
import os, sys, statistics, numpy as np, torch
sys.path.insert(0, '.')
from balloon_learning_environment_amd import vec_state
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402
n = 65536
sim = vec_state.VecSimulator(n)
field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)
sim.set_grid(torch.from_numpy(field).cuda())
state = reset_host.sample_initial_state(n, seed=1000)
sim.set_state(state)
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
acts = torch.randint(0, 3, (32, n), dtype=torch.uint8, device='cuda', generator=gen)
rew = torch.zeros((32, n), device='cuda'); term = torch.zeros((32, n), dtype=torch.uint8, device='cuda')
sim.step_n(acts, rew, term); torch.cuda.synchronize()
snap = {k: t.clone() for k, t in sim.state.items()}
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
res = {}
for K in (1, 2, 4, 8, 16, 20, 32):
  ts = []
  for r in range(12):
    for k, t in sim.state.items(): t.copy_(snap[k])
    torch.cuda.synchronize()
    e0.record()
    for rep in range(4): sim.step_n(acts[:K], rew[:K], term[:K])
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 4)
  res[K] = statistics.median(ts)
  print(K, 'steps/launch: %.1f us per launch, %.2f us per step' % (res[K], res[K] / K))
S = (res[32] - res[16]) / 16
print('per step %.2f us, fixed per launch %.1f us (from 16 vs 32), from 1-step launch: %.1f' % (S, res[32] - 32 * S, res[1] - S))
