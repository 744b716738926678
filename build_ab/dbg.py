import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from balloon_learning_environment_amd import vec_state
n, steps = 65536, 126
rng = np.random.default_rng(12)
field = (rng.standard_normal((21, 21, 10, 9, 2)) * 6.0).astype(np.float32)
gen = torch.Generator(device='cuda')
sim = vec_state.VecSimulator(n)
sim.set_grid(torch.from_numpy(field).cuda())
sim.reset_device(seed=31)
gen.manual_seed(5)
obs = torch.empty(n, 1099, dtype=torch.float32, device='cuda')
keep = {}
for i in range(steps + 1):
  if i > 0:
    sim.step(torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda', generator=gen))
  noise = torch.randn((n, 2), dtype=torch.float32, device='cuda', generator=gen) * 1.5
  sim.observe(noise, out=obs)
  if i >= 118: keep[i] = obs.cpu().numpy().copy()
np.save(sys.argv[1], np.stack([keep[i] for i in sorted(keep)]))
