"""Histogram of reachable levels per environment after 125 steps (needs the BLE_OBS_TIMING profiling build, see profiles/instr/ble_observe_instr.h: BLE_HIP_LIB=...).
   python profiles/reach_histogram.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from balloon_learning_environment_amd import vec_state
n = 65536
sim = vec_state.VecSimulator(n)
field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)
sim.set_grid(torch.from_numpy(field).cuda())
sim.reset_device(seed=1)
obs = torch.empty(n, 1099, device='cuda')
acts = torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda')
for i in range(125):
  sim.step(acts); sim.observe(out=obs)
r = obs[:, 1099 - 3].cpu().numpy().astype(int)
h = np.bincount(r, minlength=190)
print({k: int(v) for k, v in enumerate(h) if v})
print('>=125', (r >= 125).mean(), '>=127', (r >= 127).mean(), '>=129', (r>=129).mean())
