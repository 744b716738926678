"""Fused 32-step launches of the transition at a small batch, first with the four-waves-per-environment kernel
(ble_step_split_kernel), then with the one-lane kernel (ble_step_kernel), for the profiler:  python profiles/split_launches.py [n] [launches]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balloon_learning_environment_amd import _lib, vec_state  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 16
field = (np.random.default_rng(0).standard_normal(vec_state.GRID_SHAPE) * 5.0).astype(np.float32)
init = reset_host.sample_initial_state(n, seed=1000)
acts = torch.randint(0, 3, (32, n), dtype=torch.uint8, device='cuda')
rew = torch.zeros((32, n), device='cuda'); term = torch.zeros((32, n), dtype=torch.uint8, device='cuda')
for split in ('1', '0'):
  _lib.set_step_form(split)
  sim = vec_state.VecSimulator(n); sim.set_grid(field); sim.set_state(init)
  for _ in range(launches):
    sim.step_n(acts, rew, term)
  torch.cuda.synchronize()
  sim.check_errors()
