"""Fused rollouts in the ground-truth wind (ABI 3: noise generated in-kernel) at small batches: the one-lane kernel against the four-wave
kernel.  python profiles/noise_fused_ab.py [n ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balloon_learning_environment_amd import _lib, vec_state  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 32768]
field = (np.random.default_rng(0).standard_normal(vec_state.GRID_SHAPE) * 5.0).astype(np.float32)
for n in sizes:
  init = reset_host.sample_initial_state(n, seed=1000)
  acts = torch.randint(0, 3, (32, n), dtype=torch.uint8, device='cuda')
  rew = torch.zeros((32, n), device='cuda'); term = torch.zeros((32, n), dtype=torch.uint8, device='cuda')
  for split in ('0', '4'):
    _lib.set_step_form(split)
    sim = vec_state.VecSimulator(n); sim.set_grid(field); sim.set_state(init)
    sim.step_n(acts, rew, term, noise_seed=5); torch.cuda.synchronize()
    sim.set_state(init)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(8):
      sim.step_n(acts, rew, term, noise_seed=5)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (8 * 32)
    print(f'n={n:6d} waves={split}: ground-truth wind, fused {us:.2f} us/step = {n / us * 1e6:.3e} env-steps/s', flush=True)
_lib.set_step_form(None)
