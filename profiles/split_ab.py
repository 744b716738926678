"""A/B of the two forms of the transition at small batches: one lane per environment (ble_step_kernel) against one environment on
four wavefronts (ble_step_split_kernel), forced with BLE_STEP_SPLIT; fused 32-step launches and single-step launches.
  python profiles/split_ab.py [n ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balloon_learning_environment_amd import _lib, vec_state  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402

modes = os.environ.get('SPLIT_AB_MODES', '0,4').split(',')       # BLE_STEP_SPLIT values: 0 one lane, 2 / 4 wavefronts per environment
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768]
field = (np.random.default_rng(0).standard_normal(vec_state.GRID_SHAPE) * 5.0).astype(np.float32)
for n in sizes:
  init = reset_host.sample_initial_state(n, seed=1000)
  acts = torch.randint(0, 3, (64, n), dtype=torch.uint8, device='cuda')
  rew = torch.zeros((32, n), device='cuda'); term = torch.zeros((32, n), dtype=torch.uint8, device='cuda')
  for split in modes:
    _lib.set_step_form(split)
    sim = vec_state.VecSimulator(n); sim.set_grid(field)
    res = {}
    for label, reps in (('fused32', 12), ('single', 200)):
      sim.set_state(init)
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      if label == 'fused32':
        sim.step_n(acts[:32], rew, term); torch.cuda.synchronize()
        sim.set_state(init)
        e0.record()
        for r in range(reps):
          sim.step_n(acts[(r % 2) * 32:(r % 2) * 32 + 32], rew, term)
        e1.record(); torch.cuda.synchronize()
        res[label] = e0.elapsed_time(e1) * 1e3 / (reps * 32)
      else:
        for r in range(8):
          sim.step(acts[r])
        torch.cuda.synchronize(); e0.record()
        for r in range(reps):
          sim.step(acts[r % 64])
        e1.record(); torch.cuda.synchronize()
        res[label] = e0.elapsed_time(e1) * 1e3 / reps
    live = float((sim.state['status'] == 0).float().mean().item())
    print(f'n={n:6d} split={split}: fused {res["fused32"]:.2f} us/step = {n / res["fused32"] * 1e6:.3e} env-steps/s; '
          f'single-step launch {res["single"]:.2f} us; live at end {live:.3f}', flush=True)
_lib.set_step_form(None)
