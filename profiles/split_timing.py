"""Where the four waves of ble_step_split_kernel spend a step (timing build: bash profiles/build_variant.sh split_timing
-DBLE_SPLIT_TIMING; run with BLE_HIP_LIB=build_ab/libble_split_timing.so).  Shader-clock cycles per role and section."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balloon_learning_environment_amd import _lib, device as dev, vec_state  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = 32
_lib.set_step_form('1')
sim = vec_state.VecSimulator(n)
sim.set_grid((np.random.default_rng(0).standard_normal(vec_state.GRID_SHAPE) * 5.0).astype(np.float32))
sim.set_state(reset_host.sample_initial_state(n, seed=1000))
acts = torch.randint(0, 3, (k, n), dtype=torch.uint8, device='cuda')
rew = torch.zeros((k, n), device='cuda'); term = torch.zeros((k, n), dtype=torch.uint8, device='cuda')
dbg = torch.zeros(64 * k, dtype=torch.int64, device='cuda')
for rep in range(3):
  dbg.zero_()
  code = sim.lib.ble_step_n_f32(ctypes.byref(sim._struct), acts.data_ptr(), sim.grid.data_ptr(), 0, None, rew.data_ptr(), term.data_ptr(),
                                sim.err_flags.data_ptr(), dbg.data_ptr(), n, 18, k, dev.stream_ptr(sim.device))
  assert code == 0
  torch.cuda.synchronize()
d = dbg.cpu().numpy()[:32].reshape(4, 8).astype(np.float64)
names = ['per-step part', 'map barrier wait', 'stride rhs', 'publish + barrier', 'reads after barrier', 'end of step']
roles = ['0 vertical', '1 thermal', '2 sun+power', '3 envelope+ACS']
print(f'n = {n}, {k} steps per launch; cycles per wave per agent step (mean over waves)')
for r in range(4):
  waves = d[r, 7]
  per = d[r, :6] / waves / k
  print(f'  role {roles[r]:15s} total {per.sum():8.0f} | ' + ' | '.join(f'{nm} {v:7.0f}' for nm, v in zip(names, per)))
