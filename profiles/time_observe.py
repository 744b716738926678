"""Per-launch timing of ble_observe_f32 while the WindGP window fills, and -- with a timing build
(hipcc ... -DBLE_OBS_TIMING -o lib.so; BLE_HIP_LIB=lib.so) -- the in-kernel cycle marks quoted in
DESIGN.md 3b.  Usage: python profiles/time_observe.py [n_envs]"""
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from balloon_learning_environment_amd import vec_state
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sim = vec_state.VecSimulator(n)
field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)
sim.set_grid(torch.from_numpy(field).cuda())
sim.reset_device(seed=1)
obs = torch.empty(n, 1099, device='cuda')
acts = torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda')
for i in range(125):
  sim.step(acts)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  sim.observe(out=obs)
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  if i in (0, 1, 2, 5, 10, 20, 40, 60, 80, 100, 118, 119, 120, 124):
    print(i + 1, 'obs in window: %.3f ms  -> %.3g env-obs/s' % (dt * 1e3, n / dt), flush=True)
sim.check_errors()
import os
if os.environ.get('BLE_HIP_LIB'):
  t = obs[:, -7:].double().mean(0).cpu().numpy()
  names = ['phase0 (hist, elev table, column)', 'phase1 (ambient | newton | K)', 'phase2 cholesky', 'phase3 alpha', 'phase4 queries', '', '']
  for a, b in zip(names, t):
    print('%-40s %10.0f cycles' % (a, b))
if os.environ.get('BLE_HIP_LIB'):
  r = obs[:, -12:-9].double().mean(0).cpu().numpy()
  print('phase-1 roles: ambient %.0f, newton %.0f, K build %.0f cycles' % tuple(r))
