"""Per-launch timing of ble_observe_f32 while the WindGP window fills, and -- with a timing build
(profiles/build_variant.sh obs_timing '-DBLE_OBS_INSTR_HEADER="../../profiles/instr/ble_observe_instr.h"' -DBLE_OBS_TIMING; BLE_HIP_LIB=build_ab/libble_obs_timing.so) -- the in-kernel cycle marks quoted in
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
  # layout written by the -DBLE_OBS_TIMING epilogue of ble_observe_kernel (csrc/ble_observe.h)
  o = obs.double().mean(0).cpu().numpy()
  D = 1099
  marks = o[D - 11:D - 5]
  names = ['phase 0 (prologue, elevation table, ring) -> B1', 'compaction + phase 1 (ambient | newton | drop) -> B3',
           'refit cholesky (0 when the factor is carried)', '-', 'diagonal-block inverses', 'sweep + features']
  for a, b in zip(names, marks):
    print('%-55s %10.0f cycles' % (a, b))
  print('phase-1 roles: ambient %.0f, newton + range %.0f, drop rows 0-63 %.0f, drop rows 64+ %.0f cycles' % tuple(o[D - 16:D - 12]))
  print('phase-0 sub-marks from kernel start: prologue issued %.0f, nodes+site ready %.0f, table filled %.0f' % tuple(o[D - 20:D - 17]))
  print('sweep sub-marks from its start (core done | specials in LDS | sums reduced | per-level tail | padding): wave 0 %s ; wave 1 %s' % (' '.join('%.0f' % v for v in o[D - 34:D - 29]), ' '.join('%.0f' % v for v in o[D - 29:D - 24])))
  print('arrival at the first barrier, waves 0-3: %.0f %.0f %.0f %.0f' % tuple(o[D - 38:D - 34]))
  print('sweep core of wave 1, end of row block 0..7 from the sweep start: ' + ' '.join('%.0f' % v for v in o[D - 46:D - 38]))
  print('tiles %.2f, reachable levels %.1f, share with a 9th tile %.3f' % (o[D - 4], o[D - 3], o[D - 2]))
