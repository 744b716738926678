"""Per-phase instruction counts of ble_observe_kernel (profile build: profiles/build_variant.sh phase '-DBLE_OBS_INSTR_HEADER="../../profiles/instr/ble_observe_instr.h"' -DBLE_OBS_PHASE_PROFILE, see
profiles/prof_obs_phases.sh).  After the window is full, launches that return after phase k (stop code in bits 8.. of
`append`; nothing is committed) are issued in the order stop = 1, 2, 3 (three each), then one complete launch: the counters of a phase are
differences of consecutive groups.   python profiles/obs_phases.py [n_envs]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balloon_learning_environment_amd import vec_state, device as dev, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sim = vec_state.VecSimulator(n)
field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)
sim.set_grid(torch.from_numpy(field).cuda())
sim.set_state(reset_host.sample_initial_state(n, seed=1000))
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
acts = torch.randint(0, 3, (64, n), dtype=torch.uint8, device='cuda', generator=gen)
obs = torch.empty(n, 1099, device='cuda')
noise = sim.wind_noise(seed=1234)
for i in range(124):
  sim.step(acts[i % 64], noise); sim.wind_noise(seed=1234, out=noise); sim.observe(noise, out=obs)
sim.step(acts[0], noise); sim.wind_noise(seed=1234, out=noise)
torch.cuda.synchronize()
for stop in (1, 2, 3, 0):
  for _ in range(3 if stop else 1):      # (a completed launch commits the window: only the first one is a steady-state slide)
    code = sim.lib.ble_observe_f32(ctypes.byref(sim._struct), sim.grid.data_ptr(), sim.grid_env_stride, dev.ptr(noise),
                                   sim._obs_reset.data_ptr(), ctypes.byref(sim._gp_struct), 1 | (stop << 8), obs.data_ptr(),
                                   sim.err_flags.data_ptr(), sim.n, dev.stream_ptr(sim.device))
    _lib.check(code, 'ble_observe_f32')
    torch.cuda.synchronize()
print('done')
