"""The observation leg alone (for rocprofv3): 121 window-filling step/noise/observe triples + 8 steady-state ones.
  python profiles/obs_only.py [n_envs]"""
import sys, statistics, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balloon_learning_environment_amd import vec_state
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
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
ts = []
for i in range(129):
  sim.step(acts[i % 64], noise); sim.wind_noise(seed=1234, out=noise)
  e0.record(); sim.observe(noise, out=obs); e1.record(); torch.cuda.synchronize()
  ts.append(e0.elapsed_time(e1))
sim.check_errors()
print('observe launch ms: first %.3f, at 60 obs %.3f, steady (last 8) median %.3f' % (ts[0], ts[59], statistics.median(ts[-8:])))
