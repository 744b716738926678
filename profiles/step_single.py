"""ONE agent step per launch (ble_step_f32, the policy-in-the-loop shape), 300 launches of 65 536 environments, for
rocprofv3 --kernel-trace --stats:   python profiles/step_single.py [n_envs]"""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for i in range(44):
  sim.step(acts[i % 64])
torch.cuda.synchronize()
e0.record()
for i in range(256):
  sim.step(acts[i % 64])
e1.record(); torch.cuda.synchronize()
sim.check_errors()
print('ble_step_f32, %d envs: %.2f us per launch back to back (256 launches)' % (n, e0.elapsed_time(e1) * 1e3 / 256))
