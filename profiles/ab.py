"""A/B timing of library variants inside ONE process / one gpurun call (boxes of the pool differ).

  python profiles/ab.py [--obs] [--n 65536] build_ab/libble_a.so build_ab/libble_b.so ...
Each variant runs in a subprocess (BLE_HIP_LIB selects the library): the headline rollout (32-step
launches of ble_step_kernel, median of 15 x 64 steps) and, with --obs, the steady-state observation launch.
"""
import json, os, subprocess, sys

CHILD = r'''
import sys, os, time, json, statistics, numpy as np, torch
SUB = int(os.environ.get('AB_SUBSTEPS', '18'))
sys.path.insert(0, '.')
from balloon_learning_environment_amd import vec_state
sys.path.insert(0, 'tests')      # (the host-side state sampler is test tooling)
import reset_host
n = int(sys.argv[1]); do_obs = sys.argv[2] == '1'
sim = vec_state.VecSimulator(n)
field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)
sim.set_grid(torch.from_numpy(field).cuda())
state = reset_host.sample_initial_state(n, seed=1000)
sim.set_state(state)
K = 64
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
acts = torch.randint(0, 3, (K, n), dtype=torch.uint8, device='cuda', generator=gen)
rew = torch.zeros((K, n), device='cuda'); term = torch.zeros((K, n), dtype=torch.uint8, device='cuda')
def run():
  for k in range(0, K, 32):
    sim.step_n(acts[k:k+32], rew[k:k+32], term[k:k+32], substeps=SUB)
run(); torch.cuda.synchronize()
snap = {k: t.clone() for k, t in sim.state.items()}
ts = []
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for r in range(int(os.environ.get("AB_REPS", "15"))):
  for k, t in sim.state.items(): t.copy_(snap[k])
  torch.cuda.synchronize(); e0.record(); run(); e1.record(); torch.cuda.synchronize()
  ts.append(e0.elapsed_time(e1) * 1e3 / K)
out = {'us_per_step_median': statistics.median(ts), 'us_per_step_min': min(ts)}
if do_obs:
  sim.set_state(state)
  obs = torch.empty(n, 1099, device='cuda')
  noise = sim.wind_noise(seed=1234)
  for i in range(121):
    sim.step(acts[i % K], noise); sim.wind_noise(seed=1234, out=noise); sim.observe(noise, out=obs)
  torch.cuda.synchronize()
  to = []
  for i in range(8):
    sim.step(acts[i], noise); sim.wind_noise(seed=1234, out=noise)
    e0.record(); sim.observe(noise, out=obs); e1.record(); torch.cuda.synchronize()
    to.append(e0.elapsed_time(e1))
  out['obs_ms_median'] = statistics.median(to); out['obs_ms_min'] = min(to)
  if 'timing' in sys.argv[3]:
    t = obs[:, -20:].double().mean(0).cpu().numpy()
    out['marks'] = [float(v) for v in t]
try:
  sim.check_errors()
except Exception as e:
  out['error_flags'] = repr(e)[:80]
out['live'] = float((sim.state['status'] == 0).float().mean())
print('RESULT ' + json.dumps(out))
'''

def main():
  args = sys.argv[1:]
  do_obs = '--obs' in args
  n = 65536
  if '--n' in args:
    n = int(args[args.index('--n') + 1])
  libs = [a for a in args if a.endswith('.so')]
  for rnd in range(2):           # two passes: shows the run-to-run spread
    for lib in libs:
      env = dict(os.environ, BLE_HIP_LIB=os.path.abspath(lib))
      r = subprocess.run([sys.executable, '-c', CHILD, str(n), '1' if do_obs else '0', lib], env=env, capture_output=True, text=True)
      line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
      print(f'{os.path.basename(lib):40s}', line[0][7:] if line else ('FAILED ' + r.stderr[-400:]), flush=True)

if __name__ == '__main__':
  main()
