"""Where the time of ONE ble_step_kernel launch goes, wave by wave (profiling build, see profiles/instr/ble_step_instr.h):
   bash profiles/build_variant.sh step_timing -DBLE_STEP_BLOCK=64 '-DBLE_STEP_INSTR_HEADER="../../profiles/instr/ble_step_instr.h"'
   (one wave per workgroup: the marks are indexed by blockIdx.x; the product build runs four waves per workgroup)
   BLE_HIP_LIB=build_ab/libble_step_timing.so python profiles/step_wave_timing.py [n_envs]
Launches of K = 1 and K = 32 agent steps; every wave records its entry / exit wall clock (100 MHz) and the shader-clock length
of its phases.  Prints the launch span (first entry -> last exit), the dispatch ramp, the finish dispersion and the phases."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balloon_learning_environment_amd import vec_state, device as dev
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sim = vec_state.VecSimulator(n)
field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)
sim.set_grid(torch.from_numpy(field).cuda())
sim.set_state(reset_host.sample_initial_state(n, seed=1000))
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
acts = torch.randint(0, 3, (32, n), dtype=torch.uint8, device='cuda', generator=gen)
rew = torch.zeros((32, n), device='cuda'); term = torch.zeros((32, n), dtype=torch.uint8, device='cuda')
waves = (n + 63) // 64
dbg = torch.zeros((waves, 64), dtype=torch.int64, device='cuda')


def launch(k):
  code = sim.lib.ble_step_n_f32(ctypes.byref(sim._struct), acts.data_ptr(), sim.grid.data_ptr(), 0, None, rew.data_ptr(), term.data_ptr(),
                                sim.err_flags.data_ptr(), dbg.data_ptr(), n, 18, k, dev.stream_ptr(sim.device))
  assert code == 0


for _ in range(3):
  launch(32)
torch.cuda.synchronize()
snap = {k: t.clone() for k, t in sim.state.items()}
names = ['state loads landed', 'ACS cubics + barrier', 'per-episode constants', 'agent steps', 'stores acknowledged']
for k in (1, 1, 2, 32):
  for key, t in sim.state.items():
    t.copy_(snap[key])
  dbg.zero_()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record(); launch(k); e1.record(); torch.cuda.synchronize()
  d = dbg.cpu().numpy().astype(np.float64)
  t_in, t_out = d[:, 0] * 0.01, d[:, 1] * 0.01            # us
  span = t_out.max() - t_in.min()
  ramp = t_in - t_in.min()
  dur = t_out - t_in
  fin = t_out - t_in.min()
  clk = d[:, 7] / np.maximum(dur, 1e-9)                    # shader cycles per us
  q = lambda a: ' '.join(f'{np.percentile(a, p):7.2f}' for p in (0, 10, 50, 90, 99, 100))
  print(f'--- K = {k} agent steps per launch, {waves} waves; HIP events {e0.elapsed_time(e1) * 1e3:.1f} us; span first entry -> last exit {span:.2f} us')
  print(f'    percentiles                0      10      50      90      99     100')
  print(f'    wave entry after first  {q(ramp)}  us   (dispatch ramp)')
  print(f'    wave duration           {q(dur)}  us')
  print(f'    wave exit after first   {q(fin)}  us   (finish dispersion: the launch ends with the last one)')
  print(f'    shader clock            {np.median(clk):.0f} cycles/us')
  for j, name in enumerate(names):
    us = d[:, 2 + j] / np.median(clk)
    print(f'    {name:24s}{q(us)}  us')
  c = np.median(clk)
  steps_us = d[:, 8:8 + k] / c                     # [waves, k]
  print(f'    agent step 0 of the launch {q(steps_us[:, 0])}  us')
  if k > 1:
    print(f'    agent steps 1 .. {k - 1}         {q(steps_us[:, 1:].ravel())}  us   (every wave, every step)')
    print(f'    per-wave mean of steps 1..  {q(steps_us[:, 1:].mean(1))}  us')
  ev = d[:, 40:44]
  ev_names = ['exact solar chain', 'layer transition crossed', 'p +- 1 Pa straddles a transition', 'window above 21 km']
  for j, name in enumerate(ev_names):
    hit = ev[:, j] > 0
    print(f'    {name:34s} lanes {int(ev[:, j].sum()):7d} in {int(hit.sum()):5d} waves; agent-step phase of those waves {np.median(d[hit, 5]) / c if hit.any() else 0:8.2f} us median vs {np.median(d[~hit, 5]) / c if (~hit).any() else 0:8.2f} without')
  sec = d[:, 48:54] / c / k
  sec_names = ['atmosphere + safety layers', 'ephemeris', 'wind blend', 'three solar nodes', 'substep loop (18)', 'status + reward']
  print('    sections of an agent step, us per step (median over waves):', ', '.join(f'{nm} {np.median(sec[:, j]):.2f}' for j, nm in enumerate(sec_names)),
        f'; sum {np.median(sec.sum(1)):.2f}')
  by_xcd = [np.median(dur[np.arange(waves) % 8 == x]) for x in range(8)]
  print('    median shader clock by workgroup index mod 8 (XCD), cycles/us:', ' '.join(f'{np.median(clk[np.arange(waves) % 8 == x]):.0f}' for x in range(8)))
  print('    median wave duration by workgroup index mod 8 (XCD):', ' '.join(f'{v:.2f}' for v in by_xcd))
  late = np.argsort(dur)[-6:]
  print('    slowest waves (wave, duration us, agent-step phase us, events):', [(int(w), round(float(dur[w]), 2), round(float(d[w, 5] / c), 2), [int(v) for v in ev[w]]) for w in late])
sim.check_errors()
