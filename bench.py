#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised BLE transition on MI355X.

  python bench.py --gpus 1 --steps 192 --warmup 32
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the headline config): 65 536 environments PER GPU,
random policy, one decoded wind grid shared by all environments (synthetic N(0, 5^2) m/s
float32 field, seed 0), initial conditions drawn like BalloonArena.reset
(reset_host.sample_initial_state).  One "step" = one agent step (180 s = 18 x 10 s
substeps + wind lookup + 3 safety layers + reward/terminal) of every environment of the
rank; ble_step_n_f32 runs up to 32 consecutive steps per launch of ble_step_kernel (the
random policy's actions are known up front, so the state stays in registers between steps).  Weak scaling: per-GPU work is fixed; with N > 1 the
grid is broadcast once over RCCL and rewards/terminals are gathered to rank 0 every 32
steps on a side stream (inside the timed region).  Terminated environments are frozen by the kernel
and are NOT counted: value = (sum over timed steps of live environments) / seconds.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGORITHMIC_BYTES_PER_ENV_STEP = 280      # 152 B state + 128 B grid gather
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
GATHER_EVERY = 32


def cpu_baseline(state, actions, field, seconds_target=10.0):
  """Times the CPU oracle (oracle/ble_oracle.c, fp64 restatement pinned to the reference)
  on the host cores of this box.  A reported baseline, not the optimisation target."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import numpy as np
  import oracle
  cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  n = state['x'].size        # the whole 65 536-env batch: enough work per OpenMP region for every core
  ost = oracle.new_state(n)
  for k in oracle.FLOAT_FIELDS:
    ost[k][:] = state[k][:n].astype(np.float64)
  for k in oracle.U8_FIELDS:
    ost[k][:] = state[k][:n]
  ost['start_unix'][:] = state['start_unix'][:n]; ost['time_elapsed_s'][:] = state['time_elapsed_s'][:n]
  ost['sunrise_h'][:] = state['start_unix'][:n] + state['sunrise_h_rel'][:n]
  ost['sunset'][:] = state['start_unix'][:n] + state['sunset_rel'][:n]
  oracle.step(ost, actions[0][:n], field=field, threads=cores)   # warm-up (page in, spin up threads)
  # The container may expose more hardware threads than its CFS quota lets it run (the GPU boxes show
  # 256 threads under a 16-CPU quota; a 256-thread OpenMP team then spends its time throttled).  The
  # team size is the quota; `cores` reports the threads actually used.
  cores_seen, quota = cores, None
  try:
    q, per = open('/sys/fs/cgroup/cpu.max').read().split()
    quota = None if q == 'max' else float(q) / float(per)
  except Exception:
    try:
      q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
      quota = q / per if q > 0 else None
    except Exception:
      quota = None
  if quota is not None:
    cores = max(1, min(cores, int(round(quota))))
  oracle.step(ost, actions[1][:n], field=field, threads=cores)
  steps = 0; live = 0; t0 = time.perf_counter()
  while True:
    live += int((ost['status'] == 0).sum())
    oracle.step(ost, actions[(steps + 1) % len(actions)][:n], field=field, threads=cores)
    steps += 1
    if time.perf_counter() - t0 > seconds_target or steps >= 512:
      break
  dt = time.perf_counter() - t0
  # single-thread figure on a slice of the same batch (about 3 s)
  m = 2048
  one = oracle.new_state(m)
  for k in one:
    one[k][:] = ost[k][:m]
  oracle.step(one, actions[0][:m], field=field, threads=1)
  t1 = time.perf_counter(); s1 = 0; live1 = 0
  while time.perf_counter() - t1 < 3.0 and s1 < 16:
    live1 += int((one['status'] == 0).sum())
    oracle.step(one, actions[(s1 + 1) % len(actions)][:m], field=field, threads=1)
    s1 += 1
  dt1 = time.perf_counter() - t1
  return {'value': live / dt, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
          'sample': f'{n} envs x {steps} agent steps of the same workload, fp64 C oracle with OpenMP over envs, {dt:.1f} s',
          'value_single_thread': live1 / dt1, 'sample_single_thread': f'{m} envs x {s1} agent steps, 1 thread, {dt1:.1f} s',
          'host_cpu_count': os.cpu_count(), 'affinity_threads': cores_seen, 'cgroup_cpu_quota': quota}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=192)
  ap.add_argument('--warmup', type=int, default=32)
  ap.add_argument('--envs-per-gpu', type=int, default=65536)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--active-count', action='store_true', help='analysis only: use the in-kernel live-env counter')
  ap.add_argument('--substeps', type=int, default=18, help='analysis only: physics substeps per agent step (18 = the metric)')
  ap.add_argument('--per-env-grids', action='store_true',
                  help='BASELINE config 5 shape: every env flies in its own forecast, decoded on the device by the '
                       'VAE-decoder restatement (synthetic weights); use with --envs-per-gpu 32768 (10.4 GB of grids)')
  ap.add_argument('--observe', type=int, default=0, metavar='N',
                  help='extra leg (not part of `value`): N timed step+observation pairs with the full 1099-feature '
                       'Perciatelli observation (ble_observe_f32) after the WindGP window (120 observations) has filled')
  args = ap.parse_args()

  import numpy as np
  import torch
  import torch.distributed as dist
  from balloon_learning_environment_amd import distributed as bdist
  from balloon_learning_environment_amd import reset_host
  from balloon_learning_environment_amd import vec_state

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
  assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU path)'
  dev_index = local_rank % torch.cuda.device_count()   # (== local_rank on a real multi-GPU node)
  torch.cuda.set_device(dev_index)
  device = torch.device('cuda', dev_index)
  if world > 1:
    backend = os.environ.get('BLE_DIST_BACKEND', 'nccl')   # nccl = RCCL over xGMI; gloo only for single-GPU smoke tests
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=device)
    else:
      dist.init_process_group(backend)

  n = args.envs_per_gpu
  k_total = args.steps + args.warmup
  # ---- synthetic inputs (host, seeded), then resident in HBM before the timed region
  state = reset_host.sample_initial_state(n, seed=1000 + rank)
  sim = vec_state.VecSimulator(n, device)
  sim.set_state(state)
  grid = torch.zeros(vec_state.GRID_SHAPE, dtype=torch.float32, device=device)
  field = None
  if rank == 0:
    field = (np.random.default_rng(0).standard_normal(vec_state.GRID_SHAPE) * 5.0).astype(np.float32)
    grid.copy_(torch.from_numpy(field))
  bdist.broadcast_grid(grid, src=0)                     # once per field, over xGMI when world > 1
  decode_ms = None
  if args.per_env_grids:
    # config 5: no broadcast at all -- every rank decodes its own latents into per-env grids
    from balloon_learning_environment_amd.env import generative_wind_field
    sampler = generative_wind_field.GenerativeWindFieldSampler(device=device, seed=0)
    latents = sampler.sample_latents(n, seed=100 + rank)
    grids = torch.empty((n, 21, 21, 10, 9, 2), dtype=torch.float32, device=device)
    sampler.decode(latents[:256], grids[:256]); torch.cuda.synchronize()
    d0 = torch.cuda.Event(enable_timing=True); d1 = torch.cuda.Event(enable_timing=True)
    d0.record(); sampler.decode(latents, grids); d1.record(); torch.cuda.synchronize()
    decode_ms = d0.elapsed_time(d1)
    sim.set_grid(grids, per_env=True)
  else:
    sim.set_grid(grid)
  gen = torch.Generator(device=device); gen.manual_seed(7 + rank)
  actions = torch.randint(0, 3, (k_total, n), dtype=torch.uint8, device=device, generator=gen)
  rewards = torch.zeros((k_total, n), dtype=torch.float32, device=device)
  terminals = torch.zeros((k_total, n), dtype=torch.uint8, device=device)
  active = torch.zeros((k_total, vec_state.COUNT_SLOTS), dtype=torch.int64, device=device)
  gatherer = bdist.OutputGatherer(GATHER_EVERY, n, device, world) if world > 1 else None

  def run(k0, k1):
    k = k0
    while k < k1:
      c = min(GATHER_EVERY, k1 - k)
      sim.step_n(actions[k:k + c], rewards[k:k + c], terminals[k:k + c], active[k:k + c] if args.active_count else None, substeps=args.substeps)
      if gatherer is not None and c == GATHER_EVERY:
        gatherer.gather(rewards[k:k + c], terminals[k:k + c])
      k += c
    if gatherer is not None:
      gatherer.wait()

  run(0, args.warmup)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()            # the kernels are launched on torch's current stream
  run(args.warmup, k_total)
  ev1.record()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  elapsed = bdist.max_over_ranks(elapsed, device)
  sim.check_errors()

  # live environments per step: an env is stepped iff it was not terminal after the previous
  # step (terminated envs are frozen by the kernel); counted after the timed region
  live_per_step = n - terminals[args.warmup - 1:k_total - 1].to(torch.int64).sum(dim=1) if args.warmup > 0 else None
  if live_per_step is None:
    live_per_step = torch.cat([torch.tensor([n], device=device), n - terminals[:k_total - 1].to(torch.int64).sum(dim=1)])
  if args.active_count:
    assert torch.equal(active[args.warmup:].sum(dim=1), live_per_step), 'in-kernel counter disagrees'
  live_steps = float(live_per_step.sum().item())
  live_steps_all = bdist.sum_over_ranks(live_steps, device)
  value = live_steps_all / elapsed
  # one launch of ble_step_kernel = up to GATHER_EVERY consecutive agent steps (state kept in registers)
  n_launches = -(-args.steps // GATHER_EVERY)
  kernel_ms = ev0.elapsed_time(ev1) / n_launches         # avg launch duration incl. inter-launch gaps
  bytes_per_launch = ALGORITHMIC_BYTES_PER_ENV_STEP * (live_steps / n_launches)
  achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
  traffic = None
  pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
  if os.path.exists(pmc_path):
    try:
      traffic = json.load(open(pmc_path)).get('hbm_bytes_per_launch')
    except Exception:
      traffic = None

  # ---- optional leg: observation-inclusive stepping, reported beside `value`
  observe_leg = None
  if args.observe > 0:
    sim.set_state(state)                                  # fresh episodes
    obs = torch.empty(n, 1099, dtype=torch.float32, device=device)
    one = torch.zeros(n, dtype=torch.uint8, device=device)
    sim.reset_observation_history()
    obs_gatherer = bdist.ObservationGatherer(n, 1099, device, world) if world > 1 else None
    fill = 121                                            # 6 h window = 120 observations; 121st call slides it
    # forecast != truth: the additive wind noise (ble_wind_noise_f32) is evaluated at the balloons once
    # per step -- it is both the next step's ground-truth term and this observation's error term
    noise = sim.wind_noise(seed=1234)
    for i in range(fill):
      sim.step(actions[i % k_total], noise)
      sim.wind_noise(seed=1234, out=noise)
      sim.observe(noise, out=obs)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
    t_obs = 0.0; t_pair = 0.0
    for i in range(args.observe):
      e0.record(); sim.step(actions[(fill + i) % k_total], noise); sim.wind_noise(seed=1234, out=noise)
      e1.record(); sim.observe(noise, out=obs)
      if obs_gatherer is not None:                         # "observations gathered back" (north star), 4 396 B/env
        obs_gatherer.gather(obs); obs_gatherer.wait()
      e2.record()
      torch.cuda.synchronize()
      t_obs += e1.elapsed_time(e2); t_pair += e0.elapsed_time(e2)
    sim.check_errors()
    live = float((sim.state['status'] == 0).sum().item())
    observe_leg = {'pairs': args.observe, 'ms_per_observation_launch': t_obs / args.observe,
                   'ms_per_step_plus_observation': t_pair / args.observe,
                   'env_observations_per_s': n * args.observe / (t_obs * 1e-3),
                   'env_steps_per_s_with_observation': n * args.observe / (t_pair * 1e-3),
                   'window_observations': 120, 'obs_bytes_per_env': 4396, 'live_env_fraction': live / n,
                   'includes_gather_to_rank0': world > 1,
                   'kernel': 'ble_observe_kernel (fp64 WindGP: factor slid in HBM, MFMA forward substitution)',
                   # 144 v_mfma_f64_16x16x4 per 16-column tile x 8 tiles (two error vectors + the ~120 reachable
                   # levels) x 2 048 flop + ~0.6 MFLOP of fp64 VALU (factor slide, kernel evaluations) per
                   # environment; fp64 matrix and vector peaks are both 78.6 TFLOP/s
                   'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': 78.6,
                                'achieved': n * (144 * 8 * 2048 + 0.6e6) / (t_obs / args.observe * 1e-3) / 1e12,
                                'frac': n * (144 * 8 * 2048 + 0.6e6) / (t_obs / args.observe * 1e-3) / 1e12 / 78.6,
                                'traffic': None, 'algorithmic_flop_per_env': 144 * 8 * 2048 + 0.6e6}}
    del one

  # what actually bounds the kernel (from the committed PMC summary of the same command, if present)
  issue = None
  try:
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r01_summary.json')))['derived']
    issue = {'wave_issue_utilisation': d['active_inst_any_quad'] / d['wave_cycles_per_wave_quad'],
             'valu_insts_per_env_step': d['valu_insts_per_wave'] / 32.0, 'salu_insts_per_env_step': d['salu_insts_per_wave'] / 32.0,
             'source': 'profiles/r01_summary.json (rocprofv3 --pmc, per 32-step launch)'}
  except Exception:
    issue = None

  if rank == 0:
    out = {
        'metric': 'env-steps/sec at 65 536 parallel envs; achieved HBM GB/s fraction of peak',
        'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32+f64', 'data': 'synthetic',
        'config': {'workload': (f'{n} vectorised envs per GPU, random policy, per-env forecasts decoded on the device '
                                '(BASELINE.json configs[4] shape: VAE path, synthetic weights)' if args.per_env_grids else
                                f'{n} vectorised envs per GPU, random policy, one decoded wind grid '
                                '(BASELINE.json configs[2]: 65 536 envs, 1xMI355X headline)'),
                   'envs_per_gpu': n, 'global_envs': n * world, 'substeps_per_step': args.substeps,
                   'live_env_fraction_end': float(live_per_step[-1].item()) / n,
                   'per_env_grids': bool(args.per_env_grids), 'decode_ms': decode_ms,
                   'parallelism': f'env-sharded x{world}, grid broadcast once, reward/terminal gather to rank 0 every {GATHER_EVERY} steps'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'kernel': 'ble_step_kernel', 'kernel_ms': kernel_ms, 'agent_steps_per_launch': args.steps / n_launches,
                     'kernel_us_per_agent_step': 1e3 * kernel_ms * n_launches / args.steps,
                     'algorithmic_bytes_per_env_step': ALGORITHMIC_BYTES_PER_ENV_STEP,
                     'note': 'kernel is fp32/fp64-VALU and transcendental bound, not HBM bound (DESIGN.md)',
                     'instruction_issue': issue},
    }
    if observe_leg is not None:
      out['observe'] = observe_leg
    if world == 1 and not args.no_cpu_baseline:
      acts = actions[:64].cpu().numpy()
      out['cpu_baseline'] = cpu_baseline(state, list(acts), field)
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
