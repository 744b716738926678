#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised BLE transition on MI355X.

  python bench.py                                   # N = 1, BASELINE.json configs[2] + every 1-GPU leg
  python bench.py --gpus 1 --steps 192 --warmup 32
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W [--config 3|4]
  python bench.py --gpus N ...                      # no launcher (WORLD_SIZE unset): bench.py starts the N ranks itself

One "step" = one agent step (180 s = 18 x 10 s substeps + wind lookup + 3 safety layers +
reward/terminal) of every environment of the rank.  The random policy's actions are known up front, so
ble_step_n_f32 runs up to 32 consecutive steps per launch of ble_step_kernel (state in registers; 8 when N > 1, see GATHER_EVERY).

Presets (`--config i` = BASELINE.json configs[i]; the default, 2, is the headline):
  1  4 096 envs, one decoded wind grid, 1 GPU
  2  65 536 envs PER GPU, one decoded wind grid (weak scaling when N > 1: grid broadcast once over RCCL; the reward /
     terminal rows of EVERY launch -- 8 steps per launch when sharded, so that each exchange overlaps the next launch, or
     fewer for the last launch of a region -- are gathered to rank 0 on a side stream, inside the timed region:
     `config.exchanges` counts them)
  3  65 536 envs GLOBAL, sharded contiguously over the N ranks (strong scaling), same exchanges
  4  32 768 envs per GPU (262 144 on 8), every env flies in its own forecast decoded on the device by the
     VAE-decoder restatement (synthetic weights); no broadcast

Timing (the driver's contract): W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier +
torch.cuda.synchronize() on both sides, MAX over ranks.  That K-step region is repeated `--reps` times
(default 31) from the same post-warm-up state -- a single 20-step region is one 0.5 ms kernel launch, far too
short for one sample -- and the MEDIAN repetition is the reported value (min / max / first beside it).
The wall-clock repetitions launch through argument views built beforehand and carry no event records; the kernel's own
AVERAGE duration (roofline.achieved) comes from 9 more, event-bracketed repetitions of the same region.
Terminated environments are frozen by the kernel and are NOT counted.

`n_gpus` is the number of ranks that actually joined (an all-reduce), and must equal --gpus.

Besides `value`, the default run reports (rank 0; every leg is the same code path as the headline):
  `policy_in_the_loop`  ONE launch per agent step (ble_step_f32, what an agent loop issues): >= 200 launches, the median
                 HIP-event time of a launch and the rate of the back-to-back sequence
  `roofline.traffic`    HBM bytes per launch MEASURED IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
                 separate passes, kernel trace only) over the headline leg with the same --steps / --warmup, i.e. the
                 same launch shape; 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md's gfx950 correction).  null (with
                 `traffic_note`) if rocprofv3 is not on the box or the run is itself being profiled
  `configs`      env-steps/s of configs[1], [3]'s per-GPU shard (8 192 envs; the real sharded run when N > 1)
                 and [4]'s per-GPU share (32 768 envs with per-env grids), and the single-env facade
                 (configs[0]'s counterpart: BalloonEnv.step with the device observation)
  `config.ground_truth_wind`  the headline rollout flown in WindField.get_ground_truth (noise generated in-kernel, ABI 3)
  `roofline`     bound = "valu-issue": frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the headline launch shape measured IN
                 THIS RUN (a third --pmc pass; `instruction_issue` holds the counters); `hbm_formal` (algorithmic
                 bytes / kernel time / 8 TB/s) and `hbm_measured` (PMC traffic / kernel time) are the secondary fields
  `per_rank`     kernel time and exposed exchange time of a timed region on every rank (HIP events)
  `observe`      the closed-loop cost: step + wind noise + the 1099-feature observation (ble_observe_f32)
                 with a full WindGP window, its own roofline and measured traffic
  `observe_configs`  the same closed loop at the batch sizes of the other BASELINE configs on one GPU -- 4 096 (configs[1]), 8 192
                 (one GPU's share of configs[3]), 32 768 with per-environment grids (configs[4]'s share): ms per observation
                 launch, env-steps/s of step + noise + observation, fp64 fraction; N = 1 only
  `cpu_baseline` the fp64 C oracle on this box's host cores (N = 1 only)
Prints ONE JSON line on rank 0.
"""
import argparse
import csv
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGORITHMIC_BYTES_PER_ENV_STEP = 280      # 152 B state + 128 B grid gather
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6                   # MI355X_MICROARCH.md: fp64 vector = fp64 matrix peak
# Agent steps per launch = per exchange.  One rank: 32 (the per-launch fixed cost, ~9 us, is paid once per 32 steps and there
# is nothing to exchange).  Several ranks: every launch's reward / terminal rows go to rank 0 on a side stream while the NEXT
# launch computes, so only the last gather of a region is exposed -- 8-step launches keep that tail short (5 B x 65 536 x 8 =
# 2.6 MB per rank over its own xGMI link) at three launches' fixed cost per 20-step region instead of one; with 32 the
# driver's 20-step region would end with an unoverlapped 6.5 MB-per-rank exchange (BLE_STEPS_PER_GATHER overrides).
GATHER_EVERY = 32
GATHER_EVERY_SHARDED = 8
PRESETS = {1: 'configs[1]: 4 096 vectorised envs, random policy, one decoded wind field, 1xMI355X',
           2: 'configs[2]: 65 536 vectorised envs per GPU, random policy, one decoded wind grid (headline)',
           3: 'configs[3]: 65 536 envs sharded across the GPUs, wind field RCCL-broadcast, rewards/terminals gathered',
           4: 'configs[4]: 32 768 envs per GPU (262 144 on 8), per-env forecasts decoded on the device (VAE path, synthetic weights)'}


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


class Rollout:
  """One preset's workload on this rank: state + grid resident in HBM, `time_reps` runs the timed region."""

  def __init__(self, n, device, rank, world, *, per_env_grids=False, shared_field=None, seed_base=1000,
               steps=192, warmup=32, substeps=18, noise_seed=None, env_offset=None, vehicle=None):
    import numpy as np
    import torch
    from balloon_learning_environment_amd import distributed as bdist
    from balloon_learning_environment_amd import vec_state
    self.torch, self.bdist, self.np = torch, bdist, np
    self.n, self.device, self.rank, self.world = n, device, rank, world
    self.steps, self.warmup, self.substeps = steps, warmup, substeps
    # None: forecast wind; else WindField.get_ground_truth, noise generated in-kernel.  ONE seed for the whole job:
    # reset and noise streams are keyed by (seed, GLOBAL env index, episode) -- this rank's environments are env_offset .. + n of
    # the global batch (ABI 4; with local indices all shards flew identical fields: ADVICE r4)
    self.noise_seed = noise_seed
    self.env_offset = rank * n if env_offset is None else int(env_offset)
    k_total = steps + warmup
    # synthetic inputs: the product's own episode reset (ble_reset_at_f32, sample = 1: the reference's initial-condition
    # distributions from a Philox stream keyed by (seed, global env index, episode)), resident in HBM.  The legs that
    # restart the episodes copy this snapshot back.  (Rounds 1-4 drew them with a NumPy sampler that is test tooling now.)
    self.sim = vec_state.VecSimulator(n, device, env_offset=self.env_offset)
    if vehicle:           # a flight vehicle other than the reference's default (ABI 5): the kernels' VehicleRt instantiations
      self.sim.set_vehicle(**vehicle)
    self.sim.reset_device(seed=seed_base)
    self.sim.check_errors()
    self.initial_state = {k: t.clone() for k, t in self.sim.state.items()}
    self.decode_ms = None
    if per_env_grids:     # no broadcast at all: every rank decodes its own latents into per-env grids
      from balloon_learning_environment_amd.env import generative_wind_field
      sampler = generative_wind_field.GenerativeWindFieldSampler(device=device, seed=0)
      # one latent per GLOBAL environment index (with a per-rank generator stream every shard decoded the same fields unless
      # the seed was varied by hand): the shards' fields are the unsharded batch's
      latents = sampler.sample_latents_keyed(torch.arange(n, device=device) + self.env_offset, torch.ones(n, dtype=torch.int64, device=device), seed=100)
      grids = torch.empty((n,) + tuple(vec_state.GRID_SHAPE), dtype=torch.float32, device=device)
      sampler.decode(latents[:min(n, 256)], grids[:min(n, 256)]); torch.cuda.synchronize()
      d0 = torch.cuda.Event(enable_timing=True); d1 = torch.cuda.Event(enable_timing=True)
      d0.record(); sampler.decode(latents, grids); d1.record(); torch.cuda.synchronize()
      self.decode_ms = d0.elapsed_time(d1)
      self.sim.set_grid(grids, per_env=True)
      del sampler, latents
    else:
      self.sim.set_grid(shared_field)
    gen = torch.Generator(device=device); gen.manual_seed(7 + rank)
    self.actions = torch.randint(0, 3, (k_total, n), dtype=torch.uint8, device=device, generator=gen)
    self.launches_per_region = -(-steps // GATHER_EVERY)
    # one receive slot per launch of a region: when the region's wait() returns, rank 0 holds EVERY launch's rows
    self.gatherer = bdist.OutputGatherer(GATHER_EVERY, n, device, world, slots=self.launches_per_region) if world > 1 else None


  def restart_episodes(self):
    """Back to the initial states of this preset (device-to-device copies)."""
    for k, t in self.sim.state.items():
      t.copy_(self.initial_state[k])

  def initial_host_state(self):
    return {k: t.cpu().numpy() for k, t in self.initial_state.items()}

  def plan(self, k0, k1):
    """The launches of steps k0 .. k1 - 1, prepared once (VecSimulator.prepare_step_n: checks and argument marshalling
    happen here, outside any timed region): (launch, packed output buffer, reward rows, terminal rows) per launch.  A
    launch's rewards and terminals live in ONE buffer (5 B per env-step) so that they go to rank 0 in one message."""
    out = []
    k = k0
    while k < k1:
      c = min(GATHER_EVERY, k1 - k)
      buf, r, t = self.bdist.packed_output_block(c, self.n, self.device)
      out.append((self.sim.prepare_step_n(self.actions[k:k + c], r, t, None, substeps=self.substeps, noise_seed=self.noise_seed), buf, r, t))
      k += c
    return out

  def run(self, k0, k1, plan=None, on_compute_enqueued=None):
    # every launch's rows -- the last, shorter one of a region too -- go to rank 0; the region ends when they have arrived
    self.bdist.run_region(plan if plan is not None else self.plan(k0, k1), self.gatherer, on_compute_enqueued)

  def time_reps(self, reps):
    """Warm-up, snapshot, then `reps` x (restore snapshot; barrier+sync; K steps; sync+barrier).
    Returns per-repetition lists: wall seconds (MAX over ranks), live env-steps (SUM over ranks), HIP-event ms."""
    torch, dist_mod = self.torch, self.torch.distributed
    barrier = (lambda: dist_mod.barrier()) if self.world > 1 else (lambda: None)
    self.run(0, self.warmup)
    torch.cuda.synchronize()
    snap = {k: t.clone() for k, t in self.sim.state.items()}
    live0 = float((self.sim.state['status'] == 0).sum().item())
    wall, live, ev_ms, ev_exposed_ms = [], [], [], []
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True); evk = torch.cuda.Event(enable_timing=True)
    plan = self.plan(self.warmup, self.warmup + self.steps)
    g0 = (self.gatherer.gathers, self.gatherer.rows_gathered) if self.gatherer is not None else (0, 0)
    # The wall-clock repetitions carry no event records (two records cost ~10 us of host time per region: 2.4 % of a
    # 20-step region); the kernel's own duration comes from separate, event-bracketed repetitions of the same region.
    n_ev = min(reps, 9)
    for rep_i in range(reps + n_ev):
      with_events = rep_i >= reps
      for k, t in self.sim.state.items():
        t.copy_(snap[k])
      torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
      t0 = time.perf_counter()
      if with_events: ev0.record()            # the kernels are launched on torch's current stream
      # (evk: behind the last launch, in front of the compute stream's wait for the exchanges -- ev0 .. evk is this rank's kernel
      # time, evk .. ev1 the exchange time that no launch hides)
      self.run(self.warmup, self.warmup + self.steps, plan, (lambda: evk.record()) if with_events else None)
      if with_events: ev1.record()
      torch.cuda.synchronize(); barrier()
      dt = time.perf_counter() - t0
      torch.cuda.synchronize()
      if with_events:
        ev_ms.append(ev0.elapsed_time(evk)); ev_exposed_ms.append(evk.elapsed_time(ev1))
        continue
      wall.append(self.bdist.max_over_ranks(dt, self.device))
      # an env is stepped iff it was not terminal after the previous step; counted outside the timed region
      term = torch.cat([item[3] for item in plan])[:self.steps - 1].to(torch.int64).sum(dim=1)
      l = live0 + float((self.n - term).sum().item())
      live.append(self.bdist.sum_over_ranks(l, self.device))
    self.sim.check_errors()
    self.live_fraction_end = float((self.sim.state['status'] == 0).sum().item()) / max(1, self.n)
    regions = reps + n_ev
    if self.gatherer is not None:      # counted, not assumed: exchanges and agent-step rows per timed region
      self.gathers_per_region = (self.gatherer.gathers - g0[0]) / regions
      self.rows_gathered_per_region = (self.gatherer.rows_gathered - g0[1]) / regions
      assert self.gathers_per_region == self.launches_per_region and self.rows_gathered_per_region == self.steps
    else:
      self.gathers_per_region = self.rows_gathered_per_region = 0
    self.ev_exposed_ms = ev_exposed_ms
    return wall, live, ev_ms

  def per_rank(self, ev_ms):
    """Per rank: kernel time and exposed (unhidden) exchange time of a timed region, HIP events on the compute stream --
    so that a multi-GPU line decomposes itself."""
    torch = self.torch
    mine = torch.tensor([1e3 * statistics.fmean(ev_ms), 1e3 * statistics.fmean(self.ev_exposed_ms)], dtype=torch.float64, device=self.device)
    if self.world > 1:
      rows = [torch.zeros_like(mine) for _ in range(self.world)]
      torch.distributed.all_gather(rows, mine)
    else:
      rows = [mine]
    rows = [r.cpu().tolist() for r in rows]
    return {'kernel_us_per_timed_region': [r[0] for r in rows], 'exposed_exchange_us_per_timed_region': [r[1] for r in rows],
            'what': 'HIP events on each rank\'s compute stream: first launch .. last launch | last launch .. the exchanges waited for'}

  def summary(self, reps):
    wall, live, ev_ms = self.time_reps(reps)
    per_rank = self.per_rank(ev_ms)
    rates = [l / w for l, w in zip(live, wall)]
    order = sorted(range(reps), key=lambda i: rates[i])
    med = order[reps // 2]
    ev_med = statistics.median(ev_ms)
    return {'env_steps_per_s': rates[med], 'env_steps_per_s_min': min(rates), 'env_steps_per_s_max': max(rates),
            'env_steps_per_s_first_repetition': rates[0], 'repetitions': reps, 'steps_per_repetition': self.steps,
            'ms_per_step': 1e3 * wall[med] / self.steps, 'ms_per_step_min': 1e3 * min(wall) / self.steps,
            'ms_per_step_max': 1e3 * max(wall) / self.steps,
            'kernel_ms_median': ev_med / self.launches_per_region, 'kernel_ms_min': min(ev_ms) / self.launches_per_region,
            'kernel_ms_mean': statistics.fmean(ev_ms) / self.launches_per_region, 'kernel_event_repetitions': len(ev_ms),
            'live_env_steps_per_repetition': live[med], 'envs_per_gpu': self.n, 'global_envs': int(round(self.bdist.sum_over_ranks(float(self.n), self.device))),
            'live_env_fraction_end': self.live_fraction_end, 'decode_ms': self.decode_ms,
            'gathers_per_region': self.gathers_per_region, 'rows_gathered_per_region': self.rows_gathered_per_region,
            'launches_per_region': self.launches_per_region, 'per_rank': per_rank}


def observe_leg(roll, pairs, world, measure=False):
  """step + wind noise + 1099-feature observation with a full WindGP window (closed-loop cost)."""
  torch, bdist = roll.torch, roll.bdist
  sim, n, device = roll.sim, roll.n, roll.device
  k_total = roll.actions.shape[0]
  roll.restart_episodes()                                # fresh episodes
  obs = torch.empty(n, 1099, dtype=torch.float32, device=device)
  sim.reset_observation_history()
  obs_gatherer = bdist.ObservationGatherer(n, 1099, device, world, mode='gather') if world > 1 else None
  fill = 121                                            # 6 h window = 120 observations; 121st call slides it
  # forecast != truth: the additive wind noise (ble_wind_noise_f32) is evaluated at the balloons once
  # per step -- it is both the next step's ground-truth term and this observation's error term
  noise = sim.wind_noise(seed=1234)
  for i in range(fill):
    sim.step(roll.actions[i % k_total], noise)
    sim.wind_noise(seed=1234, out=noise)
    sim.observe(noise, out=obs)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
  t_obs, t_pair = [], []
  for i in range(pairs):
    e0.record(); sim.step(roll.actions[(fill + i) % k_total], noise); sim.wind_noise(seed=1234, out=noise)
    e1.record(); sim.observe(noise, out=obs)
    if obs_gatherer is not None:                         # "observations gathered back" (north star), 4 396 B/env
      obs_gatherer.gather(obs); obs_gatherer.wait()
    e2.record()
    torch.cuda.synchronize()
    t_obs.append(e1.elapsed_time(e2)); t_pair.append(e0.elapsed_time(e2))
  # N > 1: the same pair with each consumer of the observation blocks (distributed.ObservationGatherer): one learner on
  # rank 0 (gather: timed above), a data-parallel learner (all_to_all), a policy replica per rank (local: no exchange).
  # MAX over ranks of the mean pair time; bytes and link-bound time of each from observation_exchange_model (arithmetic).
  exchange_modes = None
  if world > 1:
    exchange_modes = {'gather': dict(obs_gatherer.model, ms_per_step_plus_observation=bdist.max_over_ranks(statistics.fmean(t_pair), device),
                                     ms_observation_plus_exchange=bdist.max_over_ranks(statistics.fmean(t_obs), device))}
    for mode in ('all_to_all', 'local'):
      if mode == 'all_to_all' and n % world != 0:
        continue
      try:        # (a backend without this collective -- gloo on device tensors -- must not cost the line; symmetric on all ranks)
        og = bdist.ObservationGatherer(n, 1099, device, world, mode=mode)
        tp, to = [], []
        for i in range(pairs):
          e0.record(); sim.step(roll.actions[(fill + pairs + i) % k_total], noise); sim.wind_noise(seed=1234, out=noise)
          e1.record(); sim.observe(noise, out=obs)
          og.gather(obs); og.wait()
          e2.record()
          torch.cuda.synchronize()
          to.append(e1.elapsed_time(e2)); tp.append(e0.elapsed_time(e2))
        exchange_modes[mode] = dict(og.model, ms_per_step_plus_observation=bdist.max_over_ranks(statistics.fmean(tp), device),
                                    ms_observation_plus_exchange=bdist.max_over_ranks(statistics.fmean(to), device))
        del og
      except Exception as e:
        exchange_modes[mode] = dict(bdist.observation_exchange_model(mode, n, 1099, world), error=repr(e)[:300])
  sim.check_errors()
  live = float((sim.state['status'] == 0).sum().item())
  ms_obs, ms_pair = statistics.fmean(t_obs), statistics.fmean(t_pair)      # the average launch duration (HIP events), as for the headline
  # ALGORITHMIC work per env-observation, independent of how the kernel tiles it: the forward substitution
  # V = Lt^-1 [k_new | e_0 | K*^T] on 119 rows x (2 + 121 reachable levels) columns = n (n - 1) m flop; the kernel matrix
  # (119 x 123 entries x ~40 flop: distance, square root, exp); the four sums per level (119 x 123 x 8); the window slide
  # (7 021 entries x 4); the 721-entry elevation table (~200 flop per entry); the 22 cold starts; all fp64.
  # (The kernel EXECUTES more: 16-row MFMA tiles pad 119 rows to 128 and 123 columns to 128 -- 1 042 MFMAs x 2 048 flop
  # = 2.13 MFLOP for the 1.73 MFLOP substitution -- and the block inverses; `executed_mfma_tflops` reports that rate.)
  # Algorithmic bytes: 4 396 out + 2 x 60 960 factor, drop vector and zeta in/out + 152 state + 3 072 ring
  n_rows, n_cols = 119, 2 + 121
  flop = (n_rows * (n_rows - 1) * n_cols + n_rows * n_cols * 40 + n_rows * n_cols * 8 + 7021 * 4 + 721 * 200
          + 22 * 6 * 2 * 150)
  mfma_flop_executed = 1042 * 2048
  obs_bytes = 4396 + 2 * 60960 + 152 + 3072
  traffic, traffic_detail, traffic_note = None, None, 'not measured in this run'
  if measure and world == 1 and n == 65536:
    traffic_detail, traffic_note = measure_traffic('ble_observe_kernel', [sys.executable, os.path.join(ROOT, 'profiles', 'obs_only.py'), str(n)],
                                                   launches_per_group=8, groups=1)
    traffic = traffic_detail['bytes'] if traffic_detail else None
  tf = n * flop / (ms_obs * 1e-3) / 1e12
  return {'pairs': pairs, 'ms_per_observation_launch': ms_obs, 'ms_per_observation_launch_median': statistics.median(t_obs), 'ms_per_observation_launch_min': min(t_obs),
          'ms_per_step_plus_observation': ms_pair,
          'env_observations_per_s': n / (ms_obs * 1e-3), 'env_steps_per_s_with_observation': n / (ms_pair * 1e-3),
          'window_observations': 120, 'obs_bytes_per_env': 4396, 'live_env_fraction': live / n,
          'includes_gather_to_rank0': world > 1, 'exchange_modes': exchange_modes,
          'kernel': 'ble_observe_kernel (fp64 WindGP: factor carried in HBM and slid with a stored drop vector, MFMA forward substitution)',
          'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': FP64_PEAK_TFLOPS, 'achieved': tf, 'frac': tf / FP64_PEAK_TFLOPS,
                       'traffic': traffic, 'traffic_detail': traffic_detail, 'traffic_note': traffic_note,
                       'algorithmic_flop_per_env': flop, 'algorithmic_bytes_per_env': obs_bytes,
                       'executed_mfma_tflops': n * mfma_flop_executed / (ms_obs * 1e-3) / 1e12,
                       'hbm_gbs_algorithmic': n * obs_bytes / (ms_obs * 1e-3) / 1e9, 'hbm_gbs_measured': (traffic / (ms_obs * 1e-3) / 1e9) if traffic else None}}


def policy_in_the_loop_leg(roll, launches=256):
  """ONE kernel launch per agent step -- ble_step_f32, what VecBalloonArena.step and every agent loop
  (`action = agent.step(reward, obs); env.step(action)`, eval/eval_lib.py:158-171) issues -- on the headline batch:
  the back-to-back rate of `launches` launches and the median HIP-event duration of a single one."""
  torch = roll.torch
  sim, n = roll.sim, roll.n
  roll.restart_episodes()
  k_total = roll.actions.shape[0]
  for i in range(8):
    sim.step(roll.actions[i % k_total])
  torch.cuda.synchronize()
  # (a) per-launch duration: one event pair around each of 64 launches
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  per = []
  for i in range(64):
    e0.record(); sim.step(roll.actions[(8 + i) % k_total]); e1.record()
    torch.cuda.synchronize()
    per.append(e0.elapsed_time(e1) * 1e3)
  # (b) the sequence: `launches` launches back to back, one synchronisation at the end.  The environments actually
  # stepped (live when a step began) are counted by the kernel itself (ble_step_f32's active_count)
  torch.cuda.synchronize()
  c0 = int(sim.active_count.item())
  t0 = time.perf_counter(); e0.record()
  for i in range(launches):
    sim.step(roll.actions[(72 + i) % k_total])
  e1.record(); torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  sim.check_errors()
  stepped = int(sim.active_count.item()) - c0
  return {'launches': launches, 'envs': n,
          'kernel': ('ble_step_split_kernel' if n <= 32768 else 'ble_step_kernel') + ' via ble_step_f32 (n_steps = 1)',
          'us_per_launch_event_median': statistics.median(per), 'us_per_launch_event_min': min(per),
          'us_per_step_back_to_back': e0.elapsed_time(e1) * 1e3 / launches, 'env_steps_per_s': stepped / dt,
          'note': 'one launch per agent step: the per-launch fixed cost (state in/out, per-episode constants, wave launch and '
                  'finish dispersion) is paid every step; the fused headline (ble_step_n_f32) pays it once per 32 steps'}


_T0 = [time.perf_counter()]


def tick(what):
  """BLE_BENCH_TIMING=1: wall-clock seconds of every leg on stderr (where a run's time goes)."""
  if os.environ.get('BLE_BENCH_TIMING'):
    now = time.perf_counter()
    print(f'[bench timing] rank {os.environ.get("RANK", "0")} {what}: {now - _T0[0]:.1f} s', file=sys.stderr, flush=True)
    _T0[0] = now


PMC_BUDGET_S = [240.0]      # wall-clock budget of ALL nested profiler passes of one bench.py run (--pmc-budget-s)


def _pmc_pass(counters, cmd, workdir, timeout_s):
  """One rocprofv3 --pmc pass (kernel trace only, as MI355X_MICROARCH.md prescribes) of `cmd` collecting `counters` (a
  name or a list that fits one pass); returns the rows of the counter-collection CSV."""
  counters = [counters] if isinstance(counters, str) else list(counters)
  timeout_s = min(timeout_s, PMC_BUDGET_S[0])
  if timeout_s < 20:
    raise TimeoutError('profiler budget of this run spent (--pmc-budget-s)')
  out_dir = os.path.join(workdir, counters[0])
  full = ['rocprofv3', '--kernel-trace', '--pmc'] + counters + ['--output-format', 'csv', '-d', out_dir, '-o', 'p', '--'] + cmd
  env = dict(os.environ, TMPDIR='/tmp')
  t0 = time.perf_counter()
  try:
    subprocess.run(full, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
  finally:
    PMC_BUDGET_S[0] -= time.perf_counter() - t0
  rows = []
  for base, _, files in os.walk(out_dir):
    for f in files:
      if f.endswith('counter_collection.csv'):
        with open(os.path.join(base, f)) as fh:
          rows += [r for r in csv.DictReader(fh) if r['Counter_Name'] in counters]
  return rows


def being_profiled():
  return any(k.startswith('ROCPROF') for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')


def measure_traffic(kernel, cmd, launches_per_group, groups, timeout_s=100):
  """HBM bytes per launch of `kernel`, measured now: FETCH_SIZE and WRITE_SIZE in two separate rocprofv3 --pmc passes of
  `cmd`; the LAST groups x launches_per_group dispatches of the kernel are the steady-state / timed ones.  Units: the
  counters are in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read, hence the guide's
  x 2 on the fetch side (`bytes` below; `bytes_raw` is the uncorrected sum)."""
  if shutil.which('rocprofv3') is None:
    return None, 'rocprofv3 not on PATH'
  if being_profiled():
    return None, 'this run is itself under a profiler: nested PMC pass skipped'
  work = tempfile.mkdtemp(prefix='ble_pmc_', dir='/tmp')
  try:
    got = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
      rows = [r for r in _pmc_pass(counter, cmd, work, timeout_s) if kernel in r['Kernel_Name']]
      rows.sort(key=lambda r: int(r['Dispatch_Id']))
      take = rows[-groups * launches_per_group:]
      if len(take) < launches_per_group:
        return None, f'{counter}: only {len(rows)} dispatches of {kernel} found'
      got[counter] = 1024.0 * sum(float(r['Counter_Value']) for r in take) / len(take)
    return {'bytes': 2.0 * got['FETCH_SIZE'] + got['WRITE_SIZE'], 'bytes_raw': got['FETCH_SIZE'] + got['WRITE_SIZE'],
            'fetch_size_bytes': got['FETCH_SIZE'], 'write_size_bytes': got['WRITE_SIZE'],
            'dispatches_averaged': groups * launches_per_group,
            'how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) inside this bench.py run; '
                   'bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction of MI355X_MICROARCH.md)'}, None
  except Exception as e:                  # never lose the line over the profiler
    return None, f'PMC pass failed: {e!r}'[:300]
  finally:
    shutil.rmtree(work, ignore_errors=True)


ISSUE_COUNTERS = ('SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAIT_ANY')


def measure_issue(kernel, cmd, launches_per_group, groups, agent_steps_per_launch, timeout_s=100):
  """Where the kernel's time goes when HBM is not the bound, measured now: ONE more rocprofv3 --pmc pass (SQ counters only,
  kernel trace only) of `cmd`, averaged over the last groups x launches_per_group dispatches of `kernel`.
  wave_issue_utilisation = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (cycles a wave had an instruction in flight / cycles it was
  resident); valu_issue_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; instruction counts per wave and agent step."""
  if shutil.which('rocprofv3') is None:
    return None, 'rocprofv3 not on PATH'
  if being_profiled():
    return None, 'this run is itself under a profiler: nested PMC pass skipped'
  work = tempfile.mkdtemp(prefix='ble_pmc_', dir='/tmp')
  try:
    rows = [r for r in _pmc_pass(list(ISSUE_COUNTERS), cmd, work, timeout_s) if kernel in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    take = set(ids[-groups * launches_per_group:])
    if len(take) < launches_per_group:
      return None, f'only {len(ids)} dispatches of {kernel} found'
    tot = {c: 0.0 for c in ISSUE_COUNTERS}
    for r in rows:
      if int(r['Dispatch_Id']) in take:
        tot[r['Counter_Name']] += float(r['Counter_Value'])
    if tot['SQ_WAVES'] <= 0 or tot['SQ_WAVE_CYCLES'] <= 0:
      return None, 'SQ counters came back empty'
    waves = tot['SQ_WAVES']
    per = float(agent_steps_per_launch)
    return {'wave_issue_utilisation': tot['SQ_ACTIVE_INST_ANY'] / tot['SQ_WAVE_CYCLES'],
            'valu_issue_frac': tot['SQ_ACTIVE_INST_VALU'] / tot['SQ_WAVE_CYCLES'],
            'wait_frac': tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES'],
            'valu_insts_per_env_step': tot['SQ_INSTS_VALU'] / waves / per, 'salu_insts_per_env_step': tot['SQ_INSTS_SALU'] / waves / per,
            'wave_quad_cycles_per_env_step': tot['SQ_WAVE_CYCLES'] / waves / per,
            'waves_per_launch': waves / len(take), 'dispatches_averaged': len(take), 'agent_steps_per_launch': per,
            'source': 'measured in this run: rocprofv3 --kernel-trace --pmc ' + ' '.join(ISSUE_COUNTERS) + ' (one pass; per wave = 64 environments)'}, None
  except Exception as e:
    return None, f'PMC pass failed: {e!r}'[:300]
  finally:
    shutil.rmtree(work, ignore_errors=True)


def facade_leg(steps=150):
  """BASELINE configs[0]'s counterpart on this framework: the single-env gym facade (BalloonEnv.step ->
  ble_step_f32 + ble_observe_f32 on one environment, host-synchronous like the reference's API)."""
  from balloon_learning_environment_amd.env import balloon_env
  env = balloon_env.BalloonEnv(seed=0)
  for i in range(20):
    env.step(i % 3)
  t = time.perf_counter()
  for i in range(steps):
    _, _, terminal, _ = env.step(i % 3)
    if terminal:
      env.reset()
  dt = time.perf_counter() - t
  return {'steps_per_s': steps / dt, 'ms_per_step': 1e3 * dt / steps, 'steps': steps,
          'what': 'BalloonEnv.step (single env, device observation, host-synchronous gym API)'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=192)
  ap.add_argument('--warmup', type=int, default=32)
  ap.add_argument('--config', type=int, default=2, choices=sorted(PRESETS), help='BASELINE.json configs[i] (see the module docstring)')
  ap.add_argument('--reps', type=int, default=31, help='repetitions of the timed K-step region (median reported)')
  ap.add_argument('--envs-per-gpu', type=int, default=None, help='override the preset size')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extras', action='store_true', help='only the headline leg (profiling)')
  ap.add_argument('--substeps', type=int, default=18, help='analysis only: physics substeps per agent step (18 = the metric)')
  ap.add_argument('--per-env-grids', action='store_true', help='same as --config 4 data layout at the chosen size')
  ap.add_argument('--observe', type=int, default=8, metavar='N',
                  help='timed step+observation pairs of the observation leg (0 = skip)')
  ap.add_argument('--traffic', choices=('auto', 'off'), default='auto',
                  help='auto: measure roofline.traffic and roofline.instruction_issue in this run with nested rocprofv3 --pmc passes '
                       '(N = 1, full run only; three of the headline, two of the observation leg)')
  ap.add_argument('--pmc-budget-s', type=float, default=240.0,
                  help='wall-clock budget of all nested profiler passes together; passes that no longer fit are skipped and say so')
  ap.add_argument('--noise-seed', type=int, default=None,
                  help='fly the HEADLINE in the ground-truth wind (noise generated in-kernel) instead of the forecast; the default '
                       'run reports both (config.ground_truth_wind)')
  args = ap.parse_args()
  PMC_BUDGET_S[0] = args.pmc_budget_s

  # ---- `python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, and hand the
  # result line of rank 0 through.  (Under torch.distributed.run WORLD_SIZE is set and this branch is not taken.)
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.path.insert(0, ROOT)
    from balloon_learning_environment_amd import distributed as bdist_spawn
    sys.exit(bdist_spawn.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

  import numpy as np
  import torch
  import torch.distributed as dist
  from balloon_learning_environment_amd import distributed as bdist
  from balloon_learning_environment_amd import vec_state

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks'
  assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU path)'
  backend = os.environ.get('BLE_DIST_BACKEND', 'nccl')     # nccl = RCCL over xGMI; gloo only for smoke tests that put several ranks on one GPU
  if world > 1 and backend == 'nccl':
    assert torch.cuda.device_count() >= world, (f'{world} ranks over RCCL need {world} GPUs, {torch.cuda.device_count()} visible '
                                                '(BLE_DIST_BACKEND=gloo runs a smoke test with several ranks on one GPU)')
  dev_index = local_rank % torch.cuda.device_count()   # (== local_rank on a real multi-GPU node)
  torch.cuda.set_device(dev_index)
  device = torch.device('cuda', dev_index)
  if world > 1:
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=device)
    else:
      dist.init_process_group(backend)
  global GATHER_EVERY
  if world > 1:
    GATHER_EVERY = max(1, min(32, int(os.environ.get('BLE_STEPS_PER_GATHER', GATHER_EVERY_SHARDED))))
  n_joined = bdist.joined_ranks(device)                  # counted, not read from the command line
  assert n_joined == args.gpus, f'{n_joined} ranks joined, --gpus {args.gpus}'
  if args.config == 3:
    assert 65536 % world == 0, 'configs[3] shards 65 536 environments into equal contiguous slices'

  # ---- the decoded wind grid: rank 0 owns it, everyone gets it by ONE broadcast (xGMI when world > 1)
  grid = torch.zeros(vec_state.GRID_SHAPE, dtype=torch.float32, device=device)
  field = None
  if rank == 0:
    field = (np.random.default_rng(0).standard_normal(vec_state.GRID_SHAPE) * 5.0).astype(np.float32)
    grid.copy_(torch.from_numpy(field))
  bdist.broadcast_grid(grid, src=0)

  def preset_size(cfg):
    return bdist.preset_layout(cfg, rank, world)['n_local']

  def make(cfg, n=None, steps=None, warmup=None, noise_seed=None, vehicle=None):
    n = n if n is not None else preset_size(cfg)
    return Rollout(n, device, rank, world, per_env_grids=(cfg == 4 or args.per_env_grids), shared_field=grid,
                   steps=steps or args.steps, warmup=args.warmup if warmup is None else warmup, substeps=args.substeps,
                   noise_seed=noise_seed, vehicle=vehicle)

  tick('start-up (imports, process group, grid broadcast)')
  # ---- headline leg
  head = make(args.config, n=args.envs_per_gpu, noise_seed=args.noise_seed)
  hs = head.summary(args.reps)
  tick('headline leg')
  n = head.n
  bytes_per_launch = ALGORITHMIC_BYTES_PER_ENV_STEP * (hs['live_env_steps_per_repetition'] / world / head.launches_per_region)
  achieved = bytes_per_launch / (hs['kernel_ms_mean'] * 1e-3) / 1e9      # the AVERAGE launch duration (HIP events on the launch stream)
  # HBM traffic of the launch shape timed above, measured in this run (rank 0 of a 1-GPU full run; see measure_traffic)
  traffic, traffic_detail, traffic_note = None, None, None
  if world == 1 and args.traffic == 'auto' and not args.no_extras:
    child = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup),
             '--config', str(args.config), '--reps', '3', '--no-extras', '--traffic', 'off', '--substeps', str(args.substeps)]
    if args.envs_per_gpu:
      child += ['--envs-per-gpu', str(args.envs_per_gpu)]
    if args.per_env_grids:
      child += ['--per-env-grids']
    if args.noise_seed is not None:
      child += ['--noise-seed', str(args.noise_seed)]
    traffic_detail, traffic_note = measure_traffic('ble_step_kernel', child, head.launches_per_region, groups=6)
    traffic = traffic_detail['bytes'] if traffic_detail else None
  else:
    traffic_note = 'not measured in this run (N > 1, --no-extras or --traffic off)'
  # the bound that actually applies (instruction issue at one wave per SIMD): a third PMC pass of the same child run
  issue, issue_note = None, None
  if world == 1 and args.traffic == 'auto' and not args.no_extras:
    issue, issue_note = measure_issue('ble_step_kernel', child, head.launches_per_region, groups=3,
                                      agent_steps_per_launch=args.steps / head.launches_per_region)
  tick('nested PMC passes')
  if issue is None:            # labelled fallback: the committed profile of an earlier build, NOT this run
    for tag in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
      try:
        d = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_summary.json')))['derived']
        per = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_summary.json'))).get('agent_steps_per_profiled_launch', 32.0)
        issue = {'wave_issue_utilisation': d['active_inst_any_quad'] / d['wave_cycles_per_wave_quad'],
                 'valu_issue_frac': d['active_inst_valu_quad'] / d['wave_cycles_per_wave_quad'],
                 'valu_insts_per_env_step': d['valu_insts_per_wave'] / per, 'salu_insts_per_env_step': d['salu_insts_per_wave'] / per,
                 'source': f'NOT measured in this run ({issue_note or "N > 1, --no-extras or --traffic off"}): profiles/{tag}_summary.json '
                           f'(rocprofv3 --pmc, per {int(per)}-step launch)'}
        break
      except Exception:
        continue

  # ---- the other 1-GPU legs (same code path, fewer repetitions)
  ground_truth = None
  configs = {f'configs[{args.config}]': {k: hs[k] for k in ('env_steps_per_s', 'env_steps_per_s_min', 'env_steps_per_s_max', 'envs_per_gpu', 'global_envs', 'ms_per_step')}}
  observe = None
  observe_configs = {}
  policy = None
  if not args.no_extras:
    extra_reps = max(5, min(11, args.reps))
    if rank == 0 and world == 1:
      policy = policy_in_the_loop_leg(head)
      configs[f'configs[{args.config}] one launch per step'] = {'env_steps_per_s': policy['env_steps_per_s'], 'envs_per_gpu': head.n,
                                                               'us_per_launch_event_median': policy['us_per_launch_event_median'],
                                                               'us_per_step_back_to_back': policy['us_per_step_back_to_back']}
    if args.noise_seed is None and not (args.config == 4 or args.per_env_grids):
      # the same rollout in the reference's OWN wind: WindField.get_ground_truth = forecast + noise (wind_field.py:125-145),
      # the noise generated inside ble_step_kernel at every step (ABI 3); SURVEY 8(d) defines the headline with the term off
      rn = make(args.config, n=args.envs_per_gpu, noise_seed=20240917)
      sn = rn.summary(extra_reps)
      ground_truth = {k: sn[k] for k in ('env_steps_per_s', 'env_steps_per_s_min', 'env_steps_per_s_max', 'ms_per_step', 'kernel_ms_mean', 'envs_per_gpu', 'global_envs')}
      ground_truth['what'] = ('the headline rollout flown in WindField.get_ground_truth: 10 harmonics of 4-D simplex noise per env-step '
                              'evaluated inside ble_step_kernel<noise> (same launch shape, exchanges and counting as the headline)')
      configs[f'configs[{args.config}] in the ground-truth wind (noise in-kernel)'] = ground_truth
      del rn
      tick('ground-truth-wind leg')
    if world == 1 and args.noise_seed is None and not (args.config == 4 or args.per_env_grids):
      # the same rollout with a flight vehicle handed in at run time (ABI 5: ble_state_f32.vehicle -> ble_step_kernel<VehicleRt>, its constants
      # in scalar registers instead of the instruction stream): another skin, drag coefficient, power system and valve
      veh = dict(envelope_max_superpressure=2500.0, envelope_cod=0.27, nighttime_power_load_w=150.0, daytime_power_load_w=110.0,
                 acs_valve_hole_diameter_m=0.05, battery_capacity_wh=3400.0)      # (the default envelope and masses: the sampler's pressures suit it)
      rv = make(args.config, n=args.envs_per_gpu, vehicle=veh)
      sv = rv.summary(extra_reps)
      configs[f'configs[{args.config}] with a run-time vehicle (ABI 5)'] = dict(
          {k: sv[k] for k in ('env_steps_per_s', 'env_steps_per_s_min', 'env_steps_per_s_max', 'ms_per_step', 'kernel_ms_mean', 'envs_per_gpu', 'live_env_fraction_end')},
          kernel='ble_step_kernel<VehicleRt> (one lane per environment whatever the batch size)', vehicle=veh)
      del rv
      tick('run-time vehicle leg')
    if args.observe > 0 and not (args.config == 4 or args.per_env_grids):
      observe = observe_leg(head, args.observe, world, measure=(args.traffic == 'auto'))
      tick('observation leg')
    head_initial_host_state = head.initial_host_state() if (world == 1 and not args.no_cpu_baseline) else None
    del head
    torch.cuda.empty_cache()
    for cfg in (1, 2, 3, 4):
      if cfg == args.config:
        continue
      if cfg in (1, 2) and world > 1:
        continue                                         # single-GPU presets
      size = None
      if cfg == 3 and world == 1:
        size = 8192                                      # one GPU's share of configs[3] on an 8-GPU node
      if os.environ.get('BLE_BENCH_SIDE_ENVS') and cfg in (3, 4):
        size = int(os.environ['BLE_BENCH_SIDE_ENVS'])    # (tests: the side legs' code path at a small batch)
      # side legs time 192-step regions after 32 warm-up steps whatever --steps / --warmup say (the default run's shape: six
      # 32-step launches on one rank): a 20-step region of a 10 us-per-step shard is 0.2 ms, of which the host's launch +
      # synchronise is 12 %, and five warm-up steps leave the first repetitions on a cold clock
      side_steps = int(os.environ.get('BLE_BENCH_SIDE_STEPS', '192'))       # (tests: over gloo -- two ranks on one GPU -- every gather goes through the host)
      r = make(cfg, n=size, steps=side_steps, warmup=32)
      s = r.summary(extra_reps)
      key = f'configs[{cfg}]' + (' per-GPU shard (8 192 of 65 536), 1 GPU' if (cfg == 3 and world == 1) else '')
      configs[key] = {k: s[k] for k in ('env_steps_per_s', 'env_steps_per_s_min', 'env_steps_per_s_max', 'envs_per_gpu', 'global_envs', 'ms_per_step', 'decode_ms',
                                        'steps_per_repetition', 'kernel_ms_mean')}
      configs[key]['workload'] = PRESETS[cfg]
      configs[key]['kernel'] = ('ble_step_split_kernel (one environment on four wavefronts: n <= 32 768)' if s['envs_per_gpu'] <= 32768
                                else 'ble_step_kernel (one lane per environment)')
      if rank == 0 and world == 1 and cfg in (1, 3):      # the policy-in-the-loop shape (one launch per agent step) at this batch size
        pl = policy_in_the_loop_leg(r)
        configs[key]['one_launch_per_step'] = {'us_per_step_back_to_back': pl['us_per_step_back_to_back'], 'env_steps_per_s': pl['env_steps_per_s'],
                                               'us_per_launch_event_median': pl['us_per_launch_event_median']}
      if world == 1 and args.observe > 0 and cfg in (1, 3, 4):
        # the closed loop every drop-in agent sees -- step + wind noise + observation with a full WindGP window -- at THIS config's
        # batch size (so that a multi-GPU run of configs[3] / [4] can be read against a one-GPU closed-loop number)
        ol = observe_leg(r, min(args.observe, 6), world)
        observe_configs[key] = {'envs': s['envs_per_gpu'], 'per_env_grids': cfg == 4,
                                'ms_per_observation_launch': ol['ms_per_observation_launch'], 'ms_per_step_plus_observation': ol['ms_per_step_plus_observation'],
                                'env_steps_per_s_with_observation': ol['env_steps_per_s_with_observation'], 'env_observations_per_s': ol['env_observations_per_s'],
                                'fp64_frac_algorithmic': ol['roofline']['frac'], 'executed_mfma_tflops': ol['roofline']['executed_mfma_tflops'],
                                'hbm_gbs_algorithmic': ol['roofline']['hbm_gbs_algorithmic'], 'window_observations': ol['window_observations'],
                                'live_env_fraction': ol['live_env_fraction']}
        configs[key]['closed_loop'] = observe_configs[key]
      tick(f'side leg {key}')
      del r
      torch.cuda.empty_cache()
      if cfg != 4:       # the same leg in the reference's own wind (noise generated in-kernel; the ten harmonics on the four waves)
        rg = make(cfg, n=size, steps=side_steps, warmup=32, noise_seed=20240917)
        sg = rg.summary(max(3, extra_reps // 2))
        configs[key]['env_steps_per_s_ground_truth_wind'] = sg['env_steps_per_s']
        configs[key]['ms_per_step_ground_truth_wind'] = sg['ms_per_step']
        del rg
        torch.cuda.empty_cache()
        tick(f'side leg {key} in the ground-truth wind')
    if rank == 0:
      try:
        configs['configs[0] counterpart: single-env facade'] = facade_leg()
      except Exception as e:            # the facade is not the measured product; never lose the line over it
        configs['configs[0] counterpart: single-env facade'] = {'error': repr(e)}
      tick('facade leg')

  if rank == 0:
    out = {
        'metric': 'env-steps/sec at 65 536 parallel envs; achieved HBM GB/s fraction of peak',
        'value': hs['env_steps_per_s'], 'unit': 'env-steps/s', 'n_gpus': n_joined, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': hs['ms_per_step'], 'higher_is_better': True, 'scaling': 'strong' if args.config == 3 else 'weak',
        'vs_baseline': None, 'dtype': 'f32+f64', 'data': 'synthetic',
        'dtype_note': 'state stored f32; vertical chain (p, T, V, n_air, thermal and ACS increments) computed in f64, solar geometry / wind blend in f32',
        'config': {'workload': PRESETS[args.config] + (' [per-env grids]' if args.per_env_grids and args.config != 4 else ''),
                   'preset': f'configs[{args.config}]', 'envs_per_gpu': n, 'global_envs': hs['global_envs'],
                   'substeps_per_step': args.substeps, 'live_env_fraction_end': hs['live_env_fraction_end'],
                   'wind': 'forecast (noise term 0: SURVEY 8(d))' if args.noise_seed is None else 'ground truth = forecast + in-kernel noise',
                   'ground_truth_wind': ground_truth,
                   'per_env_grids': bool(args.config == 4 or args.per_env_grids), 'decode_ms': hs['decode_ms'],
                   'parallelism': f'env-sharded x{world}, ' + ('no broadcast (per-rank decode)' if args.config == 4 else 'grid broadcast once') +
                                  (f', reward + terminal rows of every launch (<= {GATHER_EVERY} steps) gathered to rank 0 as ONE packed message on a side stream' if world > 1
                                   else ', single rank: nothing to exchange'),
                   # counted by the gatherer inside the timed regions (0 on one rank): never an exchange that did not run
                   'exchanges': {'gathers_per_timed_region': hs['gathers_per_region'], 'agent_step_rows_gathered_per_timed_region': hs['rows_gathered_per_region'],
                                 'launches_per_timed_region': hs['launches_per_region'],
                                 'bytes_per_rank_per_timed_region': 5 * n * hs['rows_gathered_per_region']}},
        'repetitions': {k: hs[k] for k in ('repetitions', 'steps_per_repetition', 'env_steps_per_s_min', 'env_steps_per_s_max',
                                            'env_steps_per_s_first_repetition', 'ms_per_step_min', 'ms_per_step_max')},
        # What bounds ble_step_kernel is VALU issue at one wave per SIMD, so that is `bound` / `frac` (measured in this run by
        # a nested rocprofv3 --pmc pass: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES).  The SURVEY 8(d) HBM figures stay as named
        # secondary fields: `hbm_formal` = algorithmic bytes / kernel time, `hbm_measured` = PMC traffic / kernel time.
        'roofline': {'bound': 'valu-issue', 'achieved': (issue or {}).get('valu_issue_frac'), 'peak': 1.0,
                     'unit': 'VALU-issuing cycles per wave cycle (one wave per SIMD)', 'frac': (issue or {}).get('valu_issue_frac'),
                     'frac_source': (issue or {}).get('source', 'measured in this run: nested rocprofv3 --pmc pass (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)'),
                     'traffic': traffic, 'traffic_detail': traffic_detail, 'traffic_note': traffic_note,
                     'hbm_formal': {'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                                    'algorithmic_bytes_per_env_step': ALGORITHMIC_BYTES_PER_ENV_STEP,
                                    'what': 'SURVEY 8(d): 280 B per env-step x live env-steps of a launch / its average duration'},
                     'hbm_measured': ({'achieved': traffic / (hs['kernel_ms_mean'] * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                       'frac': traffic / (hs['kernel_ms_mean'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       'ratio_to_algorithmic': traffic / bytes_per_launch} if traffic else None),
                     'kernel': 'ble_step_kernel', 'kernel_ms': hs['kernel_ms_mean'], 'kernel_ms_median': hs['kernel_ms_median'], 'kernel_ms_min': hs['kernel_ms_min'],
                     'kernel_event_repetitions': hs['kernel_event_repetitions'],
                     'agent_steps_per_launch': args.steps / -(-args.steps // GATHER_EVERY),
                     'kernel_us_per_agent_step': 1e3 * hs['kernel_ms_mean'] * (-(-args.steps // GATHER_EVERY)) / args.steps,
                     'note': 'the measured HBM traffic (`traffic`, bytes per launch) is a few % of the algorithmic bytes because the state stays '
                             f'in registers for the {args.steps / -(-args.steps // GATHER_EVERY):g} agent steps of a launch and the grid gather is served by L2: '
                             'nothing is re-read, HBM is not the bound (DESIGN.md 3)',
                     'valu_issue_frac': (issue or {}).get('valu_issue_frac'),
                     'wave_issue_utilisation': (issue or {}).get('wave_issue_utilisation'),
                     'instruction_issue': issue},
        'per_rank': hs['per_rank'],
        'configs': configs,
    }
    if policy is not None:
      out['policy_in_the_loop'] = policy
    if observe is not None:
      out['observe'] = observe
    if observe_configs:
      out['observe_configs'] = observe_configs
    if world == 1 and not args.no_cpu_baseline and not args.no_extras:
      acts = np.random.default_rng(7).integers(0, 3, (64, n)).astype(np.uint8)
      out['cpu_baseline'] = cpu_baseline(head_initial_host_state, list(acts), field)     # the SAME initial states the GPU flew
      tick('cpu baseline')
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
