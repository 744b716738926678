"""bench.py's launch paths on the GPU box (1 GPU): the self-spawned N > 1 path with its exchanges inside the timed region."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=600):
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device; none visible')
  env = dict(os.environ, **(env_extra or {}))
  env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                     timeout=timeout)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout[-2000:]
  return json.loads(lines[0])


def test_bench_gpus_2_without_a_launcher_spawns_two_ranks_and_gathers_the_partial_block():
  """The shape of the driver's command with N = 2 and no torchrun: bench.py starts both ranks itself (gloo here: two ranks
  share the box's one GPU; RCCL refuses that), counts them, and the rows of every launch -- sharded runs launch 8 steps at a
  time so that each exchange overlaps the next launch: 8 + 8 + 4 here -- are gathered inside the region."""
  out = _bench(['--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3', '--no-extras'], {'BLE_DIST_BACKEND': 'gloo'})
  assert out['n_gpus'] == 2 and out['steps'] == 20 and out['warmup'] == 5
  assert out['config']['global_envs'] == 2 * 65536 and out['config']['envs_per_gpu'] == 65536
  ex = out['config']['exchanges']
  assert ex['gathers_per_timed_region'] == 3 and ex['launches_per_timed_region'] == 3 and ex['agent_step_rows_gathered_per_timed_region'] == 20
  assert ex['bytes_per_rank_per_timed_region'] == 5 * 65536 * 20
  assert out['value'] > 1e6 and out['scaling'] == 'weak'
  # configs[3]: 65 536 environments GLOBAL, 32 768 per rank
  out3 = _bench(['--gpus', '2', '--steps', '40', '--warmup', '5', '--reps', '3', '--no-extras', '--config', '3'], {'BLE_DIST_BACKEND': 'gloo'})
  assert out3['n_gpus'] == 2 and out3['config']['global_envs'] == 65536 and out3['config']['envs_per_gpu'] == 32768
  assert out3['config']['exchanges']['gathers_per_timed_region'] == 5 and out3['config']['exchanges']['agent_step_rows_gathered_per_timed_region'] == 40
  assert out3['scaling'] == 'strong'
  # configs[4]: every rank decodes its own per-environment forecasts (no broadcast); 8 192 per rank here instead of 32 768
  # (two ranks share one GPU: 2 x 10.4 GB of grids would fit, the decode time would not be worth it in a test)
  out4 = _bench(['--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3', '--no-extras', '--config', '4', '--envs-per-gpu', '8192'],
                {'BLE_DIST_BACKEND': 'gloo'})
  assert out4['n_gpus'] == 2 and out4['config']['per_env_grids'] and out4['config']['global_envs'] == 2 * 8192
  assert 'no broadcast' in out4['config']['parallelism'] and out4['config']['exchanges']['gathers_per_timed_region'] == 3
  assert out4['config']['decode_ms'] > 0
  # the exchange cadence is a knob: one 20-step launch and one exchange per region
  out1 = _bench(['--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3', '--no-extras'], {'BLE_DIST_BACKEND': 'gloo', 'BLE_STEPS_PER_GATHER': '32'})
  assert out1['config']['exchanges']['gathers_per_timed_region'] == 1 and out1['config']['exchanges']['agent_step_rows_gathered_per_timed_region'] == 20


def test_bench_gpus_2_default_legs_run_sharded():
  """What the driver's N > 1 command runs beyond the headline (no --no-extras): the ground-truth-wind leg, the observation
  leg with its three consumers of the observation blocks (gather to the learner rank, all_to_all for a data-parallel
  learner, rank-local), and the configs[3] / configs[4] legs -- two ranks on the box's one GPU over gloo, a small headline
  batch so that the test stays short."""
  out = _bench(['--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3', '--observe', '2', '--envs-per-gpu', '2048'],
               {'BLE_DIST_BACKEND': 'gloo', 'BLE_BENCH_SIDE_ENVS': '4096', 'BLE_BENCH_SIDE_STEPS': '32'}, timeout=900)
  # (side legs at 4 096 environments per rank and 32-step regions: the code path, not the rate -- over gloo every one of a 192-step
  #  region's 24 gathers crosses the host, 0.1 s each: 57 s of this test's 60 were that)
  assert out['n_gpus'] == 2 and out['config']['envs_per_gpu'] == 2048 and out['config']['global_envs'] == 4096
  gt = out['config']['ground_truth_wind']
  assert gt is not None and gt['env_steps_per_s'] > 1e4 and gt['global_envs'] == 4096     # (gloo moves the gathers through the host)
  assert out['roofline']['traffic'] is None and 'N > 1' in out['roofline']['traffic_note']
  modes = out['observe']['exchange_modes']
  assert set(modes) == {'gather', 'all_to_all', 'local'}
  block = 2048 * 1099 * 4
  assert modes['gather']['bytes_per_link'] == block and modes['gather']['ms_per_step_plus_observation'] > 0
  assert modes['all_to_all']['bytes_per_link'] == block // 2 and modes['local']['bytes_per_link'] == 0
  assert modes['local'].get('ms_per_step_plus_observation', 0) > 0          # no exchange: cannot fail on any backend
  assert 'ms_per_step_plus_observation' in modes['all_to_all'] or 'error' in modes['all_to_all']
  keys = ' '.join(out['configs'])
  assert 'configs[3]' in keys and 'configs[4]' in keys
  for k, v in out['configs'].items():
    if k.startswith('configs[3]') or k.startswith('configs[4]'):
      assert v['env_steps_per_s'] > 1e5, k
  assert 'cpu_baseline' not in out                                            # rank 0 at N = 1 only


def test_bench_gpus_8_every_leg_with_eight_ranks():
  """The driver's 8-GPU command shape with EIGHT ranks -- the node size the scaling run uses -- on the box's one GPU over gloo, small
  batches: the rank / offset arithmetic (contiguous shards of the global index range), the eight-way gathers and all_to_all, the per-rank
  decomposition and every default leg run with world = 8, not only with 2."""
  out = _bench(['--gpus', '8', '--steps', '20', '--warmup', '5', '--reps', '3', '--observe', '2', '--envs-per-gpu', '1024'],
               {'BLE_DIST_BACKEND': 'gloo', 'BLE_BENCH_SIDE_ENVS': '1024', 'BLE_BENCH_SIDE_STEPS': '32', 'OMP_NUM_THREADS': '2'}, timeout=900)
  assert out['n_gpus'] == 8 and out['config']['envs_per_gpu'] == 1024 and out['config']['global_envs'] == 8192 and out['scaling'] == 'weak'
  ex = out['config']['exchanges']
  assert ex['gathers_per_timed_region'] == 3 and ex['agent_step_rows_gathered_per_timed_region'] == 20
  assert len(out['per_rank']['kernel_us_per_timed_region']) == 8 and min(out['per_rank']['kernel_us_per_timed_region']) > 0
  modes = out['observe']['exchange_modes']
  assert set(modes) == {'gather', 'all_to_all', 'local'} and modes['all_to_all']['bytes_per_link'] == 1024 * 1099 * 4 // 8
  keys = ' '.join(out['configs'])
  assert 'configs[3]' in keys and 'configs[4]' in keys and out['config']['ground_truth_wind']['global_envs'] == 8192
  for k, v in out['configs'].items():
    if k.startswith('configs[3]') or k.startswith('configs[4]'):
      assert v['global_envs'] == 8 * 1024 and v['env_steps_per_s'] > 1e5, k


def test_bench_under_the_drivers_launcher_command():
  """The driver's own N > 1 command line, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
  --master-port P bench.py --gpus N --steps K --warmup W` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's environment), N = 2
  on the box's one GPU over gloo."""
  import socket
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device; none visible')
  sock = socket.socket(); sock.bind(('127.0.0.1', 0)); port = sock.getsockname()[1]; sock.close()
  env = dict(os.environ, BLE_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
    env.pop(k, None)
  r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                      '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3',
                      '--no-extras', '--envs-per-gpu', '4096'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line, from rank 0
  out = json.loads(lines[0])
  assert out['n_gpus'] == 2 and out['steps'] == 20 and out['warmup'] == 5 and out['config']['global_envs'] == 8192
  assert out['value'] > 1e6 and out['config']['exchanges']['agent_step_rows_gathered_per_timed_region'] == 20


def test_bench_refuses_a_world_that_is_not_gpus():
  env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-extras'],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
  assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in r.stderr


def test_rccl_collectives_of_the_sharded_path_on_one_rank():
  """The exchanges of DESIGN section 7 through the real backend ("nccl" = RCCL) as far as one GPU allows: a single-rank
  process group runs the grid broadcast, the packed reward / terminal gather on its side stream (dist.gather itself, not
  the world == 1 shortcut), the observation gather and its all_to_all variant, the MAX / SUM all-reduces of the timing and the barrier -- RCCL loads,
  builds its communicator and moves the bytes.  (Several ranks on one GPU are refused by RCCL: the world-2 runs use gloo.)"""
  code = r'''
import os, socket, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from balloon_learning_environment_amd import distributed as bdist, vec_state
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
assert dist.get_backend() == 'nccl'
grid = torch.randn(vec_state.GRID_SHAPE, device=dev); ref = grid.clone()
dist.broadcast(grid, src=0); assert torch.equal(grid, ref)
n, k = 8192, 20
buf, r, t = bdist.packed_output_block(k, n, dev)
r.copy_(torch.rand(k, n, device=dev)); t.copy_((torch.rand(k, n, device=dev) < 0.01).to(torch.uint8))
out = torch.zeros(1, 5 * k * n, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
  dist.gather(buf, [out[0]], dst=0)
torch.cuda.current_stream(dev).wait_stream(side)
assert torch.equal(out[0], buf)
obs = torch.randn(n, 1099, device=dev); got = torch.zeros(1, n, 1099, device=dev)
dist.gather(obs, [got[0]], dst=0); assert torch.equal(got[0], obs)
a2a = torch.zeros(n, 1099, device=dev)
dist.all_to_all_single(a2a, obs); assert torch.equal(a2a, obs)        # the data-parallel learner's exchange (ObservationGatherer mode 'all_to_all')
v = torch.tensor([3.25], dtype=torch.float64, device=dev)
dist.all_reduce(v, op=dist.ReduceOp.MAX); dist.all_reduce(v, op=dist.ReduceOp.SUM); assert float(v.item()) == 3.25
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print('rccl ok')
'''
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
  assert r.returncode == 0 and 'rccl ok' in r.stdout, r.stderr[-2000:]
