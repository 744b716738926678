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
  share the box's one GPU; RCCL refuses that), counts them, and the 20-step launch's rows are gathered inside the region."""
  out = _bench(['--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3', '--no-extras'], {'BLE_DIST_BACKEND': 'gloo'})
  assert out['n_gpus'] == 2 and out['steps'] == 20 and out['warmup'] == 5
  assert out['config']['global_envs'] == 2 * 65536 and out['config']['envs_per_gpu'] == 65536
  ex = out['config']['exchanges']
  assert ex['gathers_per_timed_region'] == 1 and ex['agent_step_rows_gathered_per_timed_region'] == 20
  assert ex['bytes_per_rank_per_timed_region'] == 5 * 65536 * 20
  assert out['value'] > 1e6 and out['scaling'] == 'weak'
  # configs[3]: 65 536 environments GLOBAL, 32 768 per rank
  out3 = _bench(['--gpus', '2', '--steps', '40', '--warmup', '5', '--reps', '3', '--no-extras', '--config', '3'], {'BLE_DIST_BACKEND': 'gloo'})
  assert out3['n_gpus'] == 2 and out3['config']['global_envs'] == 65536 and out3['config']['envs_per_gpu'] == 32768
  assert out3['config']['exchanges']['gathers_per_timed_region'] == 2 and out3['scaling'] == 'strong'
  # configs[4]: every rank decodes its own per-environment forecasts (no broadcast); 8 192 per rank here instead of 32 768
  # (two ranks share one GPU: 2 x 10.4 GB of grids would fit, the decode time would not be worth it in a test)
  out4 = _bench(['--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3', '--no-extras', '--config', '4', '--envs-per-gpu', '8192'],
                {'BLE_DIST_BACKEND': 'gloo'})
  assert out4['n_gpus'] == 2 and out4['config']['per_env_grids'] and out4['config']['global_envs'] == 2 * 8192
  assert 'no broadcast' in out4['config']['parallelism'] and out4['config']['exchanges']['gathers_per_timed_region'] == 1
  assert out4['config']['decode_ms'] > 0


def test_bench_refuses_a_world_that_is_not_gpus():
  env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-extras'],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
  assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in r.stderr
