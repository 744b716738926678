"""N>1 logic on CPU: world_size-2 gloo processes exercise the sharding, the grid broadcast
and the reward/terminal gather that bench.py uses over RCCL on GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from balloon_learning_environment_amd import distributed as bdist


def test_shard_range_partitions_exactly():
  for n in (0, 1, 7, 64, 65536, 262144, 1000003):
    for world in (1, 2, 3, 8):
      spans = [bdist.shard_range(n, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      for a, b in zip(spans, spans[1:]):
        assert a[1] == b[0]
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    # grid broadcast: only rank 0 has the field
    grid = torch.zeros((21, 21, 10, 9, 2), dtype=torch.float32)
    if rank == 0:
      grid.copy_(torch.from_numpy((np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)))
    bdist.broadcast_grid(grid, src=0)
    # per-rank outputs of K steps for the rank's shard of 2 x 6 envs
    k, n_global = 4, 12
    lo, hi = bdist.shard_range(n_global, rank, world)
    env_ids = torch.arange(lo, hi, dtype=torch.float32)
    reward = torch.stack([env_ids * 10 + s for s in range(k)])            # [k, n_local]
    terminal = (reward.to(torch.int64) % 3 == 0).to(torch.uint8)
    g = bdist.OutputGatherer(k, hi - lo, 'cpu', world)
    g.gather(reward.contiguous(), terminal.contiguous())
    g.wait()
    mx = bdist.max_over_ranks(float(rank + 1), 'cpu'); sm = bdist.sum_over_ranks(float(rank + 1), 'cpu')
    # observation blocks [n_local, 1099] to rank 0
    og = bdist.ObservationGatherer(hi - lo, 1099, 'cpu', world)
    og.gather((env_ids[:, None] + torch.arange(1099, dtype=torch.float32)[None, :] / 2048).contiguous())
    og.wait()
    torch.save(dict(grid=grid, reward=g.reward, terminal=g.terminal, mx=mx, sm=sm, obs=og.obs), os.path.join(out_dir, f'r{rank}.pt'))
  finally:
    dist.destroy_process_group()


def test_broadcast_and_gather_world2(tmp_path):
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  outs = [torch.load(os.path.join(tmp_path, f'r{r}.pt')) for r in range(world)]
  ref_grid = torch.from_numpy((np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32))
  for o in outs:
    assert torch.equal(o['grid'], ref_grid)               # every rank holds rank 0's field
    assert o['mx'] == 2.0 and o['sm'] == 3.0
  # rank 0 holds [world, k, n_local]: it reassembles the global [k, n_global] block in env order
  o = outs[0]
  glob = torch.cat([o['reward'][r] for r in range(world)], dim=1)
  expect = torch.stack([torch.arange(12, dtype=torch.float32) * 10 + s for s in range(4)])
  assert torch.equal(glob, expect)
  tglob = torch.cat([o['terminal'][r] for r in range(world)], dim=1)
  assert torch.equal(tglob, (expect.to(torch.int64) % 3 == 0).to(torch.uint8))
  assert outs[1]['reward'].numel() == 0                   # senders keep nothing
  obs = torch.cat([o['obs'][r] for r in range(world)], dim=0)
  assert obs.shape == (12, 1099)
  assert torch.equal(obs, torch.arange(12, dtype=torch.float32)[:, None] + torch.arange(1099, dtype=torch.float32)[None, :] / 2048)
  assert outs[1]['obs'].numel() == 0                      # only the destination rank holds the gathered blocks
