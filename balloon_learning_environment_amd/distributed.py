"""Multi-GPU sharding: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Environments are independent (no cross-env term anywhere in the transition), so the env
index range is partitioned contiguously across ranks and the per-step data path needs no
collective.  Exactly two exchanges exist:
  * the decoded wind grid (317 520 B) is broadcast from rank 0 once per episode batch;
  * rewards / terminals are gathered back to rank 0 (point-to-point `gather`) every
    `gather_every` agent steps, on a side stream so that it overlaps the next steps.
The helpers are device-agnostic so that the N>1 logic is covered by gloo tests on CPU.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_global: int, rank: int, world: int) -> Tuple[int, int]:
  """Contiguous [lo, hi) slice of the env index range owned by `rank` (sizes differ by <= 1)."""
  base, rem = divmod(n_global, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


# BASELINE.json configs[i] -> (environments of the whole job, per-env forecasts?) as a function of the world size
def preset_layout(config: int, rank: int, world: int) -> dict:
  """How preset `config` (= BASELINE.json configs[config]) lays out on `world` ranks:
     1  4 096 envs on one GPU, shared grid         2  65 536 envs PER GPU, shared grid (weak scaling)
     3  65 536 envs GLOBAL, contiguous shards      4  32 768 envs per GPU with per-env decoded grids, no broadcast
  Returns global_envs, this rank's [lo, hi) slice of the global env index range, per_env_grids, broadcast."""
  if config == 1:
    n_global, per_env = 4096 * world, False
  elif config == 2:
    n_global, per_env = 65536 * world, False
  elif config == 3:
    n_global, per_env = 65536, False
  elif config == 4:
    n_global, per_env = 32768 * world, True
  else:
    raise ValueError(f'unknown preset {config}')
  lo, hi = shard_range(n_global, rank, world)
  return {'global_envs': n_global, 'lo': lo, 'hi': hi, 'n_local': hi - lo, 'per_env_grids': per_env,
          'broadcast_grid': not per_env, 'grid_bytes_per_rank': (hi - lo) * 317520 if per_env else 317520,
          'gather_bytes_per_step': (hi - lo) * 5}


def broadcast_grid(grid: torch.Tensor, src: int = 0) -> torch.Tensor:
  """Rank `src` owns the wind grid (21,21,10,9,2) float32; everyone gets a copy in place."""
  if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
    dist.broadcast(grid, src=src)
  return grid


class OutputGatherer:
  """Collects [K, n_local] reward / terminal blocks of every rank on rank `dst`.

  `dist.gather` (point-to-point sends over the direct xGMI links; 5 B per env-step per
  sender), not all_gather: a ring all_gather would push world x the volume through every
  single link, and only the learner rank consumes the data.  Issued on a side stream after
  waiting for the producer so that it overlaps the following steps.
  """

  def __init__(self, k: int, n_local: int, device, world: Optional[int] = None, dst: int = 0, slots: int = 1):
    """slots: receive slots of gather_packed on rank `dst`, used round-robin.  A region of L launches issues its L gathers
    back to back on the side stream; with slots >= L every launch's rows are still there when wait() returns (one slot
    would keep only the last launch's)."""
    initialized = dist.is_available() and dist.is_initialized()
    self.world = world if world is not None else (dist.get_world_size() if initialized else 1)
    self.rank = dist.get_rank() if initialized else 0
    self.dst = dst
    self.is_dst = self.rank == dst
    self.k, self.n_local, self.device = k, n_local, device
    rows = self.world if self.is_dst else 0
    # receive buffers on rank dst ([world, k, n_local]; empty elsewhere): reward / terminal for gather(), `packed` for gather_packed()
    self.reward = torch.zeros((rows, k, n_local), dtype=torch.float32, device=device)
    self.terminal = torch.zeros((rows, k, n_local), dtype=torch.uint8, device=device)
    self.stream = torch.cuda.Stream(device=device) if torch.device(device).type == 'cuda' else None
    self.slots = max(1, int(slots))
    # a rank's row of a slot starts 16-byte aligned whatever k and n_local are (unpack() views it as float32)
    self.row_bytes = -(-5 * k * n_local // 16) * 16
    self.packed = None          # [slots, world, row_bytes] uint8 on rank dst: the packed blocks of gather_packed
    self.packed_gathers = 0     # gather_packed calls so far
    self._region_gathers = 0    # ... since the last wait(): launch j of a region goes to slot j % slots
    self._last_slot = 0
    self.gathers = 0            # exchanges issued by this rank
    self.rows_gathered = 0      # agent steps they carried

  def _gather(self, reward_block, terminal_block, c):
    if self.is_dst:
      dist.gather(reward_block, [self.reward[r][:c] for r in range(self.world)], dst=self.dst)
      dist.gather(terminal_block, [self.terminal[r][:c] for r in range(self.world)], dst=self.dst)
    else:
      dist.gather(reward_block, None, dst=self.dst)
      dist.gather(terminal_block, None, dst=self.dst)

  def gather(self, reward_block: torch.Tensor, terminal_block: torch.Tensor) -> None:
    """`reward_block`, `terminal_block`: [c, n_local] with c <= k (a region's last launch may hold fewer steps than
    the others: its rows are gathered too); rank `dst` finds them in reward[r][:c], terminal[r][:c]."""
    c = int(reward_block.shape[0])
    assert 0 < c <= self.k and tuple(terminal_block.shape) == tuple(reward_block.shape)
    self.gathers += 1; self.rows_gathered += c
    if self.world == 1:
      self.reward[0][:c].copy_(reward_block); self.terminal[0][:c].copy_(terminal_block)
      return
    if self.stream is not None:
      self.stream.wait_stream(torch.cuda.current_stream(reward_block.device))
      with torch.cuda.stream(self.stream):
        self._gather(reward_block, terminal_block, c)
        reward_block.record_stream(self.stream); terminal_block.record_stream(self.stream)
    else:
      self._gather(reward_block, terminal_block, c)

  def gather_packed(self, block: torch.Tensor, c: int) -> int:
    """ONE exchange per launch instead of two: `block` is the launch's packed output (packed_output_block: the [c, n_local]
    float32 rewards followed by the [c, n_local] uint8 terminals, 5 c n_local bytes); rank `dst` finds rank r's copy in
    slot (call number % slots) of `packed` (unpack(r, c, slot)).  Fewer, larger messages: every point-to-point gather costs
    its latency once.  Returns the slot."""
    nbytes = int(block.numel())
    assert block.dtype == torch.uint8 and block.is_contiguous() and nbytes == 5 * int(c) * self.n_local and 0 < int(c) <= self.k
    if self.packed is None:
      rows = self.world if self.is_dst else 0
      self.packed = torch.zeros((self.slots, rows, self.row_bytes), dtype=torch.uint8, device=block.device)
    # launch j of a REGION lands in slot j: the index restarts when the region's wait() returns (it used to run on across
    # regions, so a warm-up region of another length shifted the mapping -- ADVICE r4)
    slot = self._region_gathers % self.slots
    self._region_gathers += 1
    self._last_slot = slot
    self.packed_gathers += 1
    self.gathers += 1; self.rows_gathered += int(c)
    if self.world == 1:
      self.packed[slot][0][:nbytes].copy_(block)
      return slot

    def go():
      dist.gather(block, [self.packed[slot][r][:nbytes] for r in range(self.world)] if self.is_dst else None, dst=self.dst)
    if self.stream is not None:
      self.stream.wait_stream(torch.cuda.current_stream(block.device))
      with torch.cuda.stream(self.stream):
        go()
        block.record_stream(self.stream)
    else:
      go()
    return slot

  def unpack(self, r: int, c: int, slot: Optional[int] = None):
    """(reward [c, n_local] float32, terminal [c, n_local] uint8) views of rank r's packed block in `slot` (default: the
    slot of the most recent gather_packed); rank `dst` only, after wait()."""
    n = self.n_local
    if slot is None:
      slot = self._last_slot
    row = self.packed[slot][r]
    return row[:4 * c * n].view(torch.float32).view(c, n), row[4 * c * n:5 * c * n].view(c, n)

  def wait(self) -> None:
    """End of a region: the compute stream waits for the exchanges; the next gather_packed goes to slot 0 again."""
    self._region_gathers = 0
    if self.stream is not None:
      torch.cuda.current_stream(self.reward.device).wait_stream(self.stream)


def packed_output_block(c: int, n_local: int, device):
  """The outputs of one launch of c agent steps in ONE buffer, so that they travel in one message: returns
  (buffer uint8 [5 c n], reward view float32 [c, n], terminal view uint8 [c, n])."""
  buf = torch.zeros(5 * c * n_local, dtype=torch.uint8, device=device)
  return buf, buf[:4 * c * n_local].view(torch.float32).view(c, n_local), buf[4 * c * n_local:].view(c, n_local)


def run_region(launches, gatherer: Optional['OutputGatherer'], on_compute_enqueued=None) -> None:
  """One region of a sharded rollout: every launch (a callable that enqueues <= k agent steps of this rank's shard and
  fills its [c, n_local] reward / terminal blocks) is followed by the gather of exactly those blocks to the learner
  rank -- the last, shorter launch of a region included -- and the region ends when the exchanges have been waited for.
  `launches`: iterable of (launch, reward_block, terminal_block) or, for one exchange per launch, (launch, packed buffer,
  reward view, terminal view) from packed_output_block.  bench.py's timed region and the gloo tests run this."""
  for item in launches:
    item[0]()
    if gatherer is not None:
      if len(item) == 3:
        gatherer.gather(item[1], item[2])                 # (launch, reward rows, terminal rows): two exchanges
      else:
        gatherer.gather_packed(item[1], item[2].shape[0])  # (launch, packed buffer, reward view, terminal view): one
  if on_compute_enqueued is not None:
    on_compute_enqueued()        # (bench.py records an event here: what follows on the compute stream is exposed exchange time)
  if gatherer is not None:
    gatherer.wait()


XGMI_LINK_GBS = 153.0      # one direction of one xGMI link between two GPUs of a node (MI355X_MICROARCH.md); 7 links per GPU
OBSERVATION_MODES = ('gather', 'all_to_all', 'local')


def observation_exchange_model(mode: str, n_local: int, obs_dim: int, world: int) -> dict:
  """Bytes and the link-bound time of one observation exchange (arithmetic, not a measurement: DESIGN.md section 7).
  The GPUs of a node are fully connected by point-to-point xGMI links, so transfers to / from DIFFERENT peers run in parallel:
    gather      every rank sends its whole block to the learner rank: each of the learner's world - 1 links carries one block
                (n_local x obs_dim x 4 B = 288 MB at 65 536 environments) -- the links work in parallel, the exchange takes one
                block's time (1.9 ms); the learner ingests (world - 1) blocks (2.0 GB at 8 GPUs: 0.25 ms of its HBM).
    all_to_all  every rank keeps 1 / world of its block and sends 1 / world to each peer (a data-parallel learner: rank r
                consumes rows r of every block): each link carries block / world (36 MB: 0.24 ms at 8 GPUs).
    local       the consumer of a rank's observations lives on that rank (a policy replica per GPU): nothing moves."""
  block = n_local * obs_dim * 4
  if world <= 1 or mode == 'local':
    per_link, sent, received = 0, 0, 0
  elif mode == 'gather':
    per_link, sent, received = block, block, (world - 1) * block       # (sent: by every rank but the learner; received: by the learner)
  elif mode == 'all_to_all':
    per_link = block // world
    sent = received = (world - 1) * per_link
  else:
    raise ValueError(mode)
  return {'mode': mode, 'block_bytes_per_rank': block, 'bytes_per_link': per_link, 'bytes_sent_per_rank': sent,
          'bytes_received_max_per_rank': received, 'expected_ms_link_bound': 1e3 * per_link / (XGMI_LINK_GBS * 1e9),
          'link_gbs_assumed': XGMI_LINK_GBS}


class ObservationGatherer:
  """Hands the [n_local, 1099] observation block of every rank to whoever consumes it, on a side stream like
  OutputGatherer (4 396 B per env; 288 MB per block for 65 536 envs per GPU).  Three consumers (`mode`):
    'gather'      one learner on rank `dst`: a point-to-point `gather`; rank dst finds rank r's block in obs[r]
    'all_to_all'  a data-parallel learner: rank r receives rows [r n_local / world, (r + 1) n_local / world) of EVERY
                  rank's block (`all_to_all_single`); it finds the slice that came from rank q in obs[q]
                  ([world, n_local / world, obs_dim] on every rank); 1 / world of the gather's bytes per link
    'local'       the consumer is on the producing rank (a policy replica per GPU): no exchange; obs[0] IS the block
  observation_exchange_model() gives the bytes and the link-bound time of each.  Double-buffered on the producer side: the
  kernel of step t + 1 may overwrite its output while step t is in flight."""

  def __init__(self, n_local: int, obs_dim: int, device, world: Optional[int] = None, dst: int = 0, mode: str = 'gather'):
    initialized = dist.is_available() and dist.is_initialized()
    self.world = world if world is not None else (dist.get_world_size() if initialized else 1)
    self.rank = dist.get_rank() if initialized else 0
    self.dst, self.is_dst = dst, self.rank == dst
    assert mode in OBSERVATION_MODES, mode
    self.mode, self.n_local, self.obs_dim = mode, n_local, obs_dim
    if mode == 'gather':
      rows = self.world if self.is_dst else 0
      self.obs = torch.zeros((rows, n_local, obs_dim), dtype=torch.float32, device=device)
    elif mode == 'all_to_all':
      assert n_local % self.world == 0, 'all_to_all re-partitions the block into world equal row slices'
      self.obs = torch.zeros((self.world, n_local // self.world, obs_dim), dtype=torch.float32, device=device)
    else:
      self.obs = None              # gather() points it at the producer's own block
    self.stream = torch.cuda.Stream(device=device) if torch.device(device).type == 'cuda' else None
    self.exchanges = 0
    self.model = observation_exchange_model(mode, n_local, obs_dim, self.world)

  def gather(self, block: torch.Tensor) -> None:
    self.exchanges += 1
    if self.mode == 'local':
      self.obs = block[None]
      return
    if self.world == 1:
      self.obs.view(-1, self.obs_dim).copy_(block)
      return

    def go():
      if self.mode == 'gather':
        dist.gather(block, [self.obs[r] for r in range(self.world)] if self.is_dst else None, dst=self.dst)
      else:
        dist.all_to_all_single(self.obs.view(-1, self.obs_dim), block)
    if self.stream is not None:
      self.stream.wait_stream(torch.cuda.current_stream(block.device))
      with torch.cuda.stream(self.stream):
        go()
        block.record_stream(self.stream)
    else:
      go()

  def wait(self) -> None:
    if self.stream is not None and self.mode != 'local':
      torch.cuda.current_stream(self.obs.device).wait_stream(self.stream)


def max_over_ranks(value: float, device) -> float:
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return value
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def sum_over_ranks(value: float, device) -> float:
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return value
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.SUM)
  return float(t.item())


def joined_ranks(device) -> int:
  """How many ranks actually take part in this job (an all-reduce of ones; 1 without a process group)."""
  return int(round(sum_over_ranks(1.0, device)))


def spawn_local_ranks(argv: List[str], n: int, env_extra: Optional[dict] = None, timeout: Optional[float] = None) -> int:
  """Runs `argv` as n processes of ONE node, one rank per process, with the environment torch.distributed.run would
  set (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port).  Rank 0
  inherits stdout (it prints the result line); the other ranks' stdout goes to stderr.  Returns 0 if every rank
  exited with 0; if one fails the others are terminated (by PID) and its code is returned."""
  import os
  import socket
  import subprocess
  import sys
  import time
  with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
  procs = []
  for rank in range(n):
    env = dict(os.environ)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if env_extra:
      env.update(env_extra)
    procs.append(subprocess.Popen(argv, env=env, stdout=None if rank == 0 else sys.stderr))
  deadline = None if timeout is None else time.monotonic() + timeout
  code = 0
  pending = list(procs)
  while pending:
    for p in list(pending):
      rc = p.poll()
      if rc is not None:
        pending.remove(p)
        if rc != 0 and code == 0:
          code = rc
    if code != 0 or (deadline is not None and time.monotonic() > deadline):
      for p in pending:
        p.terminate()
      for p in pending:
        try:
          p.wait(10)
        except subprocess.TimeoutExpired:
          p.kill()
      return code if code != 0 else -9
    time.sleep(0.05)
  return code
