"""BASELINE.json configs[0] in closed loop on the device (fixture F13).

The reference's StationSeekerAgent flew the reference's arena loop for micro_eval's 960 steps
(tests/golden/make_golden.py::f13_station_seeker_episode).  Here the same episode is driven through
the C ABI (ble_observe_f32 + ble_step_f32, one environment):

 * teacher-forced: at every step the fixture's state is written to the device, the DEVICE
   observation is handed to the restated agent (oracle/station_seeker_oracle.py, pinned to the
   reference agent on all 960 steps) and its action must equal the reference agent's on EVERY
   step; the observation is compared with the feature oracle on the same float32 inputs (1e-5 on every
   entry) and with the reference's 1099-vector (1e-5 + the reference's own sensitivity to the float32
   rounding of its float64 states, computed here), the transition with the reference's next state
   (1e-5, discrete exact);
 * free-running: the device flies its own closed loop (its observation -> agent -> its transition)
   from the fixture's initial state; the first step at which an action differs from the
   reference's is reported (a closed loop amplifies 1e-7 observation differences through argmax
   ties -- the best level's margin over the runner-up is < 1e-6 on several steps) and the episode
   must stay a valid 960-step station-seeking flight.
"""
import numpy as np
import pytest
import torch

import helpers
import oracle
import station_seeker_oracle as sso
from helpers import FLOORS, STATE_FLOATS, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vec_state():
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device; none visible')
  from balloon_learning_environment_amd import vec_state
  return vec_state


def _arrays(row):
  return {k: np.array([v]) for k, v in row.items()}


def _oracle_vector(job):
  """One 1099-vector of the feature oracle from a snapshot of its history (a 120 x 120 GP fit in NumPy: 66 ms) -- the snapshots of an
  episode are independent of each other, so a process pool forms them side by side."""
  import ctypes
  import features_oracle
  try:      # the C oracle's OpenMP regions would start one thread per hardware thread in every worker
    ctypes.CDLL('libgomp.so.1').omp_set_num_threads(1)
  except OSError:
    pass
  field, alpha, locs, errs, row = job
  fo = features_oracle.FeatureOracle(field, alpha)
  fo.locs, fo.errs, fo.row = locs, errs, row
  return fo.features().astype(np.float64)


def test_station_seeker_episode_teacher_forced(vec_state):
  g = helpers.golden('f13_station_seeker')
  field = helpers.fixture_field(g)
  n = int(g['n_flown'])
  sim = vec_state.VecSimulator(1)
  sim.set_grid(torch.from_numpy(field).cuda())
  import features_oracle
  fo = features_oracle.FeatureOracle(field, float(np.float32(g['alpha'][0])))
  worst_obs = 0.0; worst_ref = 0.0; worst_sens = 0.0; worst_state = 0.0; compared = 0
  jobs, kept = [], []
  for i in range(n):
    row = helpers.feature_row(g, 0, i)
    sim.set_state(_arrays(row))
    noise = torch.from_numpy(g['noise_uv'][0, i:i + 1].astype(np.float32)).cuda()
    obs = sim.observe(noise).cpu().numpy()[0]
    sim.check_errors()
    # the agent's decision from the DEVICE observation == the reference agent's from the reference's
    assert sso.pick_action(obs) == g['actions'][0, i], f'step {i}: level {sso.best_level(obs)} vs {g["levels"][i]}'
    want = g['features'][0, i]
    unreachable = lambda f: (f[16::3] == 0) & (f[17::3] == 1) & (f[18::3] == 1)
    np.testing.assert_array_equal(unreachable(obs), unreachable(want), err_msg=f'step {i}')
    np.testing.assert_array_equal(obs[8:14], want[8:14], err_msg=f'step {i}')
    # the oracle on the device's own inputs (float32 state and noise): every entry within 1e-5.  The oracle sees every
    # observation; its 1099-vector is formed on the first 126 steps -- the window fills and starts to slide --, then on every 12th
    # step, and on the last 16: snapshots of its history now, the vectors from a process pool after the flight.
    row32 = {k: (float(np.float32(v)) if isinstance(v, float) else v) for k, v in row.items()}
    fo.observe(row32, g['noise_uv'][0, i].astype(np.float32).astype(np.float64))
    if i < 126 or i % 12 == 0 or i >= n - 16:
      jobs.append((field, fo.alpha, [list(v) for v in fo.locs[-130:]], [list(v) for v in fo.errs[-130:]], row32))      # (the 6 h window holds <= 120)
      kept.append((i, obs.copy(), want.copy()))
    # the transition with the agent's action and the ground-truth wind
    act = torch.tensor([g['actions'][0, i]], dtype=torch.uint8).cuda()
    reward, terminal = sim.step(act, noise)
    got = sim.row_dict(sim.rows(0, 1)[0].cpu().tolist())         # (one transfer: ble_state_rows_f64)
    sim.check_errors()
    for k in STATE_FLOATS:
      e = float(rel_err(np.float32(got[k]), g[k][0, i + 1], FLOORS[k]))
      worst_state = max(worst_state, e)
      assert e <= 1e-5, (i, k, e)
    for k in ('status', 'last_command', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s'):
      assert int(got[k]) == int(g[k][0, i + 1]), (i, k)
    assert abs(float(reward[0]) - g['reward'][0, i]) <= 1e-5 and int(terminal[0]) == 0
  import multiprocessing as mp
  with mp.get_context('fork').Pool(min(16, len(jobs))) as pool:
    vectors = pool.map(_oracle_vector, jobs, chunksize=4)
  for (i, obs, want), same in zip(kept, vectors):
    err = np.abs(obs.astype(np.float64) - same)
    assert err.max() <= 1e-5, (i, err.max(), int(err.argmax()))
    # the reference's vector directly: 1e-5 + what the float32 rounding of the inputs does to the reference itself
    sens = np.abs(same - want.astype(np.float64))
    err_ref = np.abs(obs.astype(np.float64) - want.astype(np.float64))
    assert (err_ref - sens).max() <= 1e-5, (i, err_ref.max(), int((err_ref - sens).argmax()))
    worst_obs = max(worst_obs, float(err.max())); worst_ref = max(worst_ref, float(err_ref.max())); worst_sens = max(worst_sens, float(sens.max()))
    compared += 1
  assert compared >= 200
  print(f'F13 teacher-forced: 960/960 actions equal; on {compared} steps worst |obs diff| {worst_obs:.2e} vs the oracle on the same inputs, '
        f'{worst_ref:.2e} vs the reference (own input-rounding sensitivity {worst_sens:.2e}); worst state rel err {worst_state:.2e}')


def test_station_seeker_episode_free_running(vec_state):
  g = helpers.golden('f13_station_seeker')
  field = helpers.fixture_field(g)
  n = int(g['n_flown'])
  sim = vec_state.VecSimulator(1)
  sim.set_grid(torch.from_numpy(field).cuda())
  sim.set_state(_arrays(helpers.feature_row(g, 0, 0)))
  first_diff = None; rewards = []
  for i in range(n):
    noise = torch.from_numpy(g['noise_uv'][0, i:i + 1].astype(np.float32)).cuda()   # the fixture's (step-indexed) noise
    obs = sim.observe(noise).cpu().numpy()[0]
    a = sso.pick_action(obs)
    if first_diff is None and a != g['actions'][0, i]:
      first_diff = i
    reward, terminal = sim.step(torch.tensor([a], dtype=torch.uint8).cuda(), noise)
    rewards.append(float(reward[0]))
    assert int(terminal[0]) == 0, f'terminated at step {i}'
  torch.cuda.synchronize(); sim.check_errors()
  got = sim.get_state()
  ref_mean = float(g['reward'][0, :n].mean())
  d_km = float(np.hypot(got['x'][0], got['y'][0])) / 1000.0
  ref_d_km = float(np.hypot(g['x'][0, n], g['y'][0, n])) / 1000.0
  print(f'F13 free-running: first action difference at step {first_diff} of {n}; mean reward {np.mean(rewards):.4f} '
        f'(reference {ref_mean:.4f}); final distance {d_km:.1f} km (reference {ref_d_km:.1f} km)')
  assert int(got['time_elapsed_s'][0]) == 180 * n
  if first_diff is None:       # bit-for-bit the same decisions: the trajectories then agree to the transition's tolerance
    for k in ('x', 'y', 'pressure', 'battery_charge'):
      assert float(rel_err(got[k][0], g[k][0, n], FLOORS[k])) <= 1e-3, k
  # either way it is a station-seeking flight of the same quality
  assert abs(np.mean(rewards) - ref_mean) < 0.1
