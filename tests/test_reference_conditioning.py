"""How well-conditioned is the reference's own transition?  (CPU, oracle only.)

north_star asks for "within 1e-5 relative on float32 balloon state".  The state handed to the
transition is float32, so the fp64 reference itself only sees its inputs to 1 ulp (6e-8).  This
test perturbs every mutable fp32 input of the oracle by +-1 ulp and measures how far the fp64
oracle's own output moves: explicit-Euler vertical dynamics with dh/dt = +-sqrt(|rho V - m| ...)
(env/balloon/balloon.py:412-445) amplify differences by many orders of magnitude on a small
fraction of environments.  The numbers printed here are the context for the GPU parity tests
(tests/test_gpu_parity.py), which hold EVERY environment to 1e-5 on identical inputs: that is only
reachable because the kernel carries the whole vertical chain (including the thermal and ACS
increments) in fp64 -- an fp32 increment is a perturbation of exactly this kind.
"""
import os

import numpy as np

import oracle
from helpers import FLOORS, STATE_FLOATS, rel_err

MUTABLE = ('x', 'y', 'pressure', 'ambient_temperature', 'internal_temperature', 'envelope_volume',
           'superpressure', 'mols_air', 'battery_charge')


def _oracle_state(init):
  n = init['x'].size
  ost = oracle.new_state(n)
  for f in oracle.FLOAT_FIELDS:
    ost[f][:] = np.asarray(init[f], np.float32).astype(np.float64)     # exactly the fp32 values
  for f in oracle.U8_FIELDS:
    ost[f][:] = init[f]
  ost['start_unix'][:] = init['start_unix']; ost['time_elapsed_s'][:] = init['time_elapsed_s']
  ost['sunrise_h'][:] = init['start_unix'] + init['sunrise_h_rel']; ost['sunset'][:] = init['start_unix'] + init['sunset_rel']
  return ost


def test_one_ulp_input_perturbation_moves_the_reference_beyond_1e_5():
  import reset_host
  n, steps = 65536, 3                        # the sample of tests/test_gpu_parity.py::test_config_65536_envs_full_size
  threads = min(16, os.cpu_count() or 1)
  field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5.0).astype(np.float32)
  ref = _oracle_state(reset_host.sample_initial_state(n, seed=43))
  rng = np.random.default_rng(44)
  beyond = []; worst = 0.0; medians = []
  for s in range(steps):
    # round the carried reference state to fp32 -- what the GPU batch (and any fp32 env) holds
    for f in oracle.FLOAT_FIELDS:
      ref[f][:] = ref[f].astype(np.float32).astype(np.float64)
    pert = {k: v.copy() for k, v in ref.items()}
    for k in MUTABLE:
      a = ref[k].astype(np.float32)
      up = rng.integers(0, 2, n).astype(bool)
      pert[k][:] = np.where(up, np.nextafter(a, np.float32(np.inf)), np.nextafter(a, np.float32(-np.inf))).astype(np.float64)
    live = ref['status'] == 0
    act = rng.integers(0, 3, n).astype(np.uint8)
    oracle.step(ref, act, field=field, threads=threads)
    oracle.step(pert, act, field=field, threads=threads)
    moved = np.zeros(n)
    for k in STATE_FLOATS:
      e = rel_err(pert[k], ref[k], FLOORS[k]); e[~live] = 0.0
      moved = np.maximum(moved, e)
    beyond.append(int((moved > 1e-5).sum())); worst = max(worst, float(moved.max())); medians.append(float(np.median(moved)))
  print(f'1-ulp input perturbation of the fp64 reference, {n} envs x {steps} steps: env-steps moved beyond 1e-5 per step '
        f'{beyond}, worst {worst:.2e}, median {np.median(medians):.1e}')
  # measured: [30, 174, 143] beyond 1e-5, worst 6e-3, median 2e-7.  The assertions only pin the
  # qualitative fact (the reference is ill-conditioned on a small fraction of environments).
  assert sum(beyond) >= 20, 'the reference transition is better conditioned than documented: revisit DESIGN.md section 5'
  assert worst > 1e-4
  assert np.median(medians) < 2e-6           # ...while the typical environment is benign


def test_observation_sensitivity_to_float32_inputs():
  """The observation twin of the test above.  The device holds the state as float32 (north star); the reference's fixtures
  F11 were computed from float64 states.  Feeding the pinned feature oracle the float32-ROUNDED states of F11 moves its own
  1099-vectors by up to 1e-4 -- all of it on the bearing features (arccos(wind . to-station) / pi: a 6e-8 change of x next
  to an aligned wind) -- so no float32-state implementation can match the fixtures to 1e-5 on every entry; the GPU tests
  therefore hold the device to 1e-5 against the oracle on the SAME float32 inputs and allow this sensitivity, computed
  per entry, in the direct comparison with the fixture (tests/test_gpu_observe.py)."""
  import features_oracle
  import helpers
  g = helpers.golden('f11_features')
  field = helpers.fixture_field(g)
  worst, beyond, beyond_not_bearing = 0.0, 0, 0
  for j in range(3):
    fo = features_oracle.FeatureOracle(field, float(np.float32(g['alpha'][j])))
    for i in range(g['x'].shape[1]):
      row = helpers.feature_row(g, j, i)
      fu, fv = oracle.wind_forecast(field, row['x'], row['y'], row['pressure'], row['time_elapsed_s'])
      noise = np.array([g['wind_measured'][j, i, 0] - fu[0], g['wind_measured'][j, i, 1] - fv[0]], np.float32)
      fo.observe({k: (float(np.float32(v)) if isinstance(v, float) else v) for k, v in row.items()}, noise.astype(np.float64))
      if i % 3 == 2 or i > 40:
        d = np.abs(fo.features().astype(np.float64) - g['features'][j, i].astype(np.float64))
        worst = max(worst, float(d.max()))
        over = d > 1e-5
        beyond += int(over.sum())
        bearing = np.zeros(1099, bool); bearing[17::3] = True
        beyond_not_bearing += int((over & ~bearing).sum())
  print(f'reference features under float32 input rounding: worst {worst:.3g}, {beyond} entries beyond 1e-5, {beyond_not_bearing} of them not bearings')
  assert 1e-5 < worst < 1e-3 and beyond > 0 and beyond_not_bearing == 0
