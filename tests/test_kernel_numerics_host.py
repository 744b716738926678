"""Numerics of the kernel's lane functions compiled ON THE HOST (tests/emul) against the oracle.

The HIP kernel (balloon_learning_environment_amd/csrc/*.h) is plain C++ with a thin
intrinsic layer, so the same source builds with g++.  This covers the arithmetic design
(mixed fp32/fp64 chain, atmosphere window, quadratic solar interpolation, safety layers)
on machines without a GPU.  It is not the product path -- the -m gpu tests exercise the
real library through the C ABI -- and libm here stands in for v_exp/v_log/v_rcp/v_sqrt.
"""
import numpy as np
import pytest

import oracle
from helpers import FLOORS, STATE_FLOATS, golden, rel_err, traj_state_at

def _load_emul():
  from emul import emul as e     # tests/emul/emul.py (builds libble_emul.so with g++ on first use)
  return e


def _compare(st, o2, ctx):
  for k in ('status', 'alt_fsm', 'env_fsm', 'power_paused', 'time_elapsed_s'):
    np.testing.assert_array_equal(st[k], o2[k], err_msg=f'{ctx} {k}')
  for k in STATE_FLOATS:
    e = rel_err(st[k], o2[k], FLOORS[k])
    assert e.max() <= 1e-5, f'{ctx} {k}: {e.max():.3g} at {e.argmax()}'


@pytest.mark.parametrize('name,use_field', [('f8_trajectories', False), ('f9_arena', True)])
def test_golden_trajectories_teacher_forced(name, use_field):
  e = _load_emul()
  d = golden(name)
  n, steps = d['actions'].shape
  valid = d['valid'] if 'valid' in d.files else np.ones((n, steps), np.uint8)
  field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32) if use_field else None
  for s in range(steps):
    rows = np.nonzero(valid[:, s])[0]
    st = e.state_from_oracle(traj_state_at(d, s, rows))
    o2 = e.oracle_from_state(st)
    w = None if use_field else d['wind_uv'][rows, s].astype(np.float32)
    r, t, eff, fl = e.step(st, d['actions'][rows, s], wind_uv=w, field=field)
    ro, to, eo, err = oracle.step(o2, d['actions'][rows, s], wind_uv=None if w is None else w.astype(np.float64), field=field)
    assert fl == 0 and err == 0
    _compare(st, o2, f'{name} step {s}')
    np.testing.assert_array_equal(eff, eo)
    np.testing.assert_allclose(r, ro, rtol=1e-5, atol=1e-5)


def test_random_states_and_layer_transition_chatter():
  """2 048 sampled initial states, 6 free-running steps each (teacher-forced per step), plus
  balloons parked within +-30 Pa of the 17 km lapse-rate transition, where the pressure
  chatters across the layer boundary every substep."""
  e = _load_emul()
  from balloon_learning_environment_amd import reset_host
  n = 2048
  init = reset_host.sample_initial_state(n, seed=11)
  # park a quarter of the balloons at the transition pressure of their own atmosphere
  atm = reset_host.AtmosphereTables(init['alpha'])
  k = n // 4
  rng = np.random.default_rng(5)
  init['pressure'][:k] = np.float32(atm.pres[:k, 1] + rng.uniform(-30, 30, k))
  ost = oracle.new_state(n)
  for f in oracle.FLOAT_FIELDS:
    ost[f][:] = np.asarray(init[f], np.float64)
  for f in oracle.U8_FIELDS:
    ost[f][:] = init[f]
  ost['start_unix'][:] = init['start_unix']
  ost['sunrise_h'][:] = init['start_unix'] + init['sunrise_h_rel']; ost['sunset'][:] = init['start_unix'] + init['sunset_rel']
  field = (np.random.default_rng(0).standard_normal((21, 21, 10, 9, 2)) * 5).astype(np.float32)
  st = e.state_from_oracle(ost)
  crossings = 0
  for s in range(6):
    live = st['status'] == 0
    o2 = e.oracle_from_state(st)
    act = rng.integers(0, 3, n).astype(np.uint8)
    p_before = st['pressure'].copy()
    r, t, eff, fl = e.step(st, act, field=field)
    ro, to, eo, err = oracle.step(o2, act, field=field)
    assert fl == 0 and (err & ~oracle.ERR_TERMINAL_STEP) == 0
    _compare({k_: v[live] for k_, v in st.items()}, {k_: v[live] for k_, v in o2.items()}, f'random step {s}')
    np.testing.assert_array_equal(eff[live], eo[live])
    np.testing.assert_allclose(r[live], ro[live], rtol=1e-5, atol=1e-5)
    p1 = atm.pres[:, 1]
    crossings += int(((p_before[:k] - p1[:k]) * (st['pressure'][:k] - p1[:k]) < 0).sum())
  assert crossings > 50   # the transition really is being crossed
