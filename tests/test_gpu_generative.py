"""VAE decoder on the device (BASELINE config 5's per-env forecasts).
Weights are synthetic and the reference pins only the output shape (vae_test.py:43-54), so the
checks are: kernel arithmetic vs oracle/vae_oracle.py (fp32: 1e-5 relative to the field scale),
the MLP vs a float64 NumPy product (1e-4), the discrete incompressibility the construction
guarantees, and that the step / observation kernels fly on per-env decoded grids."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def sampler():
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device; none visible')
  from balloon_learning_environment_amd.env import generative_wind_field
  return generative_wind_field.GenerativeWindFieldSampler(seed=3)


def test_decoder_tail_matches_oracle(sampler):
  import vae_oracle
  rng = np.random.default_rng(0)
  for n in (1, 5, 300):
    flow = rng.standard_normal((n, 4410)).astype(np.float32) * 30
    out = torch.empty((n, 21, 21, 10, 9, 2), dtype=torch.float32, device='cuda')
    from balloon_learning_environment_amd import _lib
    _lib.check(_lib.lib().ble_decode_flow_fields_f32(torch.from_numpy(flow).cuda().data_ptr(), out.data_ptr(), n, 0), 'decode')
    torch.cuda.synchronize()
    want = vae_oracle.decode_flow(flow)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=1e-5 * np.abs(want).max())


def test_decoder_shape_and_mlp(sampler):
  import vae_oracle
  from balloon_learning_environment_amd.env import generative_wind_field as g
  # vae_test.py:43-54: decoding a latent gives a (21, 21, 10, 9, 2) field
  field = sampler.sample_field(np.array([0, 7], np.uint32), None)
  assert field.shape == (21, 21, 10, 9, 2) and field.dtype == np.float32 and np.isfinite(field).all()
  assert 0.5 < np.abs(field).mean() < 50.0                       # a wind field in m/s, not zeros
  z = sampler.sample_latents(16, seed=1)
  params = [(w.cpu().numpy(), b.cpu().numpy()) for w, b in sampler.params]
  want = vae_oracle.mlp(z.cpu().numpy(), params)
  got = sampler.flow_fields(z).cpu().numpy()
  np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * np.abs(want).max())
  grids = sampler.decode(z).cpu().numpy()
  np.testing.assert_allclose(grids, vae_oracle.decode_flow(got), rtol=0, atol=1e-5 * np.abs(grids).max())
  # u = d psi / da and v = -d psi / db of one stream function psi(a, b): the centred discrete
  # divergence du/db + dv/da vanishes identically (the four psi terms cancel), i.e. to rounding
  u, v = grids[..., 0].astype(np.float64), grids[..., 1].astype(np.float64)
  div = (u[:, 1:-1, 2:] - u[:, 1:-1, :-2]) + (v[:, 2:, 1:-1] - v[:, :-2, 1:-1])
  assert np.abs(div).max() < 2e-5 * np.abs(u).max()
  assert g.GRID_FLOATS == 79380


def test_step_and_observe_on_per_env_decoded_grids(sampler):
  from balloon_learning_environment_amd import vec_state
  n = 96
  grids = sampler.decode(sampler.sample_latents(n, seed=11))
  sim = vec_state.VecSimulator(n)
  sim.set_grid(grids, per_env=True)
  sim.reset_device(seed=5)
  acts = torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda')
  x0 = sim.state['x'].clone()
  for _ in range(3):
    sim.step(acts)
    obs = sim.observe()
  sim.check_errors()
  assert obs.shape == (n, 1099) and torch.isfinite(obs).all()
  moved = (sim.state['x'] - x0).abs()
  assert (moved > 1.0).float().mean() > 0.9           # the decoded winds move the balloons
  # env k flies in grid k: the displacement of env 0 equals that of a 1-env simulator on grid 0
  single = vec_state.VecSimulator(1)
  single.set_grid(grids[0].contiguous())
  state = {k: v[:1] for k, v in sim.get_state().items()}
  # (compare one more step from the same state)
  single.set_state(state)
  both = vec_state.VecSimulator(1); both.set_grid(grids[:1].contiguous(), per_env=True); both.set_state(state)
  a = torch.tensor([2], dtype=torch.uint8, device='cuda')
  single.step(a); both.step(a)
  for k in ('x', 'y', 'pressure'):
    assert torch.equal(single.state[k], both.state[k])
