"""Long lockstep flight of the forms of the transition: one lane per environment against four (and two) wavefronts per
environment, the same actions, noise and episode resets (terminated environments are re-initialised on the device every `reset_every`
steps), every state array compared bit for bit every `check_every` steps.
  python profiles/soak_split.py [n] [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balloon_learning_environment_amd import _lib, vec_state  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))      # (the host-side state sampler is test tooling)
import reset_host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
check_every, reset_every = 100, 37
field = (np.random.default_rng(0).standard_normal(vec_state.GRID_SHAPE) * 5.0).astype(np.float32)
init = reset_host.sample_initial_state(n, seed=77)
init['battery_charge'][: n // 16] = np.linspace(1.0, 400.0, n // 16).astype(np.float32)      # a steady trickle of episodes ending
sims = {}
for mode in ('0', '4') + (('2',) if os.environ.get('BLE_WITH_PAIR_FORM') else ()):      # '2': experiment builds only (-DBLE_WITH_PAIR_FORM)
  s = vec_state.VecSimulator(n); s.set_grid(field); s.set_state(init)
  sims[mode] = s
gen = torch.Generator(device='cuda'); gen.manual_seed(3)
ended = 0
compared = 0
for t in range(steps):
  acts = torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda', generator=gen)
  noise = None
  for mode, s in sims.items():
    _lib.set_step_form(mode)
    if noise is None:
      noise = s.wind_noise(seed=11).clone()       # (the same positions in all three: the same noise)
    s.step(acts, noise)
  if (t + 1) % reset_every == 0:
    mask = (sims['0'].state['status'] != 0).to(torch.uint8)
    ended += int(mask.sum().item())
    for s in sims.values():
      s.reset_device(seed=1000 + t, mask=mask)
  if (t + 1) % check_every == 0:
    torch.cuda.synchronize()
    ref = sims['0'].get_state()
    for mode in ('4',) + (('2',) if os.environ.get('BLE_WITH_PAIR_FORM') else ()):
      got = sims[mode].get_state()
      for name in ref:
        assert np.array_equal(ref[name], got[name]), (t, mode, name, int((ref[name] != got[name]).sum()))
    for s in sims.values():
      s.check_errors()
    compared += 1
_lib.set_step_form(None)
print(f'soak: {n} environments x {steps} agent steps ({n * steps:.3g} env-steps per kernel form), {compared} full comparisons of every state array: '
      f'one lane == ' + ' == '.join({'4': 'four waves', '2': 'two waves'}[m] for m in ('4',) + (('2',) if os.environ.get('BLE_WITH_PAIR_FORM') else ())) + ' bit for bit;' + f' {ended} episodes ended and were restarted; no error flag')
