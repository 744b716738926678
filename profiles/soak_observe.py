"""Long closed-loop soak of the observation kernel at full occupancy: two simulators in lockstep -- carried, slid factor
vs a refit at every call -- over `steps` agent steps of `n` environments with random actions, wind noise, skipped
observations, feature reads without an append, and immediate new episodes for the terminated environments.  Every `every`
steps ALL environments are compared; error flags are checked at the end.

  python profiles/soak_observe.py [n=65536] [steps=1500] [every=25] [drain=0]
`drain` > 0: every 37th step the batteries of that fraction of the environments are emptied (they run out of power and
start new episodes: the reset path of the history ring and of the carried factor under load).
"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from balloon_learning_environment_amd import vec_state

def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
  every = int(sys.argv[3]) if len(sys.argv) > 3 else 25
  drain = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
  rng = np.random.default_rng(5)
  field = torch.from_numpy((rng.standard_normal((21, 21, 10, 9, 2)) * 5.0).astype(np.float32)).cuda()
  sims = []
  for carry in (True, False):
    s = vec_state.VecSimulator(n); s.set_grid(field); s.reset_device(seed=77); sims.append(s)
  gen = torch.Generator(device='cuda'); gen.manual_seed(3)
  obs = [torch.empty(n, 1099, dtype=torch.float32, device='cuda') for _ in sims]
  worst = 0.0; compared = 0; beyond = 0; total = 0; resets = 0
  skipping = 0
  t0 = time.time()
  for i in range(steps):
    act = torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda', generator=gen)
    noise = torch.randn((n, 2), dtype=torch.float32, device='cuda', generator=gen)
    if drain > 0 and i % 37 == 36:
      sel = torch.rand(n, device='cuda', generator=gen) < drain
      for s in sims: s.state['battery_charge'][sel] = 0.01
    for s in sims: s.step(act)
    # a terminated environment starts a new episode at once (same seeds on both sides): left frozen, its repeated
    # observations at one time stamp would overflow the 6 h window -- the kernel reports that, rightly
    mask = (sims[0].state['status'] != 0).to(torch.uint8).contiguous()
    if i % 50 == 49:
      assert torch.equal(mask, (sims[1].state['status'] != 0).to(torch.uint8))
      resets_dbg = int(mask.sum())
    for s in sims: s.reset_device(seed=1000 + i, mask=mask)
    resets_t = mask.sum() if i == 0 else resets_t + mask.sum()
    if skipping > 0:
      skipping -= 1; continue
    if rng.random() < 0.01: skipping = int(rng.integers(1, 30))
    append = bool(rng.random() > 0.05)
    for s, o, carry in zip(sims, obs, (True, False)): s.observe(noise, append=append, out=o, carry_factor=carry)
    if i % every == 0 or i == steps - 1:
      live = sims[0].state['status'] == 0
      d = (obs[0].double() - obs[1].double()).abs()[live]
      worst = max(worst, float(d.max())); compared += 1
      beyond += int((d.amax(dim=1) > 1e-5).sum()); total += int(live.sum())
      assert torch.isfinite(obs[0][live]).all()
  for s in sims: s.check_errors()
  print(f'soak: {n} envs x {steps} steps in {time.time() - t0:.0f} s; {compared} full comparisons ({total} env-observations), '
        f'worst |carried - refit| {worst:.3g}, beyond 1e-5: {beyond} ({beyond / max(1, total):.2e}); episodes restarted: {int(resets_t)}')
  assert worst <= 2e-4 and beyond <= 1e-3 * total

if __name__ == '__main__':
  main()
