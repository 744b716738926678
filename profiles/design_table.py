"""DESIGN.md section 8's table of measured rates, generated from the committed bench lines:

  python profiles/design_table.py r05            # prints the table
  python profiles/design_table.py r05 --write    # replaces the block between the table markers of DESIGN.md

Inputs: profiles/<tag>_bench.json (default run), profiles/<tag>_bench_driver_shape.json (the driver's flags),
profiles/<tag>_summary.json (rocprofv3 averages, profiles/summarize.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = '<!-- rates:begin (profiles/design_table.py) -->', '<!-- rates:end -->'


def line(path):
  with open(path) as f:
    return json.loads(f.read().strip().splitlines()[-1])


def sci(v, digits=2):
  e = 0
  while v >= 10.0:
    v /= 10.0; e += 1
  sup = str.maketrans('0123456789', '⁰¹²³⁴⁵⁶⁷⁸⁹')
  return f'{v:.{digits}f} × 10{str(e).translate(sup)}'


def rows(d, s):
  c = d['configs']
  head, one, gt = c['configs[2]'], c['configs[2] one launch per step'], c['configs[2] in the ground-truth wind (noise in-kernel)']
  c1 = c['configs[1]']
  c3 = next(v for k, v in c.items() if k.startswith('configs[3]'))
  c4, fac = c['configs[4]'], c['configs[0] counterpart: single-env facade']
  r, o, cpu = d['roofline'], d['observe'], d['cpu_baseline']
  spl = int(round(r['agent_steps_per_launch']))
  prof = s['kernel_trace'] if spl == 32 else s['kernel_trace_20_step_launches']
  return [
      f"{fac['steps_per_s']:.0f} steps/s ({fac['ms_per_step']:.2f} ms per step)",
      f"**{sci(c1['env_steps_per_s'])}** env-steps/s ({c1['ms_per_step'] * 1e3:.2f} µs per step; {sci(c1['env_steps_per_s_ground_truth_wind'])} in the "
      f"ground-truth wind; one launch per step {c1['one_launch_per_step']['us_per_step_back_to_back']:.1f} µs)",
      f"**{sci(head['env_steps_per_s'])} env-steps/s ({head['ms_per_step'] * 1e3:.2f} µs per step of wall clock; kernel {r['kernel_ms']:.3f} ms per "
      f"{spl}-step launch by HIP events; rocprofv3 {prof['avg_us']:.0f} µs average under the profiler)**",
      f"{sci(gt['env_steps_per_s'])} ({gt['ms_per_step'] * 1e3:.1f} µs per step)",
      vehicle_leg(c),
      f"{sci(one['env_steps_per_s'])} ({one['us_per_step_back_to_back']:.1f} µs per step back to back; an isolated launch "
      f"{one['us_per_launch_event_median']:.1f} µs between an event pair; rocprofv3 {s['kernel_trace_1_step_launches']['avg_us']:.1f} µs per kernel)",
      f"**{sci(c3['env_steps_per_s'])}** ({c3['ms_per_step'] * 1e3:.2f} µs per step; {sci(c3['env_steps_per_s_ground_truth_wind'])} in the ground-truth wind)",
      f"**{sci(c4['env_steps_per_s'])}** ({c4['ms_per_step'] * 1e3:.1f} µs per step)",
      f"{sci(o['env_steps_per_s_with_observation'])} env-steps/s (observation launch {o['ms_per_observation_launch_median']:.2f} ms)",
      closed_loops(d),
      f"{sci(cpu['value'])}; {sci(cpu['value_single_thread'])}",
  ]


def vehicle_leg(c):
  v = next((x for k, x in c.items() if 'run-time vehicle' in k), None)
  return 'n/a' if v is None else f"{sci(v['env_steps_per_s'])} ({v['ms_per_step'] * 1e3:.1f} µs per step; {v['live_env_fraction_end']:.3f} of the balloons flying at the end)"


def closed_loops(d):
  """step + noise + observation at the other configs' batch sizes (bench.py `observe_configs`, round 6)."""
  oc = d.get('observe_configs') or {}
  parts = []
  for k, v in oc.items():
    parts.append(f"{v['envs']:,}".replace(',', ' ') + (' (per-env grids)' if v['per_env_grids'] else '') +
                 f": {sci(v['env_steps_per_s_with_observation'])} ({v['ms_per_observation_launch']:.2f} ms per observation launch, fp64 frac {v['fp64_frac_algorithmic']:.2f})")
  return '; '.join(parts) if parts else 'n/a'


LABELS = [
    'configs[0] counterpart: single-env façade `BalloonEnv.step` (host-synchronous gym API, device observation, wind noise on)',
    'configs[1]: 4 096 envs (four-wave kernel)',
    '**configs[2]: 65 536 envs (headline)**',
    'configs[2] in the ground-truth wind (noise generated in-kernel)',
    'configs[2] with a run-time flight vehicle (ABI 5: `ble_step_kernel<VehicleRt>`)',
    'configs[2], one launch per agent step (`ble_step_f32`, policy in the loop)',
    'configs[3]: one GPU\'s share, 8 192 envs (four-wave kernel)',
    'configs[4]: one GPU\'s share, 32 768 envs with per-env grids (four-wave kernel, two waves per SIMD)',
    'closed loop: step + noise + observation, 65 536 envs',
    'closed loop at the other configs\' sizes: 4 096 (configs[1]) / 8 192 (configs[3] share) / 32 768 with per-env grids (configs[4] share)',
    'CPU baseline (fp64 C oracle, 16 threads; 1 thread)',
]


def table(tag):
  s = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_summary.json')))
  a = line(os.path.join(ROOT, 'profiles', f'{tag}_bench.json'))
  b = line(os.path.join(ROOT, 'profiles', f'{tag}_bench_driver_shape.json'))
  out = [f"| Config | default run ({a['steps']}-step regions of {a['config']['exchanges']['launches_per_timed_region']} launches, median of "
         f"{a['repetitions']['repetitions']}) | driver's flags (`--gpus 1 --steps {b['steps']} --warmup {b['warmup']}`) |", '|---|---|---|']
  for label, x, y in zip(LABELS, rows(a, s), rows(b, s)):
    out.append(f'| {label} | {x} | {y} |')
  r, o = a['roofline'], a['observe']['roofline']
  ii = r['instruction_issue']
  out += ['',
          f"`roofline` of the default run: bound {r['bound']}, frac **{r['frac']:.2f}** (wave issue utilisation {r['wave_issue_utilisation']:.2f}; "
          f"{ii['valu_insts_per_env_step']:.0f} vector + {ii['salu_insts_per_env_step']:.0f} scalar instructions per env-step);",
          f"`hbm_formal` {r['hbm_formal']['achieved']:.0f} GB/s = {r['hbm_formal']['frac']:.3f} of 8 TB/s (280 B × live env-steps ÷ kernel time); "
          f"`hbm_measured` {r['hbm_measured']['achieved']:.0f} GB/s = {r['hbm_measured']['frac']:.4f} ({r['hbm_measured']['ratio_to_algorithmic']:.3f} × the "
          f"algorithmic bytes: {r['traffic'] / 1e6:.1f} MB per launch by two rocprofv3 --pmc passes).",
          f"Observation leg: roofline frac {o['frac']:.2f} of the fp64 peak on algorithmic flops, {o['traffic'] / 1e9:.2f} GB per launch measured."]
  return '\n'.join(out)


if __name__ == '__main__':
  tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'r05'
  text = table(tag)
  if '--write' in sys.argv:
    path = os.path.join(ROOT, 'DESIGN.md')
    doc = open(path).read()
    i, j = doc.index(BEGIN), doc.index(END)
    open(path, 'w').write(doc[:i + len(BEGIN)] + '\n' + text + '\n' + doc[j:])
  else:
    print(text)
