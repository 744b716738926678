#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output of profiles/run_profile.sh into committed evidence.

  python profiles/summarize.py r01      # reads gpurun_out/prof_r01, writes profiles/r01_*.{json,md}
Per-launch averages for ble_step_kernel only.  HBM traffic follows the microarch guide:
FETCH_SIZE / WRITE_SIZE are in KiB... here reported by rocprofv3 in KB units of 1024 B;
on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads (x2 correction
applies to 16 B/lane streaming reads only; this kernel's reads are 4 B/lane and 8 B gathers,
so both the raw and the x2 figure are recorded).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
  base = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
  out = {'tag': tag, 'kernel': 'ble_step_kernel', 'agent_steps_per_profiled_launch': 32.0}
  # kernel stats
  with open(os.path.join(base, 'trace', 'trace_kernel_stats.csv')) as f:
    rows = list(csv.DictReader(f))
  md = ['| kernel | calls | avg us | min us | max us | % |', '|---|---|---|---|---|---|']
  for r in rows[:8]:
    name = r['Name'][:70]
    md.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
              f"{float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
    if 'ble_step_kernel' in r['Name']:
      out['kernel_trace'] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3,
                             'min_us': float(r['MinNs']) / 1e3, 'max_us': float(r['MaxNs']) / 1e3}
  # the driver-shaped run (one 20-step launch per region) and the one-launch-per-step run, from their raw kernel traces
  for sub, key, pick in (('trace20', 'kernel_trace_20_step_launches', lambda durs: durs[1:]),       # [0] is the 5-step warm-up launch
                         ('trace1', 'kernel_trace_1_step_launches', lambda durs: durs[44:])):        # 44 warm-up launches
    p = os.path.join(base, sub, f'{sub}_kernel_trace.csv')
    if os.path.exists(p):
      with open(p) as f:
        durs = [(int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
                for r in csv.DictReader(f) if 'ble_step_kernel' in r['Kernel_Name']]
      durs = pick([d for _, d in sorted(durs)])
      if durs:
        srt = sorted(durs)
        out[key] = {'launches': len(durs), 'avg_us': sum(durs) / len(durs), 'median_us': srt[len(srt) // 2], 'min_us': srt[0], 'max_us': srt[-1]}
        md.append(f"| `ble_step_kernel`, {key.replace('kernel_trace_', '').replace('_', ' ')} | {len(durs)} | {out[key]['avg_us']:.2f} | {srt[0]:.2f} | {srt[-1]:.2f} | (raw trace: {sub}) |")
  counters = collections.defaultdict(list)
  meta = {}
  for name in sorted(os.listdir(base)):
    p = os.path.join(base, name, f'{name}_counter_collection.csv')
    if not os.path.exists(p) or name.startswith('pmc_obs'):     # pmc_obs*: the --observe leg, summarised below
      continue
    with open(p) as f:
      for r in csv.DictReader(f):
        if 'ble_step_kernel' not in r['Kernel_Name']:
          continue
        counters[r['Counter_Name']].append(float(r['Counter_Value']))
        meta = {'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size']), 'vgpr': int(r['VGPR_Count']),
                'agpr': int(r['Accum_VGPR_Count']), 'sgpr': int(r['SGPR_Count']), 'scratch': int(r['Scratch_Size']),
                'lds': int(r['LDS_Block_Size'])}
  out['dispatch'] = meta
  avg = {k: sum(v) / len(v) for k, v in counters.items()}
  out['pmc_per_launch'] = avg
  waves = avg.get('SQ_WAVES', 0) or 1
  d = {}
  if 'SQ_INSTS_VALU' in avg:
    d['valu_insts_per_wave'] = avg['SQ_INSTS_VALU'] / waves
    d['salu_insts_per_wave'] = avg.get('SQ_INSTS_SALU', 0) / waves
    d['vmem_insts_per_wave'] = avg.get('SQ_INSTS_VMEM', 0) / waves
    d['wave_cycles_per_wave_quad'] = avg.get('SQ_WAVE_CYCLES', 0) / waves
  for k in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_TRANS_F64',
            'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_TRANS_F32',
            'SQ_INSTS_VALU_CVT', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_INT64', 'SQ_INSTS_BRANCH'):
    if k in avg:
      d[k.replace('SQ_INSTS_', '').lower() + '_per_wave'] = avg[k] / waves
  if 'SQ_ACTIVE_INST_VALU' in avg and 'SQ_WAIT_INST_ANY' in avg:
    d['active_inst_valu_quad'] = avg['SQ_ACTIVE_INST_VALU'] / waves
    d['active_inst_any_quad'] = avg.get('SQ_ACTIVE_INST_ANY', 0) / waves
    d['wait_inst_any_quad'] = avg['SQ_WAIT_INST_ANY'] / waves
    d['wait_any_quad'] = avg.get('SQ_WAIT_ANY', 0) / waves
  out['derived'] = d
  if 'FETCH_SIZE' in avg or 'WRITE_SIZE' in avg:
    fetch = avg.get('FETCH_SIZE', 0.0) * 1024.0
    write = avg.get('WRITE_SIZE', 0.0) * 1024.0
    out['hbm'] = {'fetch_bytes_raw': fetch, 'write_bytes_raw': write, 'hbm_bytes_per_launch': fetch + write,
                  'hbm_bytes_per_launch_fetch_x2': 2 * fetch + write,
                  'tcc_hit': avg.get('TCC_HIT_sum'), 'tcc_miss': avg.get('TCC_MISS_sum')}
    traffic = {'hbm_bytes_per_launch': fetch + write, 'hbm_bytes_per_launch_fetch_x2': 2 * fetch + write,
               'source': f'profiles/{tag}_summary.json'}
  # ---- observation kernel (bench.py --observe leg)
  obs_md = []
  obs_stats = os.path.join(base, 'trace_obs', 'trace_obs_kernel_stats.csv')
  if os.path.exists(obs_stats):
    with open(obs_stats) as f:
      rows = list(csv.DictReader(f))
    obs_md = ['| kernel | calls | avg us | min us | max us | % |', '|---|---|---|---|---|---|']
    for r in rows[:6]:
      obs_md.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                    f"{float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
      if 'ble_observe_kernel' in r['Name']:
        out['observe_kernel_trace'] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3,
                                       'min_us': float(r['MinNs']) / 1e3, 'max_us': float(r['MaxNs']) / 1e3}
    oc = collections.defaultdict(list)
    for name in ('pmc_obs1', 'pmc_obs2', 'pmc_obs3', 'pmc_obs_fetch', 'pmc_obs_write'):
      p = os.path.join(base, name, f'{name}_counter_collection.csv')
      if os.path.exists(p):
        with open(p) as f:
          for r in csv.DictReader(f):
            if 'ble_observe_kernel' in r['Kernel_Name']:
              oc[r['Counter_Name']].append(float(r['Counter_Value']))
              out['observe_dispatch'] = {'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size']), 'vgpr': int(r['VGPR_Count']),
                                         'agpr': int(r['Accum_VGPR_Count']), 'sgpr': int(r['SGPR_Count']),
                                         'scratch': int(r['Scratch_Size']), 'lds': int(r['LDS_Block_Size'])}
    # the last launches are the steady state (window full); average the final 8
    steady = {k: sum(v[-8:]) / len(v[-8:]) for k, v in oc.items()}
    out['observe_pmc_per_launch_steady'] = steady
    if 'FETCH_SIZE' in steady and 'WRITE_SIZE' in steady:
      # the factor (59 KB per env, 90 % of the reads) is fetched with 16 B per lane: the guide's x2 correction
      # applies to FETCH_SIZE; WRITE_SIZE is uncalibrated (guide) and reported as is
      of, ow = steady['FETCH_SIZE'] * 1024.0, steady['WRITE_SIZE'] * 1024.0
      out['observe_hbm'] = {'fetch_bytes_raw': of, 'write_bytes_raw': ow, 'hbm_bytes_per_launch_fetch_x2': 2 * of + ow,
                            'algorithmic_bytes_per_launch': 65536 * (4396 + 2 * 60960 + 152 + 3072)}
      try:
        traffic
      except NameError:
        traffic = {}
      traffic['observe_hbm_bytes_per_launch'] = 2 * of + ow
      traffic['observe_note'] = 'FETCH_SIZE x2 (16 B/lane streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE, steady-state launch, 65 536 envs' 
  # ---- the small-batch kernel (ble_step_split_kernel) against the one-lane kernel at 8 192 environments, 32-step launches
  split_md = []
  sp_trace = os.path.join(base, 'trace_split', 'trace_split_kernel_trace.csv')
  if os.path.exists(sp_trace):
    durs = collections.defaultdict(list)
    with open(sp_trace) as f:
      for r in csv.DictReader(f):
        for key in ('ble_step_split_kernel', 'ble_step_kernel'):
          if key in r['Kernel_Name']:
            durs[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    sc = {k: collections.defaultdict(list) for k in ('ble_step_split_kernel', 'ble_step_kernel')}
    for name in ('pmc_split1', 'pmc_split2'):
      p = os.path.join(base, name, f'{name}_counter_collection.csv')
      if os.path.exists(p):
        with open(p) as f:
          for r in csv.DictReader(f):
            for key in sc:
              if key in r['Kernel_Name']:
                sc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    out['split'] = {}
    split_md = ['| kernel (8 192 environments, 32 agent steps per launch) | launches | avg us per launch | us per agent step | waves | VALU / wave / step | of which fp64 | SALU / wave / step | LDS / wave / step | issue utilisation | waiting |',
                '|---|---|---|---|---|---|---|---|---|---|---|']
    for key in ('ble_step_split_kernel', 'ble_step_kernel'):
      if not durs[key]:
        continue
      d = sorted(durs[key])[:-1] or durs[key]            # (drop the slowest: the first, cold launch)
      a = {k: sum(v) / len(v) for k, v in sc[key].items()}
      w = a.get('SQ_WAVES', 0) or 1
      f64 = sum(a.get(k, 0) for k in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_TRANS_F64'))
      e = {'launches': len(d), 'avg_us': sum(d) / len(d), 'us_per_step': sum(d) / len(d) / 32.0, 'waves': w,
           'valu_per_wave_step': a.get('SQ_INSTS_VALU', 0) / w / 32.0, 'fp64_per_wave_step': f64 / w / 32.0,
           'salu_per_wave_step': a.get('SQ_INSTS_SALU', 0) / w / 32.0, 'lds_per_wave_step': a.get('SQ_INSTS_LDS', 0) / w / 32.0,
           'issue_utilisation': a.get('SQ_ACTIVE_INST_ANY', 0) / (a.get('SQ_WAVE_CYCLES', 0) or 1),
           'wait_frac': a.get('SQ_WAIT_ANY', 0) / (a.get('SQ_WAVE_CYCLES', 0) or 1), 'pmc_per_launch': a}
      out['split'][key] = e
      split_md.append(f"| `{key}` | {e['launches']} | {e['avg_us']:.1f} | {e['us_per_step']:.2f} | {w:.0f} | {e['valu_per_wave_step']:.0f} | {e['fp64_per_wave_step']:.0f} | "
                      f"{e['salu_per_wave_step']:.0f} | {e['lds_per_wave_step']:.1f} | {e['issue_utilisation']:.3f} | {e['wait_frac']:.3f} |")
  try:
    json.dump(traffic, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'))
  except NameError:
    pass
  json.dump(out, open(os.path.join(ROOT, 'profiles', f'{tag}_summary.json'), 'w'), indent=1)
  with open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.md'), 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats, `python bench.py --steps 192 --warmup 32 --reps 5 --no-extras` (one launch = 32 agent steps) ({tag})\n\n')
    f.write('\n'.join(md) + '\n\n')
    f.write('## ble_step_kernel PMC (per launch averages; separate --pmc passes)\n\n```json\n')
    f.write(json.dumps({k: out[k] for k in ('dispatch', 'derived', 'hbm') if k in out}, indent=1))
    f.write('\n```\n')
    if obs_md:
      f.write('\n# `python profiles/obs_only.py 65536`: 121 window-filling + 8 steady-state step/noise/observation triples\n\n')
      f.write('\n'.join(obs_md) + '\n\n## ble_observe_kernel PMC (steady state: last 8 launches)\n\n```json\n')
      f.write(json.dumps({k: out[k] for k in ('observe_dispatch', 'observe_pmc_per_launch_steady', 'observe_hbm') if k in out}, indent=1))
      f.write('\n```\n')
    if split_md:
      f.write('\n# `python profiles/split_launches.py 8192 16`: the small-batch form of the transition (one environment on four wavefronts) against the one-lane kernel\n\n')
      f.write('\n'.join(split_md) + '\n')
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main(sys.argv[1] if len(sys.argv) > 1 else 'r01')
