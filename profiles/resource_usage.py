"""Kernel resource usage of the product build as a table: python profiles/resource_usage.py [out.txt]
(hipcc -Rpass-analysis=kernel-resource-usage with the flags of balloon_learning_environment_amd/_lib.py::build; no GPU needed)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'balloon_learning_environment_amd', 'csrc')
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=on', '-fPIC', '-I', SRC, '-c', '--cuda-device-only',
                    '-o', os.devnull, os.path.join(SRC, 'ble_kernels.hip'), '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
blocks = re.split(r'remark: Function Name: ', r.stderr)[1:]
keys = [('VGPRs', 'VGPRs'), ('AGPRs', 'AGPRs'), ('SGPRs', 'TotalSGPRs'), ('SGPR spills', 'SGPRs Spill'), ('VGPR spills', 'VGPRs Spill'),
        ('scratch B/lane', r'ScratchSize \[bytes/lane\]'), ('waves/SIMD', r'Occupancy \[waves/SIMD\]'), ('LDS B', r'LDS Size \[bytes/block\]')]
out = ['# Kernel resource usage of libble_hip.so (hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on, -Rpass-analysis=kernel-resource-usage).',
       '# "VGPR spills" of the one-lane step kernel go to AGPRs (no scratch): at one wave per SIMD the accumulator file is idle.', '',
       '| kernel | ' + ' | '.join(k for k, _ in keys) + ' |', '|' + '---|' * (len(keys) + 1)]
for b in blocks:
  mangled = b.split()[0]
  m = re.search(r'(\d+)(ble_\w+kernel|probe_\w+kernel)', mangled)
  name = mangled
  if m:      # template arguments: a bool (noise generated in-kernel) and / or the vehicle carrier (ABI 5: compile-time defaults | run-time struct)
    targs = mangled[m.end():].split('StateDev')[0].split('SplitArgs')[0]          # (what stands between the name and the first parameter)
    args = []
    if targs.startswith('ILb1E'): args.append('noise')
    elif targs.startswith('ILb0E'): args.append('no noise')
    if 'VehicleRt' in targs: args.append('VehicleRt')
    elif 'VehicleDefault' in targs: args.append('VehicleDefault')
    name = m.group(2) + ('<' + ', '.join(args) + '>' if args else '')
  out.append('| `' + name + '` | ' + ' | '.join(re.search(pat + r': (\d+)', b).group(1) for _, pat in keys) + ' |')
text = '\n'.join(out) + '\n'
open(sys.argv[1], 'w').write(text) if len(sys.argv) > 1 else None
print(text)
