# (old = build_ab/libble_old.so, built from the previous commit's csrc into build_ab by hand; new = the in-tree library)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03f/ab; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in old new; do
  if [ $lib = old ]; then export BLE_HIP_LIB=$ROOT/build_ab/libble_old.so; else unset BLE_HIP_LIB; fi
  for pass in a b; do
    if [ $pass = a ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; else C="SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; fi
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${lib}_$pass -o x -- python $ROOT/profiles/step_single.py > $OUT/${lib}_$pass.log 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${lib}_t -o x -- python $ROOT/profiles/step_single.py > $OUT/${lib}_t.log 2>&1
done
python - <<PY
import csv, glob, collections
for lib in ('old', 'new'):
  by = collections.defaultdict(list)
  for p in sorted(glob.glob('$OUT/%s_[ab]/*/*counter_collection.csv' % lib) + glob.glob('$OUT/%s_[ab]/*counter_collection.csv' % lib)):
    for r in csv.DictReader(open(p)):
      if 'ble_step_kernel' in r['Kernel_Name']: by[r['Counter_Name']].append(float(r['Counter_Value']))
  print(lib, {k: round(sum(v[-200:]) / len(v[-200:]) / 1024, 1) for k, v in sorted(by.items())}, '(per wave, 1-step launch)')
  for p in glob.glob('$OUT/%s_t/*/*kernel_stats.csv' % lib) + glob.glob('$OUT/%s_t/*kernel_stats.csv' % lib):
    for r in csv.DictReader(open(p)):
      if 'ble_step_kernel' in r['Name']: print(lib, 'kernel avg ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'], 'calls', r['Calls'])
PY
