# Per-phase PMC counts of the observation kernel.  Usage (GPU box): bash profiles/prof_obs_phases.sh
# needs build_ab/libble_phase.so = bash profiles/build_variant.sh phase '-DBLE_OBS_INSTR_HEADER="../../profiles/instr/ble_observe_instr.h"' -DBLE_OBS_PHASE_PROFILE
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_phases
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BLE_HIP_LIB=$ROOT/build_ab/libble_phase.so
CMD="python $ROOT/profiles/obs_phases.py 65536"
run_pmc () { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log; }
run_pmc a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA
run_pmc c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc b SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT
python - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for p in sorted(glob.glob('$OUT/*/*_counter_collection.csv')):
  rows = [r for r in csv.DictReader(open(p)) if 'ble_observe_kernel' in r['Kernel_Name']]
  by = collections.defaultdict(list)
  for r in rows: by[r['Counter_Name']].append(float(r['Counter_Value']))
  for k, v in by.items(): per[k] = v[-10:]
names = ['phase 0 (-> B1)', 'phase 1 (-> B3)', 'block inverses', 'sweep + features + store']
print('%-26s' % 'per environment' + ''.join('%28s' % x for x in names) + '%12s' % 'total')
for k, v in sorted(per.items()):
  g = [sum(v[3 * i:3 * i + 3]) / 3 / 65536 for i in range(3)] + [v[9] / 65536]      # stop 1, 2, 3, complete (cumulative)
  d = [g[0], g[1] - g[0], g[2] - g[1], g[3] - g[2]]
  print('%-26s' % k + ''.join('%28.1f' % x for x in d) + '%12.1f' % g[3])
PY
