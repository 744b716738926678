set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_obs
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/profiles/obs_only.py 65536"
run_pmc () { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log; }
run_pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_pmc sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64
run_pmc sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
python - <<PY
import csv, glob, collections, os
out = collections.defaultdict(list)
for p in glob.glob('$OUT/*/*_counter_collection.csv'):
  for r in csv.DictReader(open(p)):
    if 'ble_observe_kernel' in r['Kernel_Name']:
      out[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(out.items()):
  print('%-32s steady(last 8 avg) %.6g   first %.6g' % (k, sum(v[-8:]) / len(v[-8:]), v[0]))
PY
