set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_ic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/profiles/obs_only.py 65536"
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_WAVE_CYCLES --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
python - <<PY
import csv, glob, collections
by = collections.defaultdict(list)
for p in glob.glob('$OUT/a/*_counter_collection.csv'):
  for r in csv.DictReader(open(p)):
    if 'ble_observe_kernel' in r['Kernel_Name']: by[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(by.items()): print('%-30s %.4g per launch (steady), %.1f per env' % (k, sum(v[-8:]) / 8, sum(v[-8:]) / 8 / 65536))
PY
