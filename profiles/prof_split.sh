#!/bin/bash
# PMC profile of the two forms of the transition at a small batch (profiles/split_ab.py <n>): instruction counts and waits per wave.
#   bash profiles/prof_split.sh [n]      -> gpurun_out/prof_split/
N=${1:-4096}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --output-format csv -d $OUT/pmc -o p -- python $ROOT/profiles/split_ab.py $N > $OUT/pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/profiles/split_ab.py $N > $OUT/trace.log 2>&1
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob('$OUT/pmc/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name'][:60]
    if 'step' not in k: continue
    key = (k, r['Grid_Size'])
    acc[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[key].add(r['Dispatch_Id'])
for key, c in sorted(acc.items()):
    n = len(cnt[key]); w = c['SQ_WAVES'] / n
    print(key, 'dispatches', n, 'waves', w)
    for name in sorted(c):
        print('   %-22s per wave per dispatch %12.1f' % (name, c[name] / n / w))
PY
