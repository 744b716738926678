# Where do the step kernel's parked cycles come from?  Instruction-fetch and wait counters of ble_step_kernel
# (32-step launches of 65 536 environments).  Usage (GPU box): bash profiles/prof_step_waits.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_step_waits
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 192 --warmup 32 --reps 3 --no-extras"
run () { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_WAVE_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_IFETCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS SQ_WAVE_CYCLES SQ_WAVES
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
python - <<PY
import csv, glob, collections
by = collections.defaultdict(list)
for p in sorted(glob.glob('$OUT/*/*_counter_collection.csv')):
  for r in csv.DictReader(open(p)):
    if 'ble_step_kernel' in r['Kernel_Name']: by[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(by.items()):
  w = v[-18:]
  print('%-30s %12.4g per 32-step launch   %10.1f per wave-step' % (k, sum(w) / len(w), sum(w) / len(w) / 1024 / 32))
PY
