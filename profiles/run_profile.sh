#!/bin/bash
# Collects the rocprofv3 evidence for one round.  Run on the GPU box from the repo root:
#   bash profiles/run_profile.sh r01
# Kernel trace (+stats) and PMC counters are collected in SEPARATE runs; PMC passes never
# combine with sys/hip/hsa/memory-copy tracing.  Raw output goes to gpurun_out/ (scratch);
# profiles/summarize.py turns it into the committed profiles/<tag>_*.json / .md files.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 192 --warmup 32 --no-cpu-baseline"
if [ "${BLE_PROFILE_ONLY_OBS:-0}" != "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
run_pmc () {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/bench.py --steps 64 --warmup 32 --no-cpu-baseline > $OUT/$name.log 2>&1
}
run_pmc pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run_pmc pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH
run_pmc pmc_f64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32
run_pmc pmc_misc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VSKIPPED
run_pmc pmc_fetch FETCH_SIZE TCC_MISS_sum
run_pmc pmc_write WRITE_SIZE TCC_HIT_sum
fi
# observation kernel: trace of the --observe leg and two PMC passes
OBS="python $ROOT/bench.py --steps 32 --warmup 32 --observe 8 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_obs -o trace_obs -- $OBS > $OUT/trace_obs.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_obs1 -o pmc_obs1 -- $OBS > $OUT/pmc_obs1.log 2>&1
if [ "${BLE_PROFILE_OBS_PMC:-0}" = "1" ]; then   # this counter set aborted rocprofv3 on the first try (r01); opt-in, bounded
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/pmc_obs2 -o pmc_obs2 -- $OBS > $OUT/pmc_obs2.log 2>&1
fi
find $OUT -name "*.csv" | head -60
