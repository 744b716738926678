#!/bin/bash
# Collects the rocprofv3 evidence for one round.  Run on the GPU box from the repo root:
#   bash profiles/run_profile.sh r02
# Kernel trace (+stats) and PMC counters are collected in SEPARATE runs; PMC passes never
# combine with sys/hip/hsa/memory-copy tracing.  Raw output goes to gpurun_out/ (scratch);
# profiles/summarize.py turns it into the committed profiles/<tag>_*.json / .md files.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# ---- transition kernel: the headline leg of bench.py alone (5 repetitions of 192 steps = 30 launches of 32 steps)
BENCH="python $ROOT/bench.py --steps 192 --warmup 32 --reps 5 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
# ---- the driver's shape: python bench.py --gpus 1 --steps 20 --warmup 5 (one 20-step launch per timed region), headline leg alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace20 -o trace20 -- python $ROOT/bench.py --steps 20 --warmup 5 --reps 9 --no-extras > $OUT/trace20.log 2>&1
# ---- one launch per agent step (policy in the loop): 300 launches of ble_step_f32
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o trace1 -- python $ROOT/profiles/step_single.py 65536 > $OUT/trace1.log 2>&1
run_pmc () {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $BENCH > $OUT/$name.log 2>&1
}
run_pmc pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run_pmc pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH
run_pmc pmc_f64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32
run_pmc pmc_misc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VSKIPPED
run_pmc pmc_fetch FETCH_SIZE TCC_MISS_sum
run_pmc pmc_write WRITE_SIZE TCC_HIT_sum
# ---- observation kernel: 121 window-filling + 8 steady-state launches at 65 536 envs
OBS="python $ROOT/profiles/obs_only.py 65536"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_obs -o trace_obs -- $OBS > $OUT/trace_obs.log 2>&1
run_obs () { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $OBS > $OUT/$name.log 2>&1; }
run_obs pmc_obs1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_obs pmc_obs2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64
run_obs pmc_obs3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_obs pmc_obs_fetch FETCH_SIZE
run_obs pmc_obs_write WRITE_SIZE
# ---- the small-batch form of the transition (one environment on four wavefronts) against the one-lane kernel at one GPU's
# share of configs[3]: 16 fused 32-step launches of each at 8 192 environments
SPLIT="python $ROOT/profiles/split_launches.py 8192 16"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_split -o trace_split -- $SPLIT > $OUT/trace_split.log 2>&1
run_split () { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $SPLIT > $OUT/$name.log 2>&1; }
run_split pmc_split1 SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_LDS
run_split pmc_split2 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY
find $OUT -name "*.csv" | wc -l
