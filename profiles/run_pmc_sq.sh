#!/bin/bash
# SQ-only counter passes (faster than run_pmc.sh); usage: bash profiles/run_pmc_sq.sh <outdir>
set -e
export GSPLAT_BENCH_TRAIN_STEP=0
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0   # no child processes under the profiler
export GSPLAT_NO_BUILD=1   # the profiled process has an initialised GPU: it must not spawn make / hipcc (build before)
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-workloads > $R/$OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA
run sq3 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
