#!/bin/bash
# Collects per-kernel PMC counters for the bench workload in separate passes (never combined with tracing).
# usage (on the GPU box, from the repo root): bash profiles/run_pmc.sh <outdir>
set -e
export GSPLAT_BENCH_TRAIN_STEP=0
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0   # no child processes under the profiler
export GSPLAT_NO_BUILD=1   # the profiled process has an initialised GPU: it must not spawn make / hipcc (build before)
OUT=${1:-gpurun_out/pmc}
WORKLOAD=${2:-}   # r05: empty = the bench command (headline scene); else tools/workload_stats.py <workload> (e.g. garden1200k)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  local name=$1; shift
  if [ -z "$WORKLOAD" ]; then
    rocprofv3 --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-workloads > $R/$OUT/$name.log 2>&1 || echo "pass $name failed"
  else
    rocprofv3 --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/workload_stats.py $WORKLOAD 6 > $R/$OUT/$name.log 2>&1 || echo "pass $name failed"
  fi
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_EA0_ATOMIC_sum
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
