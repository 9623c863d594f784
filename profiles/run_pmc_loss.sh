#!/bin/bash
# PMC counters of the loss kernels (tools/loss_timing.py), separate passes, never combined with tracing.
# usage (on the GPU box, from the repo root): bash profiles/run_pmc_loss.sh <outdir> [HxW]
set -e
OUT=${1:-gpurun_out/pmc_loss}; SHAPE=${2:-1080x1920}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/loss_timing.py $SHAPE > $R/$OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R && python3 profiles/summarize_pmc.py $OUT > $OUT/summary.json
