#!/bin/bash
# r06: SQ counters of the per-gaussian forward kernels (split form, lean), one rocprofv3 --pmc pass per counter group
cd $GRAFT_REPO_ROOT
python -c "import importlib; importlib.import_module('3dgs_amd._lib').build()"
export GSPLAT_NO_BUILD=1 GSPLAT_PRE_SPLIT=${GSPLAT_PRE_SPLIT:-1}
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/r06_pre_pmc${1:-}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/workload_stats.py config3 6 > $R/$OUT/$name.log 2>&1 || echo "pass $name failed"; echo "pass $name done"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA
run sq3 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq4 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_MOPS_F32
cd $R
python3 profiles/summarize_pmc.py $OUT > $OUT/summary.json 2> $OUT/summary.err || true
python3 - $OUT/summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if any(s in k for s in ("preprocess_geom","sh_colour","preprocess_kernel","bin_scatter")):
        print(k[:70]); print("   ", {a:round(b) for a,b in v.items()})
PY
