# texture-addresser / L1 counters of the streaming kernels (two per pass: more "exceeds the capabilities of the hardware"
# and the profiler aborts); every pass under its own timeout
export GSPLAT_BENCH_TRAIN_STEP=0 GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0 GSPLAT_NO_BUILD=1
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/pmc_ta; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-workloads > $R/$OUT/$name.log 2>&1 || echo "pass $name failed"; echo "pass $name done"; }
run t1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run t2 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
run t3 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
run t4 TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run t5 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run t6 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run t7 TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum
cd $R
python3 profiles/summarize_pmc.py $OUT > $OUT/summary.json 2> $OUT/summary.err || true
