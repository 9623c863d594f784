cd $GRAFT_REPO_ROOT
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0 GSPLAT_BWD_PINGPONG=1
bash tools/ab/run.sh 300 ppd0 ppd2 ppd3
echo "== stamps, duty priority 2"
GSPLAT_LIB=tools/ab/libstamp.so timeout -k 10 300 python tools/bwd_timeline.py config3 2>&1 | grep -v "^waves alive\|XCC"
