R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
IDB_COMBOS="0:0,76:1,78:2,88:2,88:4,78:4,88:8,78:8,88:16,78:16" step r3_conv_b1_big.log timeout -k 10 400 python tools/bench_conv.py 2
step r3_gn_tests.log timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_groupnorm"
IDB_GN_CONV=0 step r3_bench_gnconv0.log timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-driver-points
IDB_GN_CONV=1 step r3_bench_gnconv1.log timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-driver-points
cat gpurun_out/r3_conv_b1_big.log; tail -n 3 gpurun_out/r3_gn_tests.log
grep -h '"value"' gpurun_out/r3_bench_gnconv*.log | cut -c1-150
