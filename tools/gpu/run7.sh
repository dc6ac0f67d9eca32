R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_gn_tests.log timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fused_groupnorm"
step r3_gnfuse_b1.log timeout -k 10 400 python tools/bench_gnfuse.py 2
tail -n 5 gpurun_out/r3_gn_tests.log; cat gpurun_out/r3_gnfuse_b1.log
