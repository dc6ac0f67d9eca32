R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_gnpatch_tests.log timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "patch or fused_groupnorm"
tail -n 15 gpurun_out/r3_gnpatch_tests.log
grep -q "failed\|error" gpurun_out/r3_gnpatch_tests.log && exit 1
B1="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline"
IDB_GN_CONV_RESNET=0 step r3_b1_gr0.log timeout -k 10 300 $B1
IDB_GN_CONV_RESNET=1 step r3_b1_gr1.log timeout -k 10 300 $B1
IDB_GN_CONV_RESNET=0 step r3_b1_gr0b.log timeout -k 10 300 $B1
IDB_GN_CONV_RESNET=1 step r3_b1_gr1b.log timeout -k 10 300 $B1
for f in gr0 gr1 gr0b gr1b; do echo $f $(grep -h '"value"' gpurun_out/r3_b1_$f.log | cut -c88-110); done
step r3_parity_gr1.log timeout -k 10 500 env IDB_GN_CONV_RESNET=1 python -m pytest tests/test_parity_gpu.py tests/test_engine_gpu.py -q -x
tail -n 5 gpurun_out/r3_parity_gr1.log
