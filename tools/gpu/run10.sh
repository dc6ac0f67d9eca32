R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_patch_tests.log timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "patch"
tail -n 15 gpurun_out/r3_patch_tests.log
grep -q "failed\|error" gpurun_out/r3_patch_tests.log && exit 1
step r3_conv128_patch.log timeout -k 10 500 python tools/bench_conv.py 128
cat gpurun_out/r3_conv128_patch.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64 --no-config2 --no-fp8-point --no-driver-points"
IDB_CONV_PATCH=0 step r3_b64_patch0.log timeout -k 10 300 $B
IDB_CONV_PATCH=1 step r3_b64_patch1.log timeout -k 10 300 $B
grep -h '"value"' gpurun_out/r3_b64_patch0.log gpurun_out/r3_b64_patch1.log | cut -c1-160
