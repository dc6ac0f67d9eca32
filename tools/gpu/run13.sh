R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_patch_tests2.log timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "patch"
tail -n 12 gpurun_out/r3_patch_tests2.log
grep -q "failed\|error" gpurun_out/r3_patch_tests2.log && exit 1
IDB_CONV_PATCH_SMALL=0 IDB_COMBOS="0:0" step r3_conv2_ps0.log timeout -k 10 300 python tools/bench_conv.py 2
IDB_CONV_PATCH_SMALL=1 IDB_COMBOS="0:0" step r3_conv2_ps1.log timeout -k 10 300 python tools/bench_conv.py 2
paste <(cut -c1-70 gpurun_out/r3_conv2_ps0.log) <(cut -c55-75 gpurun_out/r3_conv2_ps1.log)
B1="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline"
IDB_CONV_PATCH_SMALL=0 step r3_b1_ps0.log timeout -k 10 300 $B1
IDB_CONV_PATCH_SMALL=1 step r3_b1_ps1.log timeout -k 10 300 $B1
for f in ps0 ps1; do echo $f $(grep -h '"value"' gpurun_out/r3_b1_$f.log | cut -c88-110); done
