R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
B1="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline"
B64="python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline"
IDB_CONV_PATCH_SMALL=0 step r3_b1_s0.log timeout -k 10 300 $B1
IDB_CONV_PATCH_SMALL=1 IDB_CONV_PATCH_SHORTCUT=0 step r3_b1_s1c0.log timeout -k 10 300 $B1
IDB_CONV_PATCH_SMALL=1 IDB_CONV_PATCH_SHORTCUT=1 step r3_b1_s1c1.log timeout -k 10 300 $B1
IDB_CONV_PATCH_SMALL=0 step r3_b1_s0b.log timeout -k 10 300 $B1
IDB_CONV_PATCH=0 step r3_b64_p0.log timeout -k 10 300 $B64
IDB_CONV_PATCH_SHORTCUT=0 step r3_b64_p1c0.log timeout -k 10 300 $B64
IDB_CONV_PATCH_SHORTCUT=1 step r3_b64_p1c1.log timeout -k 10 300 $B64
for f in b1_s0 b1_s1c0 b1_s1c1 b1_s0b b64_p0 b64_p1c0 b64_p1c1; do echo $f $(grep -h '"value"' gpurun_out/r3_$f.log | cut -c88-110); done
