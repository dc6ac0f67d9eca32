R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
B1="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline"
B64="python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline"
BASE=$R/faceposegenerator_amd/libidb_kernels_base.so
for i in 1 2; do
IDB_LIB=$BASE step r3_ab_b1_base$i.log timeout -k 10 300 $B1
step r3_ab_b1_new$i.log timeout -k 10 300 $B1
done
IDB_LIB=$BASE step r3_ab_b64_base.log timeout -k 10 300 $B64
step r3_ab_b64_new.log timeout -k 10 300 $B64
IDB_CONV_PATCH=0 step r3_ab_b64_new_nopatch.log timeout -k 10 300 $B64
for f in b1_base1 b1_new1 b1_base2 b1_new2 b64_base b64_new b64_new_nopatch; do echo $f $(grep -h '"value"' gpurun_out/r3_ab_$f.log | cut -c88-110); done
IDB_LIB=$BASE IDB_COMBOS="0:0" step r3_conv2_base.log timeout -k 10 300 python tools/bench_conv.py 2
IDB_COMBOS="0:0" step r3_conv2_new.log timeout -k 10 300 python tools/bench_conv.py 2
paste <(cut -c1-70 gpurun_out/r3_conv2_base.log) <(cut -c55-75 gpurun_out/r3_conv2_new.log)
