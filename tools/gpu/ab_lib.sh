# same-box A/B of two builds of the library: faceposegenerator_amd/libidb_kernels_base.so (IDB_LIB) vs libidb_kernels.so, alternating
R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
B1="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline ${AB_ARGS:-}"
BASE=$R/faceposegenerator_amd/libidb_kernels_base.so
for i in 1 2 3; do
IDB_LIB=$BASE step ab_base$i.log timeout -k 10 300 $B1
step ab_new$i.log timeout -k 10 300 $B1
done
for f in base1 new1 base2 new2 base3 new3; do echo $f $(grep -h '"value"' gpurun_out/ab_$f.log | cut -c88-110); done
