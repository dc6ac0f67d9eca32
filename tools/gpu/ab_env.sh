# same-box A/B of environment switches: AB_A / AB_B are "VAR=value ..." strings, alternating three times; AB_ARGS extra bench.py arguments
R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
B1="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline ${AB_ARGS:-}"
for i in 1 2 3; do
step abe_a$i.log timeout -k 10 300 env $AB_A $B1
step abe_b$i.log timeout -k 10 300 env $AB_B $B1
done
for f in a1 b1 a2 b2 a3 b3; do echo $f $(grep -h '"value"' gpurun_out/abe_$f.log | cut -c88-110); done
