R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-driver-points"
IDB_W_PREFETCH=0 step r3_pf_off.log timeout -k 10 300 $B
IDB_W_PREFETCH=1 IDB_W_PREFETCH_WGS=64 step r3_pf_1_64.log timeout -k 10 300 $B
IDB_W_PREFETCH=4 IDB_W_PREFETCH_WGS=64 step r3_pf_4_64.log timeout -k 10 300 $B
IDB_W_PREFETCH=1 IDB_W_PREFETCH_WGS=16 step r3_pf_1_16.log timeout -k 10 300 $B
IDB_W_PREFETCH=0.1 IDB_W_PREFETCH_WGS=32 step r3_pf_01_32.log timeout -k 10 300 $B
IDB_W_PREFETCH=0 step r3_pf_off2.log timeout -k 10 300 $B
for f in off 1_64 4_64 1_16 01_32 off2; do echo $f; grep -h '"value"' gpurun_out/r3_pf_$f.log | cut -c1-120 || tail -n 5 gpurun_out/r3_pf_$f.log; done
