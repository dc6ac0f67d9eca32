R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gr_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { log=$1; shift; "$@" > $O/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $O/$log; exit $rc; fi; }
export IDB_GN_CONV_RESNET=0
step g0.log timeout -k 10 300 rocprofv3 --kernel-trace --stats -M --output-format csv -d $O/g0 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline
export IDB_GN_CONV_RESNET=1
step g1.log timeout -k 10 300 rocprofv3 --kernel-trace --stats -M --output-format csv -d $O/g1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline
find $O -name "*kernel_trace.csv" -delete
grep -h '"value"' $O/g0.log $O/g1.log | cut -c88-110
