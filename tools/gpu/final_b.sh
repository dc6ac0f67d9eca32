# profiles of the final build: rocprofv3 kernel stats (batch 1, batch 64), PMC passes, source hash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fin_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { log=$1; shift; "$@" > $O/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $O/$log; exit $rc; fi; }
step prof_b1.log timeout -k 10 400 rocprofv3 --kernel-trace --stats -M --output-format csv -d $O/b1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points
step prof_b64.log timeout -k 10 400 rocprofv3 --kernel-trace --stats -M --output-format csv -d $O/b64 -- python $R/bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points
find $O -name "*kernel_trace.csv" -delete
cd $R
python tools/profile_meta.py $O > $O/sha.txt 2>&1
bash tools/gpu/run_pmc.sh
find $O -name "*kernel_stats.csv" | head
