# two more default bench lines of the build on whatever box this call gets (box-to-box spread of the headline)
R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step fin_bench_x1.log timeout -k 10 500 python bench.py --no-cpu-baseline
step fin_bench_x2.log timeout -k 10 500 python bench.py --no-cpu-baseline
grep -h '^{' gpurun_out/fin_bench_x1.log gpurun_out/fin_bench_x2.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); p = d['path']
    print(d['value'], p['config2']['images_per_s'], p['config2_fp8']['images_per_s'], p['config2_mixed']['images_per_s'], p['driver_e2e']['images_per_s'], d['roofline']['frac'], d['roofline'].get('profiled', {}).get('avg_launch_us'))
"
