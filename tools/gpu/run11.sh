R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_rot_tests.log timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_ln_fold_gpu.py -q -x
tail -n 5 gpurun_out/r3_rot_tests.log
grep -q "failed\|error" gpurun_out/r3_rot_tests.log && exit 1
step r3_conv128_rot.log timeout -k 10 500 python tools/bench_conv.py 128
cat gpurun_out/r3_conv128_rot.log
IDB_COMBOS="0:0" step r3_conv2_rot.log timeout -k 10 300 python tools/bench_conv.py 2
cat gpurun_out/r3_conv2_rot.log
step r3_bench_rot.log timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-driver-points
grep -h '"value"' gpurun_out/r3_bench_rot.log | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('b1', r['value'], 'config2', r['path'].get('config2_batch64', r['path']).get('images_per_s') if isinstance(r.get('path'), dict) else None)
print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if 'images_per_s' in kk or kk in ('mfma_frac',)}) for k, v in r.get('path', {}).items()})
print(r['roofline'])
"
