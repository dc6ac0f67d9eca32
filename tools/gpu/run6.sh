R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_gnprobe.log timeout -k 5 120 python tools/gpu/gn_probe.py
grep -q "probe ok" gpurun_out/r3_gnprobe.log || { echo "probe failed"; tail -n 20 gpurun_out/r3_gnprobe.log; exit 1; }
step r3_gn_tests.log timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fused_groupnorm"
step r3_groups.log timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_graph_safety_gpu.py -x -q -s
for v in 0 1; do
IDB_GN_CONV=$v step r3_bench_gnconv$v.log timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-driver-points
done
step r3_parity2.log timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py -x -q -s
tail -n 6 gpurun_out/r3_gn_tests.log gpurun_out/r3_groups.log gpurun_out/r3_parity2.log
grep -h '"value"' gpurun_out/r3_bench_gnconv*.log | cut -c1-150
bash tools/gpu/run_pmc.sh
