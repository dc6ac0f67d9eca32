R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_fp8lw.log timeout -k 10 300 python -m pytest tests/test_gemm_fp8_gpu.py -x -q
step r3_full_suite.log timeout -k 10 1100 python -m pytest tests -q -m gpu -x
step r3_smoke.log timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()"
IDB_GEMM8_BIG_TILES=0 step r3_b64_fp8_small.log timeout -k 10 400 python bench.py --batch 64 --dtype fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-config2 --no-driver-points
step r3_b64_fp8_big.log timeout -k 10 400 python bench.py --batch 64 --dtype fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-config2 --no-driver-points
cd /tmp && export TMPDIR=/tmp
step r3_prof_b1.log timeout -k 10 600 rocprofv3 --kernel-trace --stats -M --output-format csv -d $R/gpurun_out/prof_b1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points
step r3_prof_b64.log timeout -k 10 600 rocprofv3 --kernel-trace --stats -M --output-format csv -d $R/gpurun_out/prof_b64 -- python $R/bench.py --steps 1 --warmup 1 --batch 64 --no-cpu-baseline --no-kernel-roofline --no-config2 --no-fp8-point --no-driver-points
cd $R
step r3_bench_default.log timeout -k 10 900 python bench.py
find gpurun_out/prof_b1 gpurun_out/prof_b64 -name "*kernel_stats.csv"
tail -n 6 gpurun_out/r3_full_suite.log; tail -n 2 gpurun_out/r3_smoke.log gpurun_out/r3_fp8lw.log
grep -h '"value"' gpurun_out/r3_b64_fp8_*.log | cut -c1-150
