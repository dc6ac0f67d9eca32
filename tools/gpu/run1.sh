# one gpurun call: steps continue after an ordinary failure (assertion), stop after a timeout / kill (124 / 137)
cd $GRAFT_REPO_ROOT
step() { log=$1; shift; "$@" > gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; exit $rc; fi; }
step r3_t1.log timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "tiled or ln_fold"
step r3_wlayout_b1.log timeout -k 10 300 python tools/bench_wlayout.py 2
step r3_t2.log timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_ln_fold_gpu.py tests/test_golden_gpu.py -x -q
IDB_W_TILED=0 step r3_bench_rows.log timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-kernel-roofline --no-driver-points
IDB_W_TILED=1 step r3_bench_tiled.log timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-kernel-roofline --no-driver-points
tail -n 3 gpurun_out/r3_t1.log gpurun_out/r3_t2.log gpurun_out/r3_bench_rows.log gpurun_out/r3_bench_tiled.log
