cd $GRAFT_REPO_ROOT
step() { log=$1; shift; "$@" > gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 gpurun_out/$log; exit $rc; fi; }
step r3_lwprobe.log timeout -k 5 120 python tools/gpu/lw_probe.py
grep -q "probe ok" gpurun_out/r3_lwprobe.log || { echo "probe failed"; tail -n 20 gpurun_out/r3_lwprobe.log; exit 1; }
step r3_t3.log timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "loader_waves or tiled or ln_fold"
step r3_lw_b1.log timeout -k 10 400 python tools/bench_lw.py 2
step r3_parity.log timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -s -k "config1_30"
for v in 0 5 6 7; do
IDB_GEMM_LW=$v step r3_bench_lw$v.log timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-kernel-roofline --no-driver-points
done
step r3_rccl.log timeout -k 10 300 python -m pytest tests/test_rccl_gpu.py -x -q
step r3_t4.log timeout -k 10 900 python -m pytest tests/test_mtcnn_gpu.py tests/test_fp8_path_gpu.py -x -q -s
tail -n 4 gpurun_out/r3_t3.log gpurun_out/r3_t4.log gpurun_out/r3_rccl.log
grep -h '"value"' gpurun_out/r3_bench_lw*.log | cut -c1-160
