cd $GRAFT_REPO_ROOT
step() { log=$1; shift; "$@" > gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 gpurun_out/$log; exit $rc; fi; }
step r3_t5.log timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "256_row or loader_waves"
step r3_conv128.log timeout -k 10 400 python tools/bench_conv.py 128
step r3_proj128.log timeout -k 10 300 python tools/bench_proj.py 128
for v in 0 512 256; do
IDB_GEMM_BIG_TILES=$v step r3_b64_big$v.log timeout -k 10 400 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-config2 --no-driver-points
done
step r3_t6.log timeout -k 10 900 python -m pytest tests/test_mtcnn_gpu.py tests/test_fp8_path_gpu.py -q -s
step r3_t7.log timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -q -s -k "batch64_f16"
cd /tmp && export TMPDIR=/tmp
step r3_prof_b1.log timeout -k 10 600 rocprofv3 --kernel-trace --stats -M --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_b1 -name "*kernel_stats.csv" | head -3
tail -n 4 gpurun_out/r3_t5.log gpurun_out/r3_t6.log gpurun_out/r3_t7.log
grep -h '"value"' gpurun_out/r3_b64_big*.log | cut -c1-150
