# images/s of the sampler over the per-GPU batch (f16, 512x512, 30 steps) on one box
R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
for b in 1 2 4 8 16 32 64; do
step sweep_b$b.log timeout -k 10 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points --no-kernel-roofline
done
for b in 1 2 4 8 16 32 64; do grep -h '^{' gpurun_out/sweep_b$b.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d['path']
print('batch', d['config']['batch_per_gpu'], 'images/s', d['value'], 'ms/step', d['ms_per_step'], 'frac_of_mfma_peak', p.get('frac_of_mfma_peak'), 'arena_mib', p.get('arena_mib'))
"; done
