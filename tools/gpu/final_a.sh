# final-build checks: full GPU suite, smoke, default bench line, configs[4]-share line
R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step fin_suite.log timeout -k 10 700 python -m pytest tests -q -m gpu
tail -n 6 gpurun_out/fin_suite.log
step fin_smoke.log timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()"
tail -n 2 gpurun_out/fin_smoke.log
step fin_bench_default.log timeout -k 10 600 python bench.py
grep -h '"value"' gpurun_out/fin_bench_default.log > gpurun_out/fin_bench_default.json
step fin_bench_768fp8.log timeout -k 10 400 python bench.py --size 768 --vpred --batch 32 --dtype fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-config2 --no-fp8-point --no-driver-points
python - <<'PY'
import json
d = json.loads(open("gpurun_out/fin_bench_default.json").readline())
print("value", d["value"], "config2", d["path"].get("config2", {}).get("images_per_s"), "fp8", d["path"].get("config2_fp8", {}).get("images_per_s"))
print("mixed", d["path"].get("config2_mixed")); print("e2e", d["path"].get("driver_e2e")); print("roofline", {k: v for k, v in d["roofline"].items() if k != "per_tile"}); print("cpu", d["cpu_baseline"])
l = [x for x in open("gpurun_out/fin_bench_768fp8.log") if x.startswith("{")]
if l:
    e = json.loads(l[-1]); print("768 fp8 b32:", e["value"], e["ms_per_step"], e["config"])
PY
