R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step r3_groups.log timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -x -q -s -k "lora_groups"
step r3_full_suite.log timeout -k 10 1100 python -m pytest tests -q -m gpu
step r3_bench_default.log timeout -k 10 900 python bench.py
tail -n 8 gpurun_out/r3_groups.log; tail -n 12 gpurun_out/r3_full_suite.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_bench_default.log") if l.startswith("{")][-1])
print("value", d["value"], "config2", d["path"].get("config2", {}).get("images_per_s"), "fp8", d["path"].get("config2_fp8", {}).get("images_per_s"))
print("mixed", d["path"].get("config2_mixed")); print("e2e", d["path"].get("driver_e2e"))
PY
