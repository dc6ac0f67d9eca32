# the bench contract test and the default bench line of the build
R=$GRAFT_REPO_ROOT
cd $R
step() { log=$1; shift; "$@" > $R/gpurun_out/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $R/gpurun_out/$log; exit $rc; fi; }
step bd_contract.log timeout -k 10 400 python -m pytest tests/test_bench_contract_gpu.py tests/test_rccl_gpu.py -q -x
tail -n 3 gpurun_out/bd_contract.log
step bd_bench.log timeout -k 10 600 python bench.py
grep -h '^{' gpurun_out/bd_bench.log > gpurun_out/bd_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bd_bench.json").readline()); r = d["roofline"]; p = d["path"]
print("value", d["value"], "config2", p["config2"]["images_per_s"], "fp8", p["config2_fp8"]["images_per_s"], "mixed", p["config2_mixed"]["images_per_s"], "e2e", p["driver_e2e"]["images_per_s"])
print("roofline", r["kernel"][:70], r["achieved"], r["frac"], r["avg_launch_us"], r["traffic"], r.get("profiled"))
print("next", [(x["kernel"][:40], x["frac"], x["share_of_gemm_time"]) for x in r["next_by_time"]])
PY
