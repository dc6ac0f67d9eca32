"""The bench.py output contract (one JSON line on stdout with the fields the driver and the judge read), exercised on the
reduced-width graph so that it runs in seconds: `python bench.py --tiny --steps 1 --warmup 1`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--steps", "1", "--warmup", "1", "--size", "128",
                        "--ddpm-steps", "3"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly ONE line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 0.05
    assert set(d["path"]["stages"]) >= {"sampling_loop_ms", "vae_decode_postprocess_ms", "d2h_uint8_ms"}
    r = d["roofline"]
    # achieved / frac / avg_launch_us: this run's live HIP-event figures; "profiled": the committed rocprofv3 summary (or null / stale)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "source", "profiled", "launches_per_cfg_forward_all_kernels"):
        assert k in r, k
    assert r["launches_per_cfg_forward_all_kernels"] > 100
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
