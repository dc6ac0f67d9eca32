"""The N > 1 plumbing of bench.py and driver.all_gather_images on RCCL, exercised on the one-GPU box: bench.py is started the way
the driver starts it for N > 1 (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`, a CHILD process, so
nothing that has initialised the GPU is replaced) with IDB_FORCE_DIST=1, which makes the single rank run
`init_process_group("nccl", device_id=...)`, the device-tensor `all_gather_into_tensor` of the decoded uint8 images, the barriers,
the MAX all-reduce of the timing and driver.all_gather_images(force=True).  No scaling claim: the first 8-GPU run must not fail
on plumbing."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_one_rank_runs_the_rccl_collectives():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"IDB_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tiny",
           "--size", "128", "--ddpm-steps", "3", "--batch", "2", "--no-cpu-baseline", "--no-kernel-roofline", "--no-config2",
           "--no-driver-points"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["config"]["forced_collectives"] == "nccl"
