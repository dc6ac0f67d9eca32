"""`python bench.py --gpus N` with no WORLD_SIZE must start the N ranks itself (torch.distributed.run as a child process, before
any GPU call) and print ONE JSON line.  Driven here with 2 gloo ranks on the CPU and bench.py's fake pipeline
(IDB_BENCH_FAKE=1), so the launcher and the whole multi-rank branch — process group, all-gather of uint8 images, barriers,
max-over-ranks timing — execute without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"IDB_BENCH_FAKE": "1", "OMP_NUM_THREADS": "1"})
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_without_world_size_spawns_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64", "--ddpm-steps", "2", "--batch", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["unit"] == "images/s" and r["value"] > 0 and "FAKE" in r["data"]
    assert abs(r["value"] - 2 * 3 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-2 * r["value"]     # whole-job images / max-over-ranks time
    assert "identity-sharded x2" in r["config"]["parallelism"]


def test_single_rank_fake_line_and_mismatched_world_size():
    p = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--size", "64", "--ddpm-steps", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["config"]["parallelism"] == "single GPU"
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "64", "--ddpm-steps", "2"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def _torchrun(nproc, args, extra_env):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"OMP_NUM_THREADS": "1"})
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


def test_forced_collectives_at_world_size_1_on_gloo():
    """IDB_FORCE_DIST=1: the driver's launch form with ONE rank still initialises the process group and runs the all-gather, the
    barriers and the MAX all-reduce (the -m gpu twin, tests/test_rccl_gpu.py, does the same on RCCL with the real pipeline)."""
    p = _torchrun(1, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "64", "--ddpm-steps", "2", "--batch", "2"],
                  {"IDB_BENCH_FAKE": "1", "IDB_FORCE_DIST": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["config"]["forced_collectives"] == "gloo"
