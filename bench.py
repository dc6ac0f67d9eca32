#!/usr/bin/env python3
"""Headline benchmark: 512x512 images/s, SD-2.1-base graph, 30-step DDPM, CFG 5.0, rank-4 LoRA
(BASELINE.json metric; the loop of /root/reference/inference_ID-Booth.py:138).

One "step" = one pass of the hot path over one batch: prompt embeddings in HBM -> 30 x (CFG UNet forward +
DDPM step) -> VAE decode -> uint8 images in HBM (+ one RCCL all-gather of the decoded images when N > 1).
Default workload = BASELINE configs[1] (batch 1 per GPU); `--batch 64` is configs[2], the throughput point.
Weights are seeded synthetic tensors of the published SD-2.1-base shapes (no network for checkpoints).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no WORLD_SIZE in the environment: starts the N ranks itself through
                                         torch.distributed.run, as a child process, before any GPU call)

The default operand dtype is f16, the reference's own (`torch_dtype=torch.float16`, inference_ID-Booth.py:103): same MFMA
rate as bf16 and 8x closer to the fp32 oracle; `--dtype bf16` selects bf16.
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_MFMA_TFLOPS = 2500.0      # dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
_TILE_SHAPES = {1: "128x160", 2: "128x128", 3: "64x160", 4: "64x64,8w", 5: "128x32",
                6: "64x160,8w", 7: "64x128,8w", 8: "128x160,8w", 9: "128x128,8w"}
_TILE_VARIANTS = {0: "idb_gemm_kernel<{},ring2>", 1: "idb_gemm_kernel<{},ring3>", 2: "idb_gemm_kernel<{},ring4>",
                  3: "idb_gemm_kernel_rs<{}>", 4: "idb_gemm_kernel_pl<{}>", 5: "idb_gemm_kernel_lw<{},4 loader waves,ring3>",
                  6: "idb_gemm_kernel_lw<{},8 loader waves,ring3>", 7: "idb_gemm_kernel_lw<{},4 loader waves,ring4>",
                  8: "idb_gemm_kernel_lw<{},4 loader waves,ring3>", 9: "idb_conv_patch_kernel<{},halo patch resident,2+2 loader waves>",
                  10: "idb_conv_patch_kernel<{},halo patch resident,2+2 loader waves,ring4>"}


class _TileNames(dict):
    """idb_gemm_plan tile id -> kernel instance name (id = shape + 10 * variant, see idb_gemm.hip)."""
    def __missing__(self, t):
        gn, t = t >= 1000, t % 1000                        # + 1000: the launch carried a fused GroupNorm (engine launch log)
        shape = _TILE_SHAPES[t % 10]
        if t // 10 in (8, 9):                              # the 256-row loader-wave tiles
            shape = shape.replace("128x", "256x", 1)
        name = _TILE_VARIANTS[t // 10].format(shape)
        if gn:
            name = (name.replace("2+2 loader waves", "2 weight-loader + 4 transforming patch-loader waves: GroupNorm+SiLU fused") if t // 10 == 10
                    else name.replace("idb_gemm_kernel_lw<", "idb_gemm_kernel_gn<").replace("4 loader waves", "4 loader + 8 normalizer waves: GroupNorm fused"))
        return name


TILE_NAMES = _TileNames()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (1 = BASELINE configs[1], 64 = configs[2])")
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "fp8"],
                    help="operand dtype; fp8 = e4m3 MFMA path for the UNet's ResnetBlock2D 3x3 convs, f16 elsewhere (BASELINE configs[4])")
    ap.add_argument("--vpred", action="store_true", help="v-prediction scheduler mode (the 768x768 SD-2.1 checkpoint of BASELINE configs[4])")
    ap.add_argument("--ddpm-steps", type=int, default=30)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="reduced-width graph (plumbing check only; not a benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--vae-chunk", type=int, default=4)
    ap.add_argument("--no-config2", action="store_true", help="skip the short batch-64 (BASELINE configs[2]) timing of the default run")
    ap.add_argument("--no-fp8-point", action="store_true", help="skip the batch-64 timing of the fp8 (e4m3 resnet convs) engine in the default run")
    ap.add_argument("--no-driver-points", action="store_true", help="skip path.config2_mixed / path.driver_e2e (driver.generate with LoRA switches, text encoder, PNG sink)")
    ap.add_argument("--cpu-baseline-threads", type=int, default=0, help="0 = all cores of the host (default); e.g. 8 for the build container's figure")
    return ap.parse_args()


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE: start the N ranks as a CHILD process tree (torch.distributed.run, one
    process per GPU, rendezvous on 127.0.0.1) and pass its exit code on.  Nothing in this process has touched the GPU yet —
    a process that has initialised HIP must never be replaced by another program on this pool."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(pipe, ucfg, vcfg, lora_raw, ddpm_steps, size, threads=0):
    """The oracle (CPU fp32 restatement) timed on the host cores on a bounded sample: 2 CFG UNet forwards (B_eff=2)
    and one VAE decode of the same graph and weights; extrapolated to one image = ddpm_steps forwards + 1 decode."""
    from oracle import sd21_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))          # the CPUs this process may actually run on
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, torch.get_num_threads()))
    if threads > 0:
        cores = min(cores, threads)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)
    lat = size // 8
    x = torch.randn(2, 4, lat, lat, generator=g)
    ctx = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g)
    lora = O.normalize_lora_keys(lora_raw)
    merged = O.merge_lora(pipe._unet_sd, lora)          # same arithmetic, fewer tiny GEMMs: favours the CPU
    with torch.no_grad():
        t0 = time.perf_counter()
        n_fw = 2
        for _ in range(n_fw):
            O.unet_forward(merged, ucfg, x, 958, ctx)
        t_unet = (time.perf_counter() - t0) / n_fw
        t0 = time.perf_counter()
        O.vae_decode(pipe._vae_sd, vcfg, x[:1])
        t_vae = time.perf_counter() - t0
    per_image = ddpm_steps * t_unet + t_vae
    res = {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{n_fw} CFG UNet forwards ({t_unet:.2f} s each) + 1 VAE decode ({t_vae:.2f} s) of the full graph, "
                     f"extrapolated to {ddpm_steps} steps + decode"}
    if threads == 0 and cores > 8:
        # the 8-thread figure as well, for comparability with the 8-vCPU build container (one forward: bounded)
        torch.set_num_threads(8)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.unet_forward(merged, ucfg, x, 958, ctx)
            t8 = time.perf_counter() - t0
        torch.set_num_threads(cores)
        res["threads8"] = {"value": 1.0 / (ddpm_steps * t8 + t_vae * t8 / t_unet), "unit": "images/s", "cores": 8,
                           "sample": f"1 CFG UNet forward on 8 threads ({t8:.2f} s); VAE decode scaled by the same thread ratio"}
    return res


def kernel_roofline(eng, batch, lat_side, n_ctx):
    """Device duration of every implicit-GEMM launch of ONE CFG UNet forward, measured with HIP events on the launch
    stream: an eager forward records each launch's descriptor; the recorded sequence is then replayed IN ORDER, REPS times,
    with an event pair around every launch (in order, so that each launch finds its weights as cold as in the sampling
    loop: a forward reads 1.7 GB of weights, far more than L2 + MALL hold; re-launching one descriptor back-to-back would
    time it with warm weights).  The split-K reduce launch is skipped by the descriptor's profiling flag so that the
    figure is the idb_gemm_kernel instance alone, as rocprofv3 lists it.  The duration of an EMPTY event pair (marker
    dispatch, measured here too) is subtracted.  The tile configuration with the largest summed duration is the dominant
    kernel."""
    import ctypes as C
    from faceposegenerator_amd import _lib as L
    REPS = 5
    rep = 2
    B = batch * rep
    lat = torch.randn(batch, 4, lat_side, lat_side, device=eng.device)
    ctx = torch.randn(B * n_ctx, eng.ucfg.cross_attention_dim, device=eng.device).to(eng.tdt)
    ts = torch.tensor([958.0], device=eng.device)
    tp = eng.time_tables(ts)
    kv = eng.cross_kv(ctx, B, n_ctx)
    eng.arena.reset()
    eng._pinned.clear()
    eng.launch_log = []
    eng.unet_nhwc(lat, rep, (tp, 0, 0), kv, n_ctx)
    torch.cuda.synchronize()
    log, eng.launch_log = eng.launch_log, None
    st = torch.cuda.current_stream().cuda_stream

    def pair():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    for e in log:
        e["desc"].flags = 1
        e["ev"] = []
        e["fn"] = eng.lib.idb_gemm
    empty = []
    for r in range(REPS + 1):                                            # pass 0 = warm-up, untimed
        for e in log:
            ws, need = e["ws"]
            ev0, ev1 = pair()
            ev0.record()
            L.check(e["fn"](C.byref(e["desc"]), None if ws is None else ws.data_ptr(), need, st))
            ev1.record()
            if r:
                e["ev"].append((ev0, ev1))
        for _ in range(8):
            ev0, ev1 = pair()
            ev0.record()
            ev1.record()
            if r:
                empty.append((ev0, ev1))
    torch.cuda.synchronize()
    overhead_ms = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
    agg = {}
    for e in log:
        e["ms"] = max(sum(a.elapsed_time(b) for a, b in e["ev"]) / REPS - overhead_ms, 1e-4)
        a = agg.setdefault(e["tile"], {"flops": 0.0, "ms": 0.0, "n": 0, "bytes": 0.0})
        a["flops"] += e["flops"]
        a["bytes"] += e["bytes"]
        a["ms"] += e["ms"]
        a["n"] += 1
    if os.environ.get("IDB_DUMP_GEMM"):
        shapes = {}
        for e in log:
            k = (e["m"], e["n"], e["k"], e["tile"], e["split_k"], e["blocks"])
            a = shapes.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e["ms"]
            a[2] += e["flops"]
        for k, a in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
            print(f"  gemm m={k[0]:7d} n={k[1]:5d} k={k[2]:6d} tile={k[3]:2d} splitk={k[4]:2d} blocks={k[5]:5d} x{a[0]:3d} "
                  f"total {a[1]:8.3f} ms  {a[2] / (a[1] * 1e-3) / 1e12:7.1f} TF/s", file=sys.stderr)
    # dominant kernel instance: the most GEMM time; instances within 15 % of that time are a tie (three instances share 16-17 % each at
    # batch 1 and trade places from run to run), broken by the arithmetic they carry
    t_max = max(v["ms"] for v in agg.values())
    dom = max((t for t in agg if agg[t]["ms"] >= 0.85 * t_max), key=lambda t: agg[t]["flops"])
    a = agg[dom]
    achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
    total_ms = sum(v["ms"] for v in agg.values())
    total_fl = sum(v["flops"] for v in agg.values())
    detail = {TILE_NAMES[t]: {"launches": v["n"], "avg_launch_us": round(v["ms"] * 1e3 / v["n"], 2),
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for t, v in sorted(agg.items())}
    return {"bound": "mfma", "kernel": TILE_NAMES[dom], "achieved": round(achieved, 1), "peak": PEAK_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_MFMA_TFLOPS, 4), "traffic": None,
            "algorithmic_bytes_per_launch_avg": round(a["bytes"] / a["n"], 1),
            "launches_per_forward": a["n"], "avg_launch_us": round(a["ms"] * 1e3 / a["n"], 2),
            "flops_per_launch_avg": round(a["flops"] / a["n"], 1), "event_pair_overhead_us": round(overhead_ms * 1e3, 2),
            "all_gemm_tflops": round(total_fl / (total_ms * 1e-3) / 1e12, 1), "launches_per_cfg_forward_all_kernels": eng.last_forward_launches,
            # the next two kernels by total time, in the same units (the dominant one is a kernel INSTANCE: when launches move to another
            # instance — e.g. the resnet convs that now run as the fused-GroupNorm patch conv — the mix behind `achieved` changes with them)
            "next_by_time": [{"kernel": TILE_NAMES[t], "launches": agg[t]["n"], "share_of_gemm_time": round(agg[t]["ms"] / total_ms, 3),
                              "achieved": round(agg[t]["flops"] / (agg[t]["ms"] * 1e-3) / 1e12, 1),
                              "frac": round(agg[t]["flops"] / (agg[t]["ms"] * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS, 4)}
                             for t in [u for u in sorted(agg, key=lambda t: -agg[t]["ms"]) if u != dom][:2]],
            "share_of_gemm_time": round(a["ms"] / total_ms, 3),
            "dominant_rule": "most GEMM time per kernel instance; instances within 15 % of the maximum tie, broken by FLOPs",
            "per_tile": detail}, dom


_KTILES = {1: (4, 5, 2), 2: (4, 4, 2), 3: (2, 5, 2), 4: (1, 2, 4), 5: (4, 1, 2), 6: (1, 5, 4), 7: (1, 4, 4), 8: (2, 5, 4), 9: (2, 4, 4)}


def mangled_gemm_name(tile: int, dtype: str) -> str:
    """Kernel symbol of an idb_gemm_plan tile id as rocprofv3 -M lists it (idb_gemm.hip: kTiles, launch_all)."""
    t = "DF16b" if dtype == "bf16" else "DF16_"
    gn, tile = tile >= 1000, tile % 1000
    mf, nf, wm = _KTILES[tile % 10]
    v = tile // 10
    if v == 10:                                            # idb_conv_patch_kernel<T, MF, NF, NS, GN> on the shape's own tile (MF = rows / 64)
        return f"idb_conv_patch_kernelI{t}Li{mf * wm // 4}ELi{nf}ELi4ELb{int(gn)}EE"
    if gn:                                                 # idb_gemm_kernel_gn<T, MF, NF, NS, WM, NV>: 4 MFMA waves (2 x 2)
        return f"idb_gemm_kernel_gnI{t}Li2ELi{nf}ELi4ELi2ELi8EE"
    if v == 3:
        return f"idb_gemm_kernel_rsI{t}Li{mf}ELi{nf}EE"
    if v == 4:
        return f"idb_gemm_kernel_plI{t}Li{mf}ELi{nf}EE"
    if v == 9:                                             # idb_conv_patch_kernel<T, 4, NF, 3, false>
        return f"idb_conv_patch_kernelI{t}Li4ELi{nf}ELi3ELb0EE"
    if v >= 5:                                             # idb_gemm_kernel_lw<T, MF, NF, NS, WM, LW>
        ns, lw = {5: (3, 4), 6: (3, 8), 7: (4, 4), 8: (3, 4)}[v]
        return f"idb_gemm_kernel_lwI{t}Li{mf * (2 if v == 8 else 1)}ELi{nf}ELi{ns}ELi{wm}ELi{lw}EE"
    return f"idb_gemm_kernelI{t}Li{mf}ELi{nf}ELi{v + 2}ELi{wm}EE"


def kernel_source_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: ties a committed rocprofv3 summary to the
    build it was taken on (tools/profile_meta.py writes it next to the summary; .git does not travel to the GPU box)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    csrc = os.path.join(here, "faceposegenerator_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for f in files + [os.path.join(here, "include", "idb_kernels.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def attach_profile_evidence(roof, dom_tile, batch, args):
    """`achieved` / `frac` / `avg_launch_us` are THIS run's live HIP-event measurement (kernel_roofline).  Beside them, for the default
    workload only, the COMMITTED rocprofv3 evidence of the same command (profiles/<round>/, see its README):
      * `profiled` = the dominant kernel's average duration in the committed `rocprofv3 --kernel-trace --stats` summary and the
        TFLOP/s / fraction that follow from this run's FLOPs per launch — attached only when the summary's recorded kernel-source
        hash equals the hash of the sources this library was built from (a kernel change after the profile drops it);
      * `traffic` = HBM bytes per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE passes (2 x FETCH_SIZE + WRITE_SIZE, the
        gfx950 correction), summarised by tools/pmc_summary.py, under the same hash condition.
    Counters cannot be read from inside the process; any other workload keeps null."""
    roof["source"] = "live HIP events on the launch stream (empty event pair subtracted), this run"
    roof["profiled"] = None
    if batch != 1 or args.tiny or args.ddpm_steps != 30 or args.size != 512:
        return
    here = os.path.dirname(os.path.abspath(__file__))
    sym = mangled_gemm_name(dom_tile, args.dtype)
    sha = kernel_source_sha16()
    for rnd in sorted(os.listdir(os.path.join(here, "profiles")), reverse=True):
        f = os.path.join(here, "profiles", rnd, f"bench_default_b1_{args.dtype}_kernel_stats.csv")
        meta = os.path.join(here, "profiles", rnd, "kernel_source_sha16.json")
        if not os.path.isfile(f):
            continue
        rec = json.load(open(meta)).get("sha16") if os.path.isfile(meta) else None
        if rec != sha:
            roof["profiled"] = {"stale": True, "note": f"profiles/{rnd} was taken on kernel sources {rec}, this build is {sha}: not attached"}
            return
        for row in csv.DictReader(open(f)):
            if sym in row["Name"]:
                us = float(row["AverageNs"]) / 1e3
                ach = roof["flops_per_launch_avg"] / (us * 1e-6) / 1e12
                roof["profiled"] = {"achieved": round(ach, 1), "frac": round(ach / PEAK_MFMA_TFLOPS, 4), "avg_launch_us": round(us, 2),
                                    "kernel_source_sha16": sha,
                                    "source": f"profiles/{rnd}/{os.path.basename(f)}: {sym} average over {row['Calls']} launches "
                                              f"(rocprofv3 --kernel-trace --stats of this command)"}
                break
        t = os.path.join(here, "profiles", rnd, f"pmc_bench_b1_{args.dtype}_traffic.json")
        if os.path.isfile(t):
            e = json.load(open(t)).get(roof["kernel"])
            if e and "hbm_bytes_per_launch" in e:
                roof["traffic"] = round(e["hbm_bytes_per_launch"], 1)
                roof["traffic_source"] = f"profiles/{rnd}/{os.path.basename(t)} (bytes per launch, {int(e['launches'])} launches)"
        return


class _FakePipe:
    """IDB_BENCH_FAKE=1 (tests/test_bench_launcher_cpu.py only): a CPU stand-in for the pipeline so that the launcher and the
    multi-rank branch of this file (process group, all-gather, max-over-ranks timing, ONE JSON line) run under gloo without a
    GPU.  Never a benchmark: the line says so in `data`."""
    use_graph = False
    vae_chunk = 4

    def load_lora_weights(self, _):
        pass

    def prepare_noise(self, batch, steps, h, w, gen):
        return torch.randn((steps + 1, batch, 4, h // 8, w // 8), generator=gen)

    def __call__(self, prompt_embeds=None, height=512, width=512, noise=None, **kw):
        from types import SimpleNamespace
        b = prompt_embeds.shape[0]
        v = (noise[0].abs().sum(dim=(1, 2, 3)) * 1000).to(torch.int64) % 251
        return SimpleNamespace(images=v.to(torch.uint8).view(b, 1, 1, 1).expand(b, height, width, 3).contiguous())


def time_steps(step, n, world, dist, fake):
    """Barrier + device sync on both sides, wall clock, MAX over ranks."""
    sync = (lambda: None) if fake else torch.cuda.synchronize
    sync()
    if world > 1:
        dist.barrier()
    sync()
    ev0 = ev1 = None
    if not fake:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    t0 = time.perf_counter()
    img = None
    for _ in range(n):
        img = step()
    if not fake:
        ev1.record()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1) if not fake else elapsed * 1e3
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=img.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    return elapsed, gpu_ms, img


def stage_times(pipe, eng, pe_d, ne_d, noise, args, reps=3):
    """Per-stage device time of one step (HIP events on the launch stream): the 30-step sampling graph, the VAE decode +
    postprocess, and the D2H copy of the uint8 images (the sink's input)."""
    sch = pipe.scheduler
    sch.set_timesteps(args.ddpm_steps)
    ts = sch.timesteps.tolist()
    coefs = torch.tensor([list(sch.step_coefficients(t)) + [5.0] for t in ts], dtype=torch.float32).to(eng.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = [0.0, 0.0, 0.0]
    host = torch.empty((pe_d.shape[0], args.size, args.size, 3), dtype=torch.uint8).pin_memory()
    for _ in range(reps):
        ev[0].record()
        lat = eng.sample(pe_d, ne_d, noise, ts, coefs, use_graph=pipe.use_graph)
        ev[1].record()
        _, u8 = eng.decode_images(lat, chunk=pipe.vae_chunk)
        ev[2].record()
        host.copy_(u8, non_blocking=True)
        ev[3].record()
        torch.cuda.synchronize()
        for i in range(3):
            acc[i] += ev[i].elapsed_time(ev[i + 1]) / reps
    return {"sampling_loop_ms": round(acc[0], 3), "vae_decode_postprocess_ms": round(acc[1], 3), "d2h_uint8_ms": round(acc[2], 3),
            "text_encoder_ms": None}


class _TimedPipe:
    """Forwards to the pipeline and accumulates the host-observed time (device synchronised on both sides) of LoRA switches."""
    def __init__(self, pipe):
        self._p, self.lora_ms, self.lora_n = pipe, 0.0, 0

    def __getattr__(self, name):
        return getattr(self._p, name)

    def __call__(self, *a, **kw):
        return self._p(*a, **kw)

    def load_lora_weights(self, src, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self._p.load_lora_weights(src, **kw)
        torch.cuda.synchronize()
        self.lora_ms += (time.perf_counter() - t0) * 1e3
        self.lora_n += 1


def driver_points(pipe, ucfg, dev, args):
    """What the reference script actually runs, through driver.generate (inference_ID-Booth.py:86-156), bounded to a few seconds each:
      config2_mixed : BASELINE configs[2] AS STATED — 8 identities x 8 prompts = 64 images, one merged LoRA set per identity
                      (8 switches, 8 batch-8 sampler calls), prompt embeddings given;
      driver_e2e    : one identity x 3 LoRA models x 21 prompts = 63 images with everything the script does per identity: CLIP-H text
                      encoding on the GPU (synthetic weights and token ids: no tokenizer files offline), 3 LoRA switches, batch-21
                      sampler calls, VAE decode, D2H, and the threaded PNG + comparison-JPG sink.
    Each is run twice; the second pass (graphs captured, arena warm) is the one reported."""
    import tempfile
    import zlib
    from faceposegenerator_amd import driver as D, spec as S, weights as W
    from faceposegenerator_amd.text_encoder import ClipTextEncoder
    eng = pipe._engine()
    tp = _TimedPipe(pipe)
    loras = {}

    def lora_for(model, which_id):
        return loras[(model, which_id)]

    out = {}
    # ---- configs[2] as stated: 8 IDs x 8 prompts
    ids = [f"ID_{i + 1}" for i in range(8)]
    cfg = D.PolicyConfig(num_prompts=8, models_to_test=("ID-Booth",), num_inference_steps=args.ddpm_steps, height=args.size, width=args.size)
    items = D.build_work_list(ids, {i: "M" for i in ids}, cfg)
    for k, i in enumerate(ids):
        loras[("ID-Booth", i)] = W.synth_lora(ucfg, seed=100 + k)
    embed = D.synthetic_embed_fn(ucfg.cross_attention_dim)
    res = {}
    for name, G in (("grouped", 8), ("subbatched", 1)):
        for rep in range(2):
            tp.lora_ms, tp.lora_n = 0.0, 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            imgs, order = D.generate(tp, items, embed, cfg, lora_for=lora_for, max_batch=64, group_identities=G)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        assert tuple(imgs.shape) == (64, args.size, args.size, 3)
        res[name] = (el, tp.lora_n, tp.lora_ms)
    el, ln, lms = res["grouped"]
    out["config2_mixed"] = {"workload": "BASELINE configs[2] as stated: 8 identities x 8 prompts through driver.generate in ONE batch-64 sampler call: each "
                                        "group of 8 samples uses its identity's merged LoRA weights (idb_gemm_desc.w_groups), prompt embeddings given",
                            "images_per_s": round(64 / el, 3), "wall_s": round(el, 3), "lora_set_loads": ln,
                            "lora_load_ms": round(lms, 2),
                            "subbatched": {"note": "the same work as 8 batch-8 calls with one LoRA switch each (round 2's form)",
                                           "images_per_s": round(64 / res["subbatched"][0], 3), "lora_switches": res["subbatched"][1],
                                           "lora_switch_ms_each": round(res["subbatched"][2] / max(1, res["subbatched"][1]), 2)}}
    # ---- the script's per-identity work: 3 models x 21 prompts, text encoder, sink
    ccfg = S.SD21_CLIP
    te = ClipTextEncoder(eng, ccfg, W.synth_clip(ccfg, 99))
    text_ms = [0.0]

    def embed_clip(prompts):
        ids_t = torch.zeros((len(prompts), 77), dtype=torch.int64)
        for r, ptxt in enumerate(prompts):
            toks = [zlib.crc32(wd.encode()) % 49000 + 1 for wd in ptxt.replace(",", " ,").split()][:75]
            ids_t[r, 0] = ccfg.bos_token_id
            ids_t[r, 1:1 + len(toks)] = torch.tensor(toks, dtype=torch.int64)
            ids_t[r, 1 + len(toks)] = ccfg.eos_token_id
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = te.encode(ids_t)
        torch.cuda.synchronize()
        text_ms[0] += (time.perf_counter() - t0) * 1e3
        return e

    cfg = D.PolicyConfig(num_inference_steps=args.ddpm_steps, height=args.size, width=args.size)
    items = D.build_work_list(["ID_1"], {"ID_1": "M"}, cfg)
    for k, m in enumerate(cfg.models_to_test):
        loras[(m, "ID_1")] = W.synth_lora(ucfg, seed=200 + k)
    with tempfile.TemporaryDirectory() as tmp:
        for rep in range(2):
            tp.lora_ms, tp.lora_n, text_ms[0] = 0.0, 0, 0.0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            imgs, order = D.generate(tp, items, embed_clip, cfg, lora_for=lora_for, max_batch=64)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            paths = D.save_outputs(imgs, order, tmp, cfg)
            t2 = time.perf_counter()
    n = len(items)
    out["driver_e2e"] = {"workload": f"one identity x {len(cfg.models_to_test)} LoRA models x {cfg.num_prompts} prompts = {n} images through driver.generate + "
                                     "save_outputs (inference_ID-Booth.py:86-156): CLIP-H text encoding on the GPU (synthetic weights / token ids), "
                                     "LoRA switches, batch-21 sampler calls, decode, D2H, threaded PNG + comparison-JPG sink",
                         "images_per_s": round(n / (t2 - t0), 3), "images_per_s_without_sink": round(n / (t1 - t0), 3),
                         "text_encoder_ms": round(text_ms[0], 2), "text_encoder_prompts": 2 * n,
                         "lora_switch_ms": round(tp.lora_ms, 2), "lora_switches": tp.lora_n,
                         "png_ms": round((t2 - t1) * 1e3, 1), "files": len(paths)}
    del te
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))           # before anything initialises the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    fake = os.environ.get("IDB_BENCH_FAKE") == "1"
    # IDB_FORCE_DIST=1 (tests/test_rccl_gpu.py): run the process-group init and the collectives of the N > 1 path at ANY world size,
    # so that the RCCL plumbing (init with device_id, all_gather_into_tensor on device tensors, barrier, MAX all-reduce) is exercised
    # on a one-GPU box before the first 8-GPU run
    force_dist = os.environ.get("IDB_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ
    import torch.distributed as dist
    if fake:
        dev = torch.device("cpu")
        if world > 1 or force_dist:
            dist.init_process_group("gloo")
    else:
        dev = torch.device(f"cuda:{local_rank}")
        torch.cuda.set_device(dev)
        if world > 1 or force_dist:
            dist.init_process_group("nccl", device_id=dev)

    from faceposegenerator_amd import spec as S, weights as W

    ucfg, vcfg = (S.TINY_UNET, S.TINY_VAE) if args.tiny else (S.SD21_UNET, S.SD21_VAE)
    if args.vpred:
        import dataclasses
        ucfg = dataclasses.replace(ucfg, prediction_type="v_prediction")
    if args.dtype == "fp8":
        args.no_kernel_roofline = True          # the per-launch replay covers the 16-bit GEMM entry point only
    t0 = time.perf_counter()
    lora_raw = None
    if fake:
        pipe, eng = _FakePipe(), None
    else:
        from faceposegenerator_amd.pipeline import StableDiffusionPipeline
        pipe = StableDiffusionPipeline.from_synthetic(ucfg, vcfg, seed=1234, torch_dtype=args.dtype).to(dev)
        lora_raw = W.synth_lora(ucfg, seed=rank + 1)          # one identity ("ID_<rank+1>") per GPU: shard by identity
        pipe.load_lora_weights(lora_raw)
        pipe.use_graph = not args.no_graph
        pipe.vae_chunk = args.vae_chunk
        eng = pipe._engine()
        torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    lat_side = args.size // 8

    def make_step(B):
        g = torch.Generator().manual_seed(1000 + rank)
        pe = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g).to(dev)
        ne = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g).to(dev)
        # noise drawn once on the host CPU generator (inference_ID-Booth.py:111 seeds it with the identity index) and
        # resident in HBM before the timed region, like every other input
        noise = pipe.prepare_noise(B, args.ddpm_steps, args.size, args.size, torch.Generator().manual_seed(rank)).to(dev)
        gathered = torch.empty((world * B, args.size, args.size, 3), dtype=torch.uint8, device=dev) if (world > 1 or force_dist) else None

        def step():
            out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=args.ddpm_steps, guidance_scale=5.0,
                       height=args.size, width=args.size, output_type="uint8", noise=noise)
            if world > 1 or force_dist:
                dist.all_gather_into_tensor(gathered, out.images)      # the ONE collective of the job (RCCL over xGMI)
                if force_dist and world == 1:
                    assert torch.equal(gathered, out.images), "all_gather_into_tensor at world size 1 must return the local images"
            return out.images
        return step, (pe, ne, noise)

    B = args.batch
    step, (pe_d, ne_d, noise) = make_step(B)
    for _ in range(args.warmup):
        img = step()
    elapsed, gpu_ms, img = time_steps(step, args.steps, world if not force_dist else max(world, 2), dist, fake)
    if force_dist:
        from faceposegenerator_amd import driver as D
        g2 = D.all_gather_images(img, [img.shape[0]] * world, force=True)         # driver.generate's collective on device tensors
        assert torch.equal(g2, img) if world == 1 else g2.shape[0] == world * img.shape[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (B, args.size, args.size, 3)

    fl_unet, fl_vae = 2.0 * S.unet_macs(ucfg, lat_side), 2.0 * S.vae_decode_macs(vcfg, lat_side)
    flops_per_image = 2 * args.ddpm_steps * fl_unet + fl_vae
    arena_mib = None if fake else round(eng.arena.total_bytes / 2 ** 20, 1)
    default_workload = B == 1 and not args.tiny and args.ddpm_steps == 30 and args.size == 512 and not fake
    config2 = None
    if default_workload and world == 1 and not args.no_config2:
        # BASELINE configs[2] (batch 64 = the attention/conv throughput point), timed briefly in the same run so that the figure
        # is driver-observed: 1 warm-up call (eager pass + graph capture + replay) and 2 timed steps
        step64, _ = make_step(64)
        step64()
        e64, g64, _ = time_steps(step64, 2, 1, dist, False)
        tf64 = 64 * 2 * flops_per_image / (g64 * 1e-3) / 1e12
        config2 = {"workload": "BASELINE configs[2]: batch 64/GPU, same graph, LoRA, 30 steps", "images_per_s": round(64 * 2 / e64, 3),
                   "steps": 2, "warmup": 1, "ms_per_step": round(e64 / 2 * 1e3, 1), "tflops": round(tf64, 1),
                   "frac_of_mfma_peak": round(tf64 / PEAK_MFMA_TFLOPS, 4), "arena_mib": round(eng.arena.total_bytes / 2 ** 20, 1)}

    drv_points = None
    if default_workload and world == 1 and not args.no_driver_points:
        if config2 is not None:
            del step64
            step64 = None
            torch.cuda.empty_cache()
        drv_points = driver_points(pipe, ucfg, dev, args)
        pipe.load_lora_weights(lora_raw)                   # back to this rank's identity for the stage timings below

    fp8_point = None
    if default_workload and world == 1 and not args.no_config2 and not args.no_fp8_point and args.dtype == "f16":
        # the same batch-64 point on the fp8 path (BASELINE configs[4]'s "fp8 MFMA weight path", here at 512x512 so that it sits beside
        # config2): a second engine with e4m3 resnet convs; its parity class is stated in DESIGN.md section 2.3 (opt-in, not the default)
        step64 = None
        pipe8 = StableDiffusionPipeline.from_synthetic(ucfg, vcfg, seed=1234, torch_dtype="fp8").to(dev)
        pipe8.load_lora_weights(lora_raw)
        pipe8.use_graph, pipe8.vae_chunk = pipe.use_graph, pipe.vae_chunk
        keep, pipe = pipe, pipe8
        step8, _ = make_step(64)
        step8()
        e8, g8, _ = time_steps(step8, 2, 1, dist, False)
        pipe = keep
        tf8 = 64 * 2 * flops_per_image / (g8 * 1e-3) / 1e12
        fp8_point = {"workload": "batch 64/GPU, 512x512, 30 steps, LoRA; e4m3 operands on v_mfma_scale_f32_16x16x128_f8f6f4 for the 44 ResnetBlock2D "
                                 "3x3 convs, f16 elsewhere (python bench.py --batch 64 --dtype fp8)",
                     "images_per_s": round(64 * 2 / e8, 3), "steps": 2, "warmup": 1, "ms_per_step": round(e8 / 2 * 1e3, 1),
                     "algorithmic_tflops": round(tf8, 1)}
        del pipe8, step8
        torch.cuda.empty_cache()

    if rank == 0:
        images = world * B * args.steps
        value = images / elapsed
        path_tflops = (B * args.steps * flops_per_image) / (gpu_ms * 1e-3) / 1e12
        res = {
            "metric": "512x512 images/sec/node, SD-2.1-base 30-step DDPM CFG=5.0 + LoRA",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if not fake else "FAKE pipeline on the CPU (launcher test, not a benchmark)",
            "config": {"workload": ("BASELINE configs[4], per-GPU share (batch 32 of 256 over 8 GPUs)" if (args.size == 768 and B == 32) else
                                    "BASELINE configs[1]" if B == 1 else "BASELINE configs[2]" if B == 64 else "custom") +
                       f": SD-2.1-base graph{' (TINY, not a benchmark)' if args.tiny else ''} + rank-4 LoRA, {args.size}x{args.size}, "
                       f"{args.ddpm_steps} DDPM steps, CFG 5.0, batch {B}/GPU, synthetic weights",
                       "batch_per_gpu": B, "ddpm_steps": args.ddpm_steps, "guidance_scale": 5.0, "hip_graph": not args.no_graph,
                       "prediction_type": "v_prediction" if args.vpred else "epsilon",
                       **({"fp8_scope": "e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4) for the 44 ResnetBlock2D 3x3 convs = 40.6 % of the "
                                        "UNet FLOPs; f16 operands elsewhere"} if args.dtype == "fp8" else {}),
                       "parallelism": f"identity-sharded x{world}, one all-gather of uint8 images per step" if world > 1 else "single GPU",
                       **({"forced_collectives": dist.get_backend()} if force_dist else {})},
        }
        if not fake:
            res["path"] = {"algorithmic_tflop_per_image": round(flops_per_image / 1e12, 3), "tflops": round(path_tflops, 1),
                           "frac_of_mfma_peak": round(path_tflops / PEAK_MFMA_TFLOPS, 4), "gpu_ms_per_step": round(gpu_ms / args.steps, 3),
                           "load_pack_s": round(t_load, 1), "arena_mib": arena_mib}
            if world == 1:
                res["path"]["stages"] = stage_times(pipe, eng, pe_d, ne_d, noise, args)
            if config2:
                res["path"]["config2"] = config2
            if fp8_point:
                res["path"]["config2_fp8"] = fp8_point
            if drv_points:
                res["path"].update(drv_points)
                res["path"]["stages"]["text_encoder_ms"] = round(2 * drv_points["driver_e2e"]["text_encoder_ms"] / drv_points["driver_e2e"]["text_encoder_prompts"], 3)   # prompt + negative prompt of one image
            if not args.no_kernel_roofline:
                res["roofline"], dom = kernel_roofline(eng, B, lat_side, 77)
                attach_profile_evidence(res["roofline"], dom, B, args)
            else:
                res["roofline"] = {"bound": "mfma", "achieved": round(path_tflops, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(path_tflops / PEAK_MFMA_TFLOPS, 4), "traffic": None}
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(pipe, ucfg, vcfg, lora_raw, args.ddpm_steps, args.size, args.cpu_baseline_threads)
        print(json.dumps(res), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
