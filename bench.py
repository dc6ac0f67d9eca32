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
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_MFMA_TFLOPS = 2500.0      # dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
_TILE_SHAPES = {1: "128x160", 2: "128x128", 3: "64x160", 4: "64x64,8w", 5: "128x32",
                6: "64x160,8w", 7: "64x128,8w", 8: "128x160,8w", 9: "128x128,8w"}
_TILE_VARIANTS = {0: "idb_gemm_kernel<{},ring2>", 1: "idb_gemm_kernel<{},ring3>", 2: "idb_gemm_kernel<{},ring4>",
                  3: "idb_gemm_kernel_rs<{}>", 4: "idb_gemm_kernel_pl<{}>"}


class _TileNames(dict):
    """idb_gemm_plan tile id -> kernel instance name (id = shape + 10 * variant, see idb_gemm.hip)."""
    def __missing__(self, t):
        return _TILE_VARIANTS[t // 10].format(_TILE_SHAPES[t % 10])


TILE_NAMES = _TileNames()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (1 = BASELINE configs[1], 64 = configs[2])")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--ddpm-steps", type=int, default=30)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="reduced-width graph (plumbing check only; not a benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--vae-chunk", type=int, default=4)
    return ap.parse_args()


def cpu_baseline(pipe, ucfg, vcfg, lora_raw, ddpm_steps, size):
    """The oracle (CPU fp32 restatement) timed on the host cores on a bounded sample: 2 CFG UNet forwards (B_eff=2)
    and one VAE decode of the same graph and weights; extrapolated to one image = ddpm_steps forwards + 1 decode."""
    from oracle import sd21_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))          # the CPUs this process may actually run on
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, torch.get_num_threads()))
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
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_fw} CFG UNet forwards ({t_unet:.2f} s each) + 1 VAE decode ({t_vae:.2f} s) of the full graph, "
                      f"extrapolated to {ddpm_steps} steps + decode"}


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
    empty = []
    for r in range(REPS + 1):                                            # pass 0 = warm-up, untimed
        for e in log:
            ws, need = e["ws"]
            ev0, ev1 = pair()
            ev0.record()
            L.check(eng.lib.idb_gemm(C.byref(e["desc"]), None if ws is None else ws.data_ptr(), need, st))
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
    dom = max(agg, key=lambda t: agg[t]["ms"])
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
            "all_gemm_tflops": round(total_fl / (total_ms * 1e-3) / 1e12, 1), "per_tile": detail}


def attach_pmc_traffic(roof, batch, args):
    """`traffic`: HBM bytes per launch of the dominant kernel from the PMC pass over THIS command (rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs, 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for
    gfx950; summarised by tools/pmc_summary.py into profiles/<round>/pmc_bench_b1_traffic.json).  Counters cannot be read
    from inside the process, so the committed summary of the default workload is attached; any other workload -> null."""
    if batch != 1 or args.dtype != "bf16" or args.tiny or args.ddpm_steps != 30 or args.size != 512:
        return
    here = os.path.dirname(os.path.abspath(__file__))
    for rnd in sorted(os.listdir(os.path.join(here, "profiles")), reverse=True):
        f = os.path.join(here, "profiles", rnd, "pmc_bench_b1_traffic.json")
        if os.path.isfile(f):
            e = json.load(open(f)).get(roof["kernel"])
            if e and "hbm_bytes_per_launch" in e:
                roof["traffic"] = round(e["hbm_bytes_per_launch"], 1)
                roof["traffic_source"] = f"profiles/{rnd}/pmc_bench_b1_traffic.json (bytes per launch, {int(e['launches'])} launches)"
            return


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline

    ucfg, vcfg = (S.TINY_UNET, S.TINY_VAE) if args.tiny else (S.SD21_UNET, S.SD21_VAE)
    t0 = time.perf_counter()
    pipe = StableDiffusionPipeline.from_synthetic(ucfg, vcfg, seed=1234, torch_dtype=args.dtype).to(dev)
    lora_raw = W.synth_lora(ucfg, seed=rank + 1)          # one identity ("ID_<rank+1>") per GPU: shard by identity
    pipe.load_lora_weights(lora_raw)
    pipe.use_graph = not args.no_graph
    pipe.vae_chunk = args.vae_chunk
    eng = pipe._engine()
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0

    B = args.batch
    g = torch.Generator().manual_seed(1000 + rank)
    pe = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g)
    ne = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g)
    pe_d, ne_d = pe.to(dev), ne.to(dev)
    lat_side = args.size // 8
    # noise drawn once on the host CPU generator (inference_ID-Booth.py:111 seeds it with the identity index) and
    # resident in HBM before the timed region, like every other input
    gen = torch.Generator().manual_seed(rank)
    noise = pipe.prepare_noise(B, args.ddpm_steps, args.size, args.size, gen).to(dev)
    gathered = torch.empty((world * B, args.size, args.size, 3), dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        out = pipe(prompt_embeds=pe_d, negative_prompt_embeds=ne_d, num_inference_steps=args.ddpm_steps, guidance_scale=5.0,
                   height=args.size, width=args.size, output_type="uint8", noise=noise)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out.images)      # the ONE collective of the job (RCCL over xGMI)
        return out.images

    for _ in range(args.warmup):
        img = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        img = step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    assert img.dtype == torch.uint8 and tuple(img.shape) == (B, args.size, args.size, 3)

    if rank == 0:
        images = world * B * args.steps
        value = images / elapsed
        fl_unet, fl_vae = 2.0 * S.unet_macs(ucfg, lat_side), 2.0 * S.vae_decode_macs(vcfg, lat_side)
        flops_per_image = 2 * args.ddpm_steps * fl_unet + fl_vae
        path_tflops = (B * args.steps * flops_per_image) / (gpu_ms * 1e-3) / 1e12
        res = {
            "metric": "512x512 images/sec/node, SD-2.1-base 30-step DDPM CFG=5.0 + LoRA",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]" if B == 1 else "BASELINE configs[2]" if B == 64 else "custom") +
                       f": SD-2.1-base graph{' (TINY, not a benchmark)' if args.tiny else ''} + rank-4 LoRA, {args.size}x{args.size}, "
                       f"{args.ddpm_steps} DDPM steps, CFG 5.0, batch {B}/GPU, synthetic weights",
                       "batch_per_gpu": B, "ddpm_steps": args.ddpm_steps, "guidance_scale": 5.0, "hip_graph": not args.no_graph,
                       "parallelism": f"identity-sharded x{world}, one all-gather of uint8 images per step" if world > 1 else "single GPU"},
            "path": {"algorithmic_tflop_per_image": round(flops_per_image / 1e12, 3), "tflops": round(path_tflops, 1),
                     "frac_of_mfma_peak": round(path_tflops / PEAK_MFMA_TFLOPS, 4), "gpu_ms_per_step": round(gpu_ms / args.steps, 3),
                     "load_pack_s": round(t_load, 1), "arena_mib": round(eng.arena.total_bytes / 2 ** 20, 1)},
        }
        if not args.no_kernel_roofline:
            res["roofline"] = kernel_roofline(eng, B, lat_side, 77)
            attach_pmc_traffic(res["roofline"], B, args)
        else:
            res["roofline"] = {"bound": "mfma", "achieved": round(path_tflops, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(path_tflops / PEAK_MFMA_TFLOPS, 4), "traffic": None}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(pipe, ucfg, vcfg, lora_raw, args.ddpm_steps, args.size)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
