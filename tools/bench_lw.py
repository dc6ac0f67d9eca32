#!/usr/bin/env python3
"""Loader-wave GEMM variants (idb_gemm_kernel_lw, tile ids 5x/6x/7x) against the ring-3 kernels on the batch-1 UNet's GEMM shapes
(cold weights rotated over > 256 MB inside a HIP graph; us per launch incl. the split-K reduce).

  python tools/bench_lw.py [B_eff]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from faceposegenerator_amd import spec as S, _lib as L
from faceposegenerator_amd.engine import HipEngine
beff = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "f16")
dev = eng.device
shapes = [(64, 320, 320, 9), (64, 640, 320, 9), (64, 960, 320, 9), (32, 640, 640, 9), (32, 1280, 640, 9), (16, 1280, 1280, 9), (16, 2560, 1280, 9),
          (8, 1280, 1280, 9), (8, 2560, 1280, 9), (64, 320, 960, 1), (64, 320, 320, 1), (64, 1280, 320, 1), (32, 640, 1920, 1), (32, 640, 640, 1),
          (32, 2560, 640, 1), (16, 1280, 3840, 1), (16, 1280, 1280, 1), (16, 5120, 1280, 1), (8, 1280, 1280, 1)]
for (h, cin, cout, taps) in shapes:
    k = cin * taps
    m = beff * h * h
    nbuf = max(2, min(24, int(500e6 // (cout * k * 2))))
    ws = [eng.tile_weight((torch.randn(cout, k, device=dev) * k ** -0.5).to(eng.tdt)) for _ in range(nbuf)]
    x = torch.randn(m, cin, device=dev).to(eng.tdt)
    out = torch.empty(m, cout, dtype=eng.tdt, device=dev)
    bias = torch.randn(cout, device=dev)
    srcs = [(x, cin, 9, h, h, 0)] if taps == 9 else [(x, cin, 1, 1, 1, 0)]
    dims = (beff, h, h) if taps == 9 else (m, 1, 1)
    # the plan the library picks (tile, split-K), then the same tile shape / split as loader-wave variants
    d = L.GemmDesc()
    eng.launch_log = []
    eng.gemm(srcs, ws[0], cout, *dims, bias=bias, out=out)
    e = eng.launch_log[0]; eng.launch_log = None
    tile0, sk0 = e["tile"], e["split_k"]
    cands = [("auto", 0, 0)]
    if tile0 // 10 == 1 and tile0 % 10 in (4, 6, 7, 8, 9):
        cands += [(f"lw{v}", 10 * v + tile0 % 10, sk0) for v in (5, 6, 7)]
    line = []
    ref = None
    for name, tile, sk in cands:
        def run(i):
            eng.gemm(srcs, ws[i % nbuf], cout, *dims, bias=bias, out=out, tile=tile, split_k=sk)
        for i in range(nbuf): run(i)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        else: assert torch.equal(out, ref), (name, "differs")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nbuf): run(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (2 * nbuf) * 1e3)
        line.append(f"{name}:{best:6.1f}")
    fl = 2.0 * m * cout * k
    print(f"{h:2d}x{h:<2d} {cin:4d}->{cout:4d} taps {taps} M={m:5d} plan tile {tile0} sk {sk0}:  " + "  ".join(line), flush=True)
    del ws
