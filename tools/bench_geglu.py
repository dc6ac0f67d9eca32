#!/usr/bin/env python3
"""GEGLU projections (FeedForward first GEMM, N = 8C, rows interleaved value/gate) of the batch-1 UNet, cold weights inside a HIP graph:
us per launch for the library's plan (persistent variant) and for the tile forms that carry a GEGLU epilogue (even NF).

  python tools/bench_geglu.py [B_eff]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S, _lib as L
from faceposegenerator_amd.engine import HipEngine
beff = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "f16")
dev = eng.device
for (h, c) in ((64, 320), (32, 640), (16, 1280)):
    m, k, n = beff * h * h, c, 8 * c
    nbuf = max(2, min(24, int(500e6 // (n * k * 2))))
    ws = [eng.tile_weight((torch.randn(n, k, device=dev) * k ** -0.5).to(eng.tdt)) for _ in range(nbuf)]
    x = torch.randn(m, k, device=dev).to(eng.tdt)
    out = torch.empty(m, n // 2, dtype=eng.tdt, device=dev)
    bias = torch.randn(n, device=dev)
    line, ref = [], None
    for tile in (0, 9, 19, 79, 7, 17, 77, 2, 12, 42):
        def run(i):
            eng.gemm([(x, k, 1, 1, 1, 0)], ws[i % nbuf], n, m, 1, 1, bias=bias, out=out, geglu=True, tile=tile)
        try:
            for i in range(nbuf): run(i)
        except L.IdbError:
            continue
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        same = torch.equal(out, ref)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nbuf): run(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (2 * nbuf) * 1e3)
        line.append(f"t{tile}:{best:6.1f}" + ("" if same else "!"))
    print(f"GEGLU {h:2d}x{h:<2d} K={k:4d} N={n:5d} M={m:5d} ({2.0 * m * n * k / 1e9:5.1f} GF) us  " + "  ".join(line), flush=True)
    del ws
