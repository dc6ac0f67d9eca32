#!/usr/bin/env python3
"""Batch-1 projection-GEMM shapes (cold weights, in a HIP graph): time per launch for each tile / split-K choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", os.environ.get("IDB_DTYPE", "f16"))
dev = eng.device
shapes = [(8192, 320, 320), (8192, 960, 320), (8192, 320, 1280), (2048, 640, 640), (2048, 1920, 640), (2048, 640, 2560),
          (512, 1280, 1280), (512, 3840, 1280), (512, 1280, 5120), (128, 1280, 1280), (128, 3840, 1280), (128, 1280, 5120)]
for (m, n, k) in shapes:
    nbuf = max(2, min(48, int(400e6 // (n * k * 2))))
    ws = [eng.tile_weight((torch.randn(n, k, device=dev) * k ** -0.5).to(eng.tdt)) for _ in range(nbuf)]
    x = torch.randn(m, k, device=dev).to(eng.tdt)
    res = torch.randn(m, n, device=dev).to(eng.tdt)
    out = torch.empty(m, n, dtype=eng.tdt, device=dev)
    line = []
    combos = ((0, 0), (17, 1), (4, 1), (14, 1), (14, 2), (17, 2), (14, 4), (17, 4), (16, 1), (9, 1))
    if os.environ.get("IDB_COMBOS"):        # e.g. IDB_COMBOS="0:0,74:1,77:2"  (tile:split_k)
        combos = [tuple(int(v) for v in c.split(":")) for c in os.environ["IDB_COMBOS"].split(",")]
    for tile, sk in combos:
        if n % 160 and tile % 10 in (1, 3, 6, 8):
            continue
        if sk > 1 and k // 64 < 5 * sk:
            continue
        def run(i):
            eng.gemm([(x, k, 1, 1, 1, 0)], ws[i % nbuf], n, m, 1, 1, out=out, residual=res, split_k=sk, tile=tile)
        try:
            for i in range(nbuf): run(i)
        except RuntimeError:
            continue
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nbuf): run(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        line.append(f"t{tile}s{sk}:{e0.elapsed_time(e1) / (2 * nbuf) * 1e3:5.1f}")
    print(f"m={m:5d} n={n:5d} k={k:5d} us/launch  " + "  ".join(line), flush=True)
    del ws
