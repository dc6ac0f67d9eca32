#!/usr/bin/env python3
"""Batch-1 (B_eff 2) conv3x3 shapes, cold operands (rotating buffer sets > MALL), in a HIP graph: us per launch
(GEMM + split-K reduce) for each (tile, split_k).  Usage: python tools/bench_conv.py [B_eff]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", os.environ.get("IDB_DTYPE", "f16"))
dev = eng.device
be = int(sys.argv[1]) if len(sys.argv) > 1 else 2
shapes = [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640), (32, 320, 640), (16, 1280, 1280), (16, 2560, 1280),
          (8, 1280, 1280), (8, 2560, 1280), (16, 640, 1280), (64, 512, 512), (128, 256, 256)]
combos = [(0, 0), (16, 1), (18, 1), (18, 2), (18, 3), (18, 4), (18, 6), (18, 8), (18, 12), (18, 16), (18, 24), (17, 4), (17, 8), (17, 16), (19, 2), (19, 4), (19, 8)]
if os.environ.get("IDB_COMBOS"):        # e.g. IDB_COMBOS="0:0,6:2,6:4,8:4,8:8"  (tile:split_k)
    combos = [tuple(int(v) for v in c.split(":")) for c in os.environ["IDB_COMBOS"].split(",")]
elif be >= 8:
    combos = [(0, 0), (8, 1), (88, 1), (98, 1), (9, 1), (89, 1), (99, 1)]
for (side, cin, cout) in shapes:
    m, k = be * side * side, 9 * cin
    per = 2 * (m * cin + cout * k + m * cout)
    nbuf = max(3, min(64, int(600e6 // per)))
    xs = [torch.randn(m, cin, device=dev).to(eng.tdt) for _ in range(nbuf)]
    ws = [eng.tile_weight((torch.randn(cout, k, device=dev) * k ** -0.5).to(eng.tdt)) for _ in range(nbuf)]
    outs = [torch.empty(m, cout, dtype=eng.tdt, device=dev) for _ in range(nbuf)]
    line = []
    for tile, sk in combos:
        if sk > 1 and k // 64 < 6 * sk:
            continue
        if cout % 160 and tile % 10 in (6, 8):
            continue
        if cout % 160 == 0 and tile % 10 == 9 and be >= 8:
            continue
        def run(i):
            eng.gemm([(xs[i], cin, 9, side, side, 0)], ws[i], cout, be, side, side, out=outs[i], split_k=sk, tile=tile)
        try:
            for i in range(nbuf): run(i)
        except RuntimeError:                  # a forced tile the shape cannot run (IDB_EUNSUPPORTED)
            continue
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nbuf): run(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (2 * nbuf) * 1e3
        line.append(f"t{tile}s{sk}:{us:6.1f}" + (f" ({2.0 * m * cout * k / us / 1e6:4.0f}TF)" if be >= 8 else ""))
    print(f"conv {cin}->{cout} @{side} m={m:5d} k={k:5d} ({2.0 * m * cout * k / 1e9:5.1f} GF) us  " + " ".join(line), flush=True)
    del xs, ws, outs
