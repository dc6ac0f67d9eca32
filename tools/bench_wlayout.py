#!/usr/bin/env python3
"""Weight layout A/B on the weight-streaming GEMMs of the batch-1 UNet (cold weights, rotated over > 256 MB, inside a HIP graph):
[n][K] rows (a K-step of a column tile = BN pieces of 128 B at a K*2-byte stride) against the K-tiled 16-row blocks of
idb_tile_weight (whole 2 KiB runs, one contiguous stream per row block).  us per launch incl. the split-K reduce.

  python tools/bench_wlayout.py [B_eff]     (default 2 = batch 1 with CFG)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
beff = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "f16")
dev = eng.device
# (h, cin, cout, taps): the convs and projections of the 8x8 / 16x16 / 32x32 levels + two 64x64 ones
shapes = [(8, 1280, 1280, 9), (8, 2560, 1280, 9), (16, 1280, 1280, 9), (16, 2560, 1280, 9), (16, 1920, 1280, 9), (32, 640, 640, 9),
          (32, 1280, 640, 9), (32, 960, 640, 9), (64, 320, 320, 9), (64, 640, 320, 9),
          (16, 1280, 1280, 1), (16, 1280, 3840, 1), (16, 5120, 1280, 1), (32, 640, 1920, 1), (32, 2560, 640, 1), (64, 320, 960, 1), (64, 1280, 320, 1)]
for (h, cin, cout, taps) in shapes:
    k = cin * taps
    nbuf = max(2, min(32, int(600e6 // (cout * k * 2))))
    ws = [(torch.randn(cout, k, device=dev) * k ** -0.5).to(eng.tdt) for _ in range(nbuf)]
    wt = [eng.tile_weight(w) for w in ws]
    x = torch.randn(beff * h * h, cin, device=dev).to(eng.tdt)
    out = torch.empty(beff * h * h, cout, dtype=eng.tdt, device=dev)
    bias = torch.randn(cout, device=dev)
    res = {}
    for name, arr in (("rows", ws), ("tiled", wt)):
        def run(i):
            if taps == 9:
                eng.gemm([(x, cin, 9, h, h, 0)], arr[i % nbuf], cout, beff, h, h, bias=bias, out=out)
            else:
                eng.gemm([(x, cin, 1, 1, 1, 0)], arr[i % nbuf], cout, beff * h * h, 1, 1, bias=bias, out=out)
        for i in range(nbuf): run(i)
        torch.cuda.synchronize()
        if name == "rows": ref = out.clone()
        else: assert torch.equal(ref, out), "layouts disagree"
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nbuf): run(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (2 * nbuf) * 1e3)
        res[name] = best
    mb = cout * k * 2 / 1e6
    print(f"{h:2d}x{h:<2d} {cin:4d}->{cout:4d} taps {taps} M={beff*h*h:5d} W {mb:5.1f} MB: rows {res['rows']:6.1f} us ({mb / res['rows'] / 1e3:4.2f} TB/s)   "
          f"tiled {res['tiled']:6.1f} us ({mb / res['tiled'] / 1e3:4.2f} TB/s)   x{res['rows'] / res['tiled']:.2f}", flush=True)
    del ws, wt
