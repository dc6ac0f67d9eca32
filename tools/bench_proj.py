#!/usr/bin/env python3
"""Large-batch plain [M][K] x [N][K]^T projections of the transformer blocks (short K): TFLOP/s per tile variant.
Usage: python tools/bench_proj.py [B_eff=128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S, _lib as L
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", os.environ.get("IDB_DTYPE", "f16"))
dev = eng.device
be = int(sys.argv[1]) if len(sys.argv) > 1 else 128
# (tokens per sample, K, N, geglu, residual)
shapes = [(4096, 320, 320, 0, 1), (4096, 320, 960, 0, 0), (4096, 320, 2560, 1, 0), (4096, 1280, 320, 0, 1),
          (1024, 640, 640, 0, 1), (1024, 640, 1920, 0, 0), (1024, 640, 5120, 1, 0), (1024, 2560, 640, 0, 1),
          (256, 1280, 1280, 0, 1), (256, 1280, 3840, 0, 0), (256, 1280, 10240, 1, 0), (256, 5120, 1280, 0, 1)]
tiles = [0, 8, 9, 88, 89, 58, 59, 41, 42]
for (tok, k, n, geglu, res) in shapes:
    m = be * tok
    x = torch.randn(m, k, device=dev).to(eng.tdt)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(eng.tdt)
    bias = torch.randn(n, device=dev)
    no = n // 2 if geglu else n
    r = torch.randn(m, no, device=dev).to(eng.tdt) if res else None
    out = torch.empty(m, no, dtype=eng.tdt, device=dev)
    line = []
    for t in tiles:
        if n % 160 and t % 10 in (1, 3, 6, 8):
            continue
        if geglu and t % 10 in (1, 3, 6, 8):           # odd NF
            continue
        def run():
            eng.gemm([(x, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=r, geglu=bool(geglu), tile=t, out=out)
        try:
            run(); torch.cuda.synchronize()
        except L.IdbError:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 3 * 1e3
        line.append(f"t{t}:{2.0 * m * n * k / us / 1e6:5.0f}")
    print(f"m={m:7d} k={k:5d} n={n:5d} geglu={geglu} res={res} | TF/s " + " ".join(line), flush=True)
    del x, w, r, out
