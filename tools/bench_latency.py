#!/usr/bin/env python3
"""Where does a small-M GEMM launch spend its time?  Fixed M x N, K swept, weights rotated over many buffers so that
every launch streams HBM-cold weights (as in the batch-1 sampling loop).  Prints per-launch time vs K-steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine

eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "bf16")
dev = eng.device
for (m, n) in ((512, 1280), (128, 1280), (2048, 640), (8192, 320)):
    for k in (320, 640, 1280, 2560, 5120, 11520):
        nbuf = max(2, min(64, int(600e6 // (n * k * 2))))
        ws = [(torch.randn(n, k, device=dev) * k ** -0.5).to(eng.tdt) for _ in range(nbuf)]
        x = torch.randn(m, k, device=dev).to(eng.tdt)
        out = torch.empty(m, n, dtype=eng.tdt, device=dev)
        for sk, fl in ((1, 0), (1, 64), (1, 128), (0, 0)):
            def run(i):
                eng.gemm([(x, k, 1, 1, 1, 0)], ws[i % nbuf], n, m, 1, 1, out=out, split_k=sk, flags=fl)
            for i in range(nbuf): run(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(nbuf): run(i)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / (2 * nbuf) * 1e3
            print(f"m={m:5d} n={n:5d} k={k:6d} ksteps={k // 64:4d} split={'auto' if sk == 0 else 1:>4} {'noMFMA' if fl == 64 else 'noLOAD' if fl == 128 else 'full  '} : {t:7.1f} us/launch  "
                  f"{2.0 * m * n * k / t / 1e6:7.1f} TF/s  weights {n * k * 2 / 1e6:6.1f} MB -> {n * k * 2 / t / 1e6:6.2f} TB/s")
        del ws
