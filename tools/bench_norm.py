#!/usr/bin/env python3
"""HBM-bound kernels against the HBM roofline: GroupNorm(+SiLU) and LayerNorm at the UNet's shapes for B_eff given
(default 32), rotating tensors (> 256 MB in flight) inside a HIP graph; reports effective GB/s = algorithmic bytes / time
(GroupNorm: read + write; its statistics pass re-reads the tensor, counted separately)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
be = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "bf16")
dev = eng.device


def timed(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e-3


for side, c in ((64, 320), (32, 640), (16, 1280), (8, 1280)):
    hw = side * side
    nbytes = be * hw * c * 2
    n = max(2, min(16, int(600e6 // (2 * nbytes))))
    xs = [torch.randn(be * hw, c, device=dev).to(eng.tdt) for _ in range(n)]
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    def gn(i):
        eng.arena.reset()
        eng.groupnorm(xs[i], c, None, 0, be, hw, gamma, beta, 1e-5, True)
    def ln(i):
        eng.arena.reset()
        eng.layernorm(xs[i], be * hw, c, gamma, beta)
    tg, tl = timed(gn, n), timed(ln, n)
    print(f"B_eff={be} {side}x{side}x{c} ({nbytes / 1e6:6.1f} MB): GroupNorm+SiLU {tg * 1e6:7.1f} us = {2 * nbytes / tg / 1e9:6.0f} GB/s "
          f"(read+write; {3 * nbytes / tg / 1e9:6.0f} GB/s with the statistics pass' re-read)   LayerNorm {tl * 1e6:7.1f} us = {2 * nbytes / tl / 1e9:6.0f} GB/s",
          flush=True)
