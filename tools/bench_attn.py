#!/usr/bin/env python3
"""Self-attention launch times at the UNet's levels for B_eff given (default 2), HIP-event timed back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
be = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "bf16")
for heads, n in ((5, 4096), (10, 1024), (20, 256), (20, 64)):
    c = heads * 64
    qkv = torch.randn(be * n, 3 * c, device=eng.device).to(eng.tdt)
    p = qkv.data_ptr()
    def run():
        eng.arena.reset()
        eng.attention(qkv, 3 * c, p + 2 * c, p + 4 * c, 3 * c, be, heads, n, n, n)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * be * heads * n * n * 64
    print(f"B_eff={be} heads={heads} n={n}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  blocks128={(n + 127) // 128 * heads * be}", flush=True)
