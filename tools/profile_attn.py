#!/usr/bin/env python3
"""Runs the level-0 self-attention launch (5 heads, 4096 tokens) N times for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
be = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", os.environ.get("IDB_DTYPE", "f16"))
heads, n = 5, 4096
c = heads * 64
qkv = torch.randn(be * n, 3 * c, device=eng.device).to(eng.tdt)
p = qkv.data_ptr()
for _ in range(10):
    eng.arena.reset()
    eng.attention(qkv, 3 * c, p + 2 * c, p + 4 * c, 3 * c, be, heads, n, n, n)
torch.cuda.synchronize()
print("flops/launch", 4.0 * be * heads * n * n * 64)
