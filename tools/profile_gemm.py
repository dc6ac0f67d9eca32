#!/usr/bin/env python3
"""Runs the dominant implicit-GEMM launch (ResnetBlock2D conv3x3 320->320 at 64x64) a fixed number of times so that
rocprofv3 --pmc passes can attribute counters to it.  Usage: python tools/profile_gemm.py <B_eff> [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine

be = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0            # 0: the library's plan; e.g. 8 = 128x160 ring 2, 88 = 256x160 loader waves
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", os.environ.get("IDB_DTYPE", "f16"))
side, cin, cout = 64, 320, 320
x = torch.randn(be * side * side, cin, device=eng.device).to(eng.tdt)
w = eng.tile_weight((torch.randn(cout, 9 * cin, device=eng.device) * (9 * cin) ** -0.5).to(eng.tdt))
for _ in range(n):
    eng.arena.reset()
    eng.gemm([(x, cin, 9, side, side, 0)], w, cout, be, side, side, tile=tile)
torch.cuda.synchronize()
m = be * side * side
print(f"conv3x3 {cin}->{cout} @{side}x{side} B_eff={be}: M={m} N={cout} K={9 * cin} flops/launch={2.0 * m * cout * 9 * cin:.4e} "
      f"algorithmic bytes/launch={2.0 * (m * cin + cout * 9 * cin + m * cout):.4e}")
