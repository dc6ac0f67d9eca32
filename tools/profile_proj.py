#!/usr/bin/env python3
"""Runs one K = C projection GEMM of the transformer blocks (attn.to_out + residual: [M][C] x [C][C]^T, M = B_eff * tokens) a fixed
number of times so that rocprofv3 --pmc passes can attribute counters to it.
Usage: python tools/profile_proj.py [B_eff=128] [tokens=4096] [C=320] [N=C] [launches=10] [residual=1] [geglu=0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine

a = [int(v) for v in sys.argv[1:]]
be, tok, c = (a + [128, 4096, 320])[:3] if len(a) < 3 else a[:3]
n = a[3] if len(a) > 3 else c
reps = a[4] if len(a) > 4 else 10
has_res = (a[5] if len(a) > 5 else 1) != 0
geglu = (a[6] if len(a) > 6 else 0) != 0
if geglu:
    has_res = False
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", os.environ.get("IDB_DTYPE", "f16"))
m = be * tok
x = torch.randn(m, c, device=eng.device).to(eng.tdt)
w = (torch.randn(n, c, device=eng.device) * c ** -0.5).to(eng.tdt)
bias = torch.randn(n, device=eng.device)
no = n // 2 if geglu else n
res = torch.randn(m, no, device=eng.device).to(eng.tdt) if has_res else None
out = torch.empty(m, no, dtype=eng.tdt, device=eng.device)
for _ in range(reps):
    eng.gemm([(x, c, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, out=out, geglu=geglu)
torch.cuda.synchronize()
alg = 2.0 * (m * c + n * c + m * no * (2 if has_res else 1))
print(f"projection M={m} K={c} N={n} residual={has_res} geglu={geglu}: flops/launch={2.0 * m * n * c:.4e} algorithmic bytes/launch={alg:.4e}")
