#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
from tools.bench_kernels import timeit
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "bf16")
dev = eng.device
for be in (32, 16, 2):
    for side, c in ((64, 320), (32, 640), (16, 1280)):
        m = be * side * side
        x = torch.randn(m, c, device=dev).to(eng.tdt)
        w = eng._pack_mat(torch.randn(8 * c, c, device=dev) * c ** -0.5, geglu=True)
        b = torch.randn(8 * c, device=dev)
        fl = 2.0 * m * 8 * c * c
        res = []
        for tile in (2, 9, 19, 7, 17, 4, 14, 42, 0):
            def run():
                eng.arena.reset()
                eng.gemm([(x, c, 1, 1, 1, 0)], w, 8 * c, m, 1, 1, bias=b, geglu=True, tile=tile)
            t = timeit(run)
            res.append(f"t{tile}: {fl / t / 1e12:6.0f}")
        def run2():
            eng.arena.reset()
            eng.gemm([(x, c, 1, 1, 1, 0)], w, 8 * c, m, 1, 1, bias=b, tile=2)
        t = timeit(run2)
        print(f"B_eff={be:3d} geglu C={c:5d} M={m:7d}: TF/s " + " ".join(res) + f" | same GEMM without GEGLU (t2): {fl / t / 1e12:6.0f}")
