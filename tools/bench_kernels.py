#!/usr/bin/env python3
"""Kernel microbenchmarks on one MI355X: implicit-GEMM tiles and attention on the layer shapes of the
SD-2.1 UNet (HIP-event timed, random data).  Usage: python tools/bench_kernels.py [batch_eff ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    beffs = [int(a) for a in sys.argv[1:]] or [2, 32]
    eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "bf16")
    dev = eng.device
    for be in beffs:
        print(f"==== B_eff = {be}")
        # (name, side, cin, cout, taps)
        shapes = [("conv 320->320 @64", 64, 320, 320, 9), ("conv 640->640 @32", 32, 640, 640, 9),
                  ("conv 1280->1280 @16", 16, 1280, 1280, 9), ("conv 1280->1280 @8", 8, 1280, 1280, 9),
                  ("lin qkv 320->960 @64", 64, 320, 960, 1), ("lin ff2 1280->320 @64", 64, 1280, 320, 1),
                  ("lin ff2 5120->1280 @16", 16, 5120, 1280, 1)]
        for name, side, cin, cout, taps in shapes:
            x = torch.randn(be * side * side, cin, device=dev).to(eng.tdt)
            w = (torch.randn(cout, taps * cin, device=dev) * (taps * cin) ** -0.5).to(eng.tdt)
            m = be * side * side
            fl = 2.0 * m * cout * taps * cin
            res = []
            for tile in ((6, 16, 3, 8, 18, 0) if os.environ.get("IDB_KB_SHORT") else (1, 2, 3, 4, 11, 12, 13, 14, 0)):
                if cout % 160 and tile < 100 and tile % 10 in (1, 3, 6, 8):
                    continue
                if tile // 10 == 4 and taps == 9:
                    continue
                flags = 2 if 100 <= tile < 200 else (32 if tile >= 200 else 0)   # 100: stores skipped; 200: stores aliased to one tile
                tl = 1 if tile >= 100 else tile
                if tl // 10 == 4 and taps == 9:
                    continue
                def run():
                    eng.arena.reset()
                    if taps == 9:
                        eng.gemm([(x, cin, 9, side, side, 0)], w, cout, be, side, side, tile=tl, flags=flags)
                    else:
                        eng.gemm([(x, cin, 1, 1, 1, 0)], w, cout, m, 1, 1, tile=tl, flags=flags)
                t = timeit(run)
                res.append(f"t{tile}:{fl / t / 1e12:6.0f}")
            print(f"{name:26s} M={m:7d} {fl / 1e9:8.1f} GF | TF/s " + " ".join(res))
        for heads, n in ((5, 4096), (10, 1024), (20, 256)):
            c = heads * 64
            qkv = torch.randn(be * n, 3 * c, device=dev).to(eng.tdt)
            p = qkv.data_ptr()
            def run():
                eng.arena.reset()
                eng.attention(qkv, 3 * c, p + 2 * c, p + 4 * c, 3 * c, be, heads, n, n, n)
            t = timeit(run)
            fl = 4.0 * be * heads * n * n * 64
            print(f"self-attn h={heads} n={n}: {t * 1e6:8.1f} us {fl / t / 1e12:6.0f} TF/s")
        for hw, c in ((4096, 320), (1024, 640), (256, 1280), (64, 2560)):
            x = torch.randn(be * hw, c, device=dev).to(eng.tdt)
            g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
            def run():
                eng.arena.reset()
                eng.groupnorm(x, c, None, 0, be, hw, g, b, 1e-5, True, 32)
            t = timeit(run)
            print(f"groupnorm hw={hw} c={c}: {t * 1e6:8.1f} us  {be * hw * c * 4 / t / 1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
