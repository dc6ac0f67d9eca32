#!/usr/bin/env python3
"""fp8 (v_mfma_scale_f32_16x16x128_f8f6f4) vs f16 implicit GEMM on the large-batch layer shapes: TFLOP/s (algorithmic FLOPs of the
UNPADDED problem; the fp8 kernel pads C = 320 to 384 per tap).  Usage: python tools/bench_fp8.py [B_eff=128] [side=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "f16")
dev = eng.device
be = int(sys.argv[1]) if len(sys.argv) > 1 else 128
side0 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
# (side divisor, cin, cout, taps)
shapes = [(1, 320, 320, 9), (1, 640, 320, 9), (2, 640, 640, 9), (2, 1280, 640, 9), (4, 1280, 1280, 9), (8, 1280, 1280, 9),
          (1, 320, 320, 1), (1, 320, 960, 1), (1, 1280, 320, 1), (2, 640, 640, 1), (4, 1280, 1280, 1), (4, 5120, 1280, 1)]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (div, cin, cout, taps) in shapes:
    side = side0 // div
    m = be * side * side
    x = torch.randn(m, cin, device=dev).to(eng.tdt)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5 if taps == 9 else torch.randn(cout, cin, device=dev) * cin ** -0.5
    bias = torch.randn(cout, device=dev)
    out = torch.empty(m, cout, dtype=eng.tdt, device=dev)
    wp = eng._pack_conv(w) if taps == 9 else eng._pack_mat(w)
    x8 = eng.quantize_fp8(x, 0.02)
    w8, ws = eng.pack_weight_fp8(w)
    fl = 2.0 * m * cout * cin * taps
    if taps == 9:
        t16 = timed(lambda: eng.gemm([(x, cin, 9, side, side, 0)], wp, cout, be, side, side, bias=bias, out=out))
        t8 = timed(lambda: eng.gemm_fp8(x8, 0.02, cin, 9, side, side, w8, ws, cout, be, side, side, bias=bias, out=out))
    else:
        t16 = timed(lambda: eng.gemm([(x, cin, 1, 1, 1, 0)], wp, cout, m, 1, 1, bias=bias, out=out))
        t8 = timed(lambda: eng.gemm_fp8(x8, 0.02, cin, 1, 1, 1, w8, ws, cout, m, 1, 1, bias=bias, out=out))
    tq = timed(lambda: eng.quantize_fp8(x, 0.02))
    print(f"{'conv3x3' if taps == 9 else 'linear '} {cin:5d}->{cout:5d} @{side:3d} M={m:7d} ({fl / 1e9:7.1f} GF): f16 {t16:8.1f} us {fl / t16 / 1e6:6.0f} TF/s | "
          f"fp8 {t8:8.1f} us {fl / t8 / 1e6:6.0f} TF/s ({t16 / t8:4.2f}x) | quantise pass {tq:6.1f} us", flush=True)
    del x, w, out, wp, x8, w8
