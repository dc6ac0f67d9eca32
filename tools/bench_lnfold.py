#!/usr/bin/env python3
"""Folded-LayerNorm consumer GEMMs (idb_gemm_desc.ln_*) against the same GEMM on pre-normalised rows, per tile variant, and the
producer GEMM with / without row_stats_out.  Usage: python tools/bench_lnfold.py [B_eff=128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S, _lib as L
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", os.environ.get("IDB_DTYPE", "f16"))
dev = eng.device
be = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timeit(run, reps=5):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (tok, k, n, geglu) in [(4096, 320, 960, 0), (4096, 320, 2560, 1), (1024, 640, 1920, 0), (1024, 640, 5120, 1), (256, 1280, 3840, 0)]:
    m = be * tok
    a = torch.randn(m, k, device=dev).to(eng.tdt)
    wp = (torch.randn(k, k, device=dev) * k ** -0.5).to(eng.tdt)
    res = torch.randn(m, k, device=dev).to(eng.tdt)
    bias_p = torch.randn(k, device=dev)
    h = torch.empty(m, k, dtype=eng.tdt, device=dev)
    eng.arena.reset()
    us_p0 = timeit(lambda: eng.gemm([(a, k, 1, 1, 1, 0)], wp, k, m, 1, 1, bias=bias_p, residual=res, out=h))
    hp = eng.gemm([(a, k, 1, 1, 1, 0)], wp, k, m, 1, 1, bias=bias_p, residual=res, out=h, row_stats=True)
    rs = hp._rs
    d_rs = rs[0]

    def prod():
        o = eng.gemm([(a, k, 1, 1, 1, 0)], wp, k, m, 1, 1, bias=bias_p, residual=res, out=h, row_stats=True)
        eng.arena.free(o._rs[0])
    us_p1 = timeit(prod)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(eng.tdt)
    u, v = torch.randn(n, device=dev), torch.randn(n, device=dev)
    no = n // 2 if geglu else n
    out = torch.empty(m, no, dtype=eng.tdt, device=dev)
    line = [f"producer {us_p0:7.1f} -> {us_p1:7.1f} us |"]
    for t in (8, 9, 18, 19):
        if (n % 160 or geglu) and t % 10 == 8:
            continue
        try:
            us0 = timeit(lambda: eng.gemm([(h, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=v, geglu=bool(geglu), tile=t, out=out))
            us1 = timeit(lambda: eng.gemm([(h, k, 1, 1, 1, 0)], w, n, m, 1, 1, geglu=bool(geglu), tile=t, out=out, ln=(d_rs, rs[1], u, v, 1e-5)))
        except L.IdbError as e:
            line.append(f"t{t}: {e}")
            continue
        line.append(f"t{t}: {us0:7.1f} -> {us1:7.1f} us ({us1 / us0 - 1:+.1%})")
    print(f"m={m:7d} k={k:5d} n={n:5d} geglu={geglu} | " + " ".join(line), flush=True)
    del a, wp, res, h, w, out
