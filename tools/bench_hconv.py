#!/usr/bin/env python3
"""GroupNorm+SiLU -> conv3x3, two ways, cold operands (rotating buffer sets > MALL) inside a HIP graph, us per layer:
  old: idb_groupnorm (apply pass only, statistics given) + idb_gemm (+ split-K reduce)
  new: idb_hconv (+ split-K reduce), the normalisation applied to the conv's input patch in LDS
Usage: python tools/bench_hconv.py [B_eff] [split_k list for the new kernel, e.g. 0,2,4,8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
dt = os.environ.get("IDB_DTYPE", "f16")
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", dt)
dev = eng.device
be = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
only = os.environ.get("IDB_SHAPES")
shapes = [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640), (32, 320, 640), (16, 1280, 1280), (16, 2560, 1280),
          (8, 1280, 1280), (8, 2560, 1280), (16, 640, 1280), (64, 512, 512)]
if only:
    shapes = shapes[:int(only)]
G = 32


def timed(run, nbuf):
    for i in range(nbuf): run(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nbuf): run(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * nbuf) * 1e3


for (side, cin, cout) in shapes:
    m, k = be * side * side, 9 * cin
    per = 2 * (2 * m * cin + cout * k + m * cout)
    nbuf = max(3, min(48, int(600e6 // per)))
    xs = [torch.randn(m, cin, device=dev).to(eng.tdt) for _ in range(nbuf)]
    ws = [(torch.randn(cout, k, device=dev) * k ** -0.5).to(eng.tdt) for _ in range(nbuf)]
    gamma, beta = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    bias = torch.randn(cout, device=dev)
    parts = []
    for i in range(nbuf):
        p, ch = eng.gn_statistics(xs[i], cin, None, 0, be, side * side, G)
        parts.append((p.clone(), ch))
    eng.arena.reset()

    def run_old(i):
        eng.arena.reset()
        hw = side * side        # statistics as if they came with the tensor (the split-K reduce of its producer): apply pass only
        xs[i]._gn = (eng.arena.alloc((be * (hw // 64) * G * 2,), torch.float32), hw // 64, G) if hw % 64 == 0 and hw <= 4096 else None
        n1 = eng.groupnorm(xs[i], cin, None, 0, be, side * side, gamma, beta, 1e-5, True, G)
        eng.gemm([(n1, cin, 9, side, side, 0)], ws[i], cout, be, side, side, bias=bias, gn_stats=G)

    def run_new(i, sk):
        eng.arena.reset()
        gn = None if os.environ.get("IDB_HC_NOGN") else (parts[i][0], parts[i][1], G, 1e-5, gamma, beta, True)
        eng.hconv([(xs[i], cin, None, 0, 9)], ws[i], cout, be, side, side, gn=gn, bias=bias, gn_stats=G, split_k=sk)

    line = [f"old {timed(run_old, nbuf):6.1f}"]
    ok = eng.hconv_supported([(xs[0], cin, None, 0, 9)], ws[0], cout, be, side, side, G)
    if ok:
        for sk in sks:
            try:
                line.append(f"new(sk={sk}) {timed(lambda i: run_new(i, sk), nbuf):6.1f}")
            except Exception as e:
                line.append(f"new(sk={sk}) n/a")
    print(f"gn+conv {cin}->{cout} @{side} m={m:5d} k={k:5d} ({2.0 * m * cout * k / 1e9:5.1f} GF) us  " + "  ".join(line), flush=True)
    del xs, ws
