#!/usr/bin/env python3
"""GroupNorm+SiLU -> conv3x3 / proj_in at B_eff 2 (cold weights rotated inside a HIP graph): idb_groupnorm(partials from the producer) +
idb_gemm against ONE idb_gemm with gn_in_* (normalizer waves).  us per layer incl. the split-K reduce."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
beff = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "f16")
dev = eng.device
shapes = [(64, 320, 320, 9, True), (64, 640, 320, 9, True), (32, 640, 640, 9, True), (32, 1280, 640, 9, True), (16, 1280, 1280, 9, True),
          (16, 2560, 1280, 9, True), (8, 1280, 1280, 9, True), (8, 2560, 1280, 9, True), (64, 320, 320, 1, False), (32, 640, 640, 1, False),
          (16, 1280, 1280, 1, False), (8, 1280, 1280, 1, False)]
G = 32
for (h, cin, cout, taps, silu) in shapes:
    k = cin * taps
    nbuf = max(2, min(16, int(400e6 // (cout * k * 2))))
    pack = (lambda: eng._pack_conv(torch.randn(cout, cin, 3, 3, device=dev) * k ** -0.5)) if taps == 9 else (lambda: eng._pack_mat(torch.randn(cout, cin, device=dev) * k ** -0.5))
    ws = [eng.tile_weight(pack()) for _ in range(nbuf)]
    x = torch.randn(beff * h * h, cin, device=dev).to(eng.tdt)
    gamma, beta, bias = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1, torch.randn(cout, device=dev)
    # real partials: per 64-row chunk and group {sum, sum of squares}
    xf = x.float().view(beff, max(1, h * h // 64), min(64, h * h), G, cin // G)
    part = torch.stack([xf.sum(dim=(2, 4)), (xf * xf).sum(dim=(2, 4))], dim=-1).contiguous()
    chunks = max(1, h * h // 64)
    fus = eng.fuses_groupnorm([(cin, taps)], ws[0], cout, beff, h, h, G, 1)
    res = {}
    for name in ("unfused", "fused"):
        if name == "fused" and not fus:
            continue
        def run(i):
            eng.arena.reset()
            if name == "unfused":
                xn = eng.arena.alloc((beff * h * h, cin), eng.tdt)
                from faceposegenerator_amd import _lib as L
                L.check(eng.lib.idb_groupnorm(x.data_ptr(), cin, None, 0, beff, h * h, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), int(silu), xn.data_ptr(), eng.dt,
                                              eng._gn_ws.data_ptr(), eng._gn_ws.numel(), None, 0, part.data_ptr(), chunks, torch.cuda.current_stream().cuda_stream), "gn")
                return eng.gemm([(xn, cin, taps, h, h, 0)], ws[i % nbuf], cout, beff, h, h, bias=bias)
            return eng.gemm([(x, cin, taps, h, h, 0)], ws[i % nbuf], cout, beff, h, h, bias=bias, gn_in=(part, chunks, G, 1e-5, gamma, beta, silu, 1))
        for i in range(nbuf): o = run(i)
        torch.cuda.synchronize()
        if name == "unfused": ref = o.clone()
        else: same = torch.equal(o, ref)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nbuf): run(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (2 * nbuf) * 1e3)
        res[name] = best
    print(f"{h:2d}x{h:<2d} {cin:4d}->{cout:4d} taps {taps} silu {int(silu)}: unfused {res['unfused']:6.1f} us" +
          (f"   fused {res['fused']:6.1f} us  x{res['unfused'] / res['fused']:.2f}  bit-identical {same}" if fus else "   (plan cannot fuse)"), flush=True)
    del ws
