#!/usr/bin/env python3
"""Per-launch floor inside a HIP graph: chains of dependent small kernels (LayerNorm on R rows x C), time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "bf16")
dev = eng.device
N = 200
for rows, c in ((4, 320), (256, 320), (2048, 320), (8192, 320), (32768, 320), (2048, 640), (512, 1280)):
    xs = [torch.randn(rows, c, device=dev).to(eng.tdt) for _ in range(2)]
    g_, b_ = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    def chain():
        for i in range(N):
            eng.arena.reset()
            from faceposegenerator_amd import _lib as L
            L.check(eng.lib.idb_layernorm(xs[i & 1].data_ptr(), xs[(i + 1) & 1].data_ptr(), rows, c, 1e-5, g_.data_ptr(), b_.data_ptr(), eng.dt,
                                          torch.cuda.current_stream().cuda_stream))
    chain(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"layernorm rows={rows:6d} c={c:5d} ({rows * c * 4 / 1e6:6.2f} MB moved): {e0.elapsed_time(e1) / (2 * N) * 1e3:6.2f} us per launch in a graph chain", flush=True)
