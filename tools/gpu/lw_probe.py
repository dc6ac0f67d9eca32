"""One small loader-wave GEMM launch (first contact with the 12-wave kernel): must finish in seconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "f16")
a = torch.randn(256, 320, device="cuda").half()
w = (torch.randn(320, 320, device="cuda") * 320 ** -0.5).half()
for tile in (16, 56, 66, 76, 58, 54):
    out = eng.gemm([(a, 320, 1, 1, 1, 0)], w, 320, 256, 1, 1, tile=tile)
    torch.cuda.synchronize()
    print(tile, float((out.float() - a.float() @ w.float().t()).abs().max()), flush=True)
print("probe ok")
