"""First contact with the fused GroupNorm kernel (16 waves: 8 MFMA + 4 loader + 4 normalizer): one small launch, must finish in seconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faceposegenerator_amd import spec as S
from faceposegenerator_amd.engine import HipEngine
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, "cuda:0", "f16")
b, h, cin, cout = 2, 16, 128, 160
x = torch.randn(b, h, h, cin, device="cuda").half()
w = eng.tile_weight(eng._pack_conv(torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5))
gamma, beta = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
for tile in (76, 56, 58, 74):
    part, chunks = eng.gn_statistics(x, cin, None, 0, b, h * h, 32)
    out = eng.gemm([(x, cin, 9, h, h, 0)], w, cout, b, h, h, tile=tile, gn_in=(part, chunks, 32, 1e-5, gamma, beta, True, 1))
    torch.cuda.synchronize()
    xn = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), 32, None, None, 1e-5)).half()
    print(tile, None if out is None else float(out.float().abs().mean()), flush=True)
print("probe ok")
