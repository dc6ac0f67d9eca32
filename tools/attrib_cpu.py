#!/usr/bin/env python3
"""Where does the operand-dtype error of one CFG UNet forward come from?  CPU-only study with the oracle's rounding hook
(oracle/sd21_oracle.py: ROUND): the fp32 restatement is re-run with the HIP engine's rounding points emulated class by
class — weights ("w"), stored non-residual activations ("act"), residual-stream tensors ("res") — and compared with the
plain fp32 run.  Decides whether an fp32 residual stream is worth its HBM bytes.

  python tools/attrib_cpu.py [tiny|full] [f16|bf16 ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from faceposegenerator_amd import spec as S, weights as W
from oracle import sd21_oracle as O


def rounder(dt, kinds):
    def f(kind, t):
        return t.to(dt).float() if kind in kinds else t
    return f


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    dts = [a for a in sys.argv[2:]] or ["f16", "bf16"]
    ucfg = S.SD21_UNET if size == "full" else S.TINY_UNET
    side = 64 if size == "full" else 16
    torch.set_num_threads(os.cpu_count() or 8)
    usd = W.synth_unet(ucfg, 1234)
    lora = O.normalize_lora_keys(W.synth_lora(ucfg, 1))
    usd = O.merge_lora(usd, lora)
    g = torch.Generator().manual_seed(2024)
    pe = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g)
    ne = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g)
    x = torch.randn(1, 4, side, side, generator=torch.Generator().manual_seed(0))
    xin, ctx = torch.cat([x, x]), torch.cat([ne, pe])
    t = 958

    def fwd(sd):
        with torch.no_grad():
            return O.unet_forward(sd, ucfg, xin, t, ctx)

    t0 = time.time()
    ref = fwd(usd)
    print(f"{size}: reference forward {time.time() - t0:.1f} s; eps std {ref.std():.4f}", flush=True)
    for name in dts:
        dt = torch.float16 if name == "f16" else torch.bfloat16
        # the engine keeps conv_in / time-embedding weights and all biases / norm affine parameters in fp32
        wsd = {k: (v.to(dt).float() if (v.ndim >= 2 and not k.startswith(("conv_in.", "time_embedding.")) and ".time_emb_proj." not in k) else v)
               for k, v in usd.items()}
        for label, sd, kinds in (("weights only", wsd, ()), ("act only", usd, ("act",)), ("res only", usd, ("res",)),
                                 ("weights + act (fp32 residual stream)", wsd, ("act",)),
                                 ("weights + act + res (= the engine)", wsd, ("act", "res"))):
            O.ROUND = rounder(dt, kinds) if kinds else None
            out = fwd(sd)
            O.ROUND = None
            d = out - ref
            print(f"  {name:5s} {label:40s} eps rel-RMS {d.norm() / ref.norm():.3e}  max-abs {d.abs().max():.3e}", flush=True)


if __name__ == "__main__":
    main()
