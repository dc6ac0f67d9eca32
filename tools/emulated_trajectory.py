#!/usr/bin/env python3
"""What does ANY implementation with 16-bit storage of weights and activations reach on BASELINE configs[1]?  The CPU oracle
re-runs the 30-step trajectory of tests/golden/sd21_config1.npz with the HIP engine's rounding points emulated
(oracle.ROUND: every stored activation and the weight matrices rounded to f16 / bf16, fp32 accumulation and fp32 latents,
exactly the engine's precision design — and, for f16, the precision class of the reference's own
`torch_dtype=torch.float16` CUDA run) and reports the distance to the plain fp32 trajectory.  CPU only, ~5 minutes per dtype.

  python tools/emulated_trajectory.py f16 [bf16] [f16:nores]      (":nores" = residual-stream tensors kept in fp32)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from faceposegenerator_amd import spec as S, weights as W
from oracle import sd21_oracle as O


def main():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sd21_config1.npz"))
    useed, vseed, lseed, batch, side, steps, eseed, nseed, calibrated = gold["meta"].tolist()
    torch.set_num_threads(os.cpu_count() or 8)
    usd = W.synth_unet(S.SD21_UNET, useed, calibrated=bool(calibrated))
    merged = O.merge_lora(usd, O.normalize_lora_keys(W.synth_lora(S.SD21_UNET, lseed)))
    g = torch.Generator().manual_seed(eseed)
    pe = torch.randn(batch, 77, 1024, generator=g)
    ne = torch.randn(batch, 77, 1024, generator=g)
    noise = O.draw_noise(torch.Generator().manual_seed(nseed), batch, steps, (side, side))
    ref = gold["latents_per_step"]
    for name in sys.argv[1:] or ["f16"]:
        nores = name.endswith(":nores")
        dt = torch.float16 if name.startswith("f16") else torch.bfloat16
        wsd = {k: (v.to(dt).float() if (v.ndim >= 2 and not k.startswith(("conv_in.", "time_embedding.")) and ".time_emb_proj." not in k)
                   else v) for k, v in merged.items()}
        O.ROUND = (lambda kind, z: z if kind == "res" else z.to(dt).float()) if nores else (lambda kind, z: z.to(dt).float())
        trace = []
        with torch.no_grad():
            lat = O.sample(wsd, S.SD21_UNET, pe, ne, noise, steps, 5.0, trace=trace)
        O.ROUND = None
        for i in (0, 4, 9, 14, 19, 24, 29):
            d = trace[i][2].numpy().astype(np.float64) - ref[i]
            print(f"  [{name}] step {i:2d}: latents rel-RMS {np.sqrt((d ** 2).mean()) / ref[i].std():.3e} max-abs {np.abs(d).max():.3e}", flush=True)
        d = lat.numpy().astype(np.float64) - gold["final_latents"]
        print(f"[{name}] emulated 16-bit-storage oracle vs fp32 oracle, final latents after {steps} steps: max-abs {np.abs(d).max():.4e} "
              f"rel-RMS {np.sqrt((d ** 2).mean()) / np.sqrt((gold['final_latents'].astype(np.float64) ** 2).mean()):.4e} "
              f"(|ref| std {gold['final_latents'].std():.2f})", flush=True)


if __name__ == "__main__":
    main()
