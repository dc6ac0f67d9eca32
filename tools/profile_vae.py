#!/usr/bin/env python3
"""VAE decode + postprocess of N 64x64 latents (full-size SD-2.1 decoder, synthetic weights) a few times, for rocprofv3 --kernel-trace --stats.
Usage: python tools/profile_vae.py [images=16] [reps=3] [chunk=4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from faceposegenerator_amd import spec as S
from faceposegenerator_amd.pipeline import StableDiffusionPipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pipe = StableDiffusionPipeline.from_synthetic(S.TINY_UNET, S.SD21_VAE, seed=1234, torch_dtype=os.environ.get("IDB_DTYPE", "f16")).to("cuda:0")
eng = pipe._engine()
lat = torch.randn(n, 4, 64, 64, device="cuda:0")
eng.decode_images(lat, chunk=chunk, want_u8=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.decode_images(lat, chunk=chunk, want_u8=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"VAE decode + postprocess: {n} images, chunk {chunk}: {dt * 1e3:.2f} ms = {dt * 1e3 / n:.3f} ms per image "
      f"({2.0 * S.vae_decode_macs(S.SD21_VAE, 64) * n / dt / 1e12:.0f} TFLOP/s)")
