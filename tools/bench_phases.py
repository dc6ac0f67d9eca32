#!/usr/bin/env python3
"""Phase split of one pipeline call at batch B: sampling loop (HIP graph) vs VAE decode + postprocess."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceposegenerator_amd import spec as S, weights as W
from faceposegenerator_amd.pipeline import StableDiffusionPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
pipe = StableDiffusionPipeline.from_synthetic(S.SD21_UNET, S.SD21_VAE, seed=1234, torch_dtype="bf16").to(dev)
pipe.load_lora_weights(W.synth_lora(S.SD21_UNET, seed=1))
pipe.use_graph = True
eng = pipe._engine()
g = torch.Generator().manual_seed(1000)
pe, ne = torch.randn(B, 77, 1024, generator=g).to(dev), torch.randn(B, 77, 1024, generator=g).to(dev)
noise = pipe.prepare_noise(B, 30, 512, 512, torch.Generator().manual_seed(0)).to(dev)
def ev():
    return torch.cuda.Event(enable_timing=True)
for it in range(3):
    e = [ev() for _ in range(3)]
    t0 = time.perf_counter()
    e[0].record()
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=30, guidance_scale=5.0, height=512, width=512,
               output_type="latent", noise=noise)
    e[1].record()
    lat = out.images
    img01, u8 = eng.decode_images(lat, chunk=pipe.vae_chunk, want_u8=True)
    t1 = time.perf_counter()
    e[2].record()
    torch.cuda.synchronize()
    print(f"B={B} iter {it}: sampling loop {e[0].elapsed_time(e[1]):8.2f} ms   decode+postprocess {e[1].elapsed_time(e[2]):7.2f} ms (host enqueue of everything {1e3 * (t1 - t0):7.2f} ms)", flush=True)
