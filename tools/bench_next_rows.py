#!/usr/bin/env python3
"""The "next" rows either side of the sampler on one MI355X with the CPU oracle timed beside each on the box's
host cores: CLIP-H text encoder (f-1), MTCNN detect + align-and-crop on the sampler's uint8 output (f-3), VAE encode (f-4).
Synthetic weights of the published shapes; the oracle legs are bounded samples (seconds).  Prints one JSON object.
Usage: python tools/bench_next_rows.py [--no-cpu]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from faceposegenerator_amd import face_align as FA, mtcnn as M, spec as S, weights as W
from faceposegenerator_amd.engine import HipEngine
from faceposegenerator_amd.pipeline import StableDiffusionPipeline
from faceposegenerator_amd.text_encoder import ClipTextEncoder

DEV = "cuda:0"
cpu = "--no-cpu" not in sys.argv
out = {}


def gpu_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cpu_time(fn, reps=1):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


# ---- f-1: text encoder, the pipeline's 2 prompts (prompt + negative prompt) x 77 tokens
cfg = S.SD21_CLIP
sd = W.synth_clip(cfg, 99)
eng = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, DEV, "f16")
te = ClipTextEncoder(eng, cfg, sd)
ids = torch.randint(1, 49000, (2, 77), generator=torch.Generator().manual_seed(2))
ids[:, 0], ids[:, 30:] = cfg.bos_token_id, 0
ids[:, 29] = cfg.eos_token_id
t = gpu_time(lambda: te.encode(ids), 10)
row = {"workload": "CLIP-H text model, 2 prompts x 77 tokens, f16", "gpu_ms": round(t * 1e3, 3), "prompts_per_s": round(2 / t, 1)}
if cpu:
    from oracle import clip_oracle as CO
    with torch.no_grad():
        tc = cpu_time(lambda: CO.clip_text_forward(sd, cfg, ids))
    row.update(cpu_oracle_ms=round(tc * 1e3, 1), cpu_threads=torch.get_num_threads())
out["text_encoder"] = row
del te, eng, sd
torch.cuda.empty_cache()

# ---- f-4: VAE encode of 512x512 images (train_ID-Booth.py:1001)
vcfg = S.SD21_VAE
esd = W.synth_vae_encoder(vcfg, 4322)
pipe = StableDiffusionPipeline(S.TINY_UNET, vcfg, W.synth_unet(S.TINY_UNET, 7), W.synth_vae(vcfg, 1235), torch_dtype="f16").to(DEV)
pipe.set_vae_encoder_weights(esd)
x = (torch.rand(8, 3, 512, 512, generator=torch.Generator().manual_seed(9)) * 2 - 1).to(DEV)
t = gpu_time(lambda: pipe.vae.encode(x).latent_dist.mean, 3)
row = {"workload": "AutoencoderKL.encode, 8 images 512x512, f16", "gpu_ms": round(t * 1e3, 2), "images_per_s": round(8 / t, 1)}
if cpu:
    from oracle import sd21_oracle as O
    with torch.no_grad():
        tc = cpu_time(lambda: O.vae_encode(esd, vcfg, x[:1].cpu()))
    row.update(cpu_oracle_ms_per_image=round(tc * 1e3, 1), cpu_threads=torch.get_num_threads())
out["vae_encode"] = row
del pipe, esd
torch.cuda.empty_cache()

# ---- f-3: MTCNN detect (landmarks) + norm_crop on a batch of 512x512 uint8 images (the sampler's output format)
w = M.synth_weights(5)
det = M.MTCNN(select_largest=True, post_process=False, device=DEV, weights=w)
g = torch.Generator().manual_seed(3)
base = torch.rand(16, 3, 64, 64, generator=g)
imgs = (F.interpolate(base, size=(512, 512), mode="bilinear") * 255).permute(0, 2, 3, 1).to(torch.uint8).contiguous()
imgs_d = imgs.to(DEV)
res = det.detect(imgs_d, landmarks=True)
nfaces = sum(0 if b is None else len(b) for b in res[0])
t = gpu_time(lambda: det.detect(imgs_d, landmarks=True), 3)
row = {"workload": "MTCNN.detect(landmarks=True), 16 images 512x512 uint8, synthetic P/R/O-Net weights", "gpu_ms": round(t * 1e3, 1),
       "images_per_s": round(16 / t, 1), "faces_found": int(nfaces)}
if cpu:
    from oracle import mtcnn_oracle as MO
    tc = cpu_time(lambda: MO.detect_face(imgs[:2], w))
    row.update(cpu_oracle_ms_per_image=round(tc * 1e3 / 2, 1), cpu_threads=torch.get_num_threads())
out["mtcnn_detect"] = row
lms = np.stack([FA.ARCFACE_TEMPLATE.astype(np.float64) * 3.0 + 60.0 + i for i in range(16)])
t = gpu_time(lambda: FA.norm_crop(imgs_d, lms), 20)
row = {"workload": "norm_crop: estimate_norm + warpAffine to 112x112, 16 faces", "gpu_ms": round(t * 1e3, 3), "faces_per_s": round(16 / t, 0)}
if cpu:
    from oracle import face_align_oracle as FO
    im_np = imgs.numpy()
    tc = cpu_time(lambda: [FO.norm_crop(im_np[i], lms[i]) for i in range(2)])
    row.update(cpu_oracle_ms_per_face=round(tc * 1e3 / 2, 1))
out["norm_crop"] = row
print(json.dumps(out))
