#!/usr/bin/env python3
"""Experiment: does running TWO half-batch sampling graphs concurrently on two HIP streams (two pipelines, two host threads) beat
one full-batch graph?  (Kernels bound by different units — MFMA convs, VALU-bound attention, HBM-bound norms — could overlap.)
Usage: python tools/bench_dual.py [B_total=64] [steps=1]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from faceposegenerator_amd import spec as S, weights as W
from faceposegenerator_amd.pipeline import StableDiffusionPipeline

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ucfg, vcfg = S.SD21_UNET, S.SD21_VAE


def make(batch, seed):
    pipe = StableDiffusionPipeline.from_synthetic(ucfg, vcfg, seed=1234, torch_dtype="f16").to(dev)
    pipe.load_lora_weights(W.synth_lora(ucfg, seed=1))
    pipe.use_graph = True
    g = torch.Generator().manual_seed(seed)
    pe = torch.randn(batch, 77, ucfg.cross_attention_dim, generator=g).to(dev)
    ne = torch.randn(batch, 77, ucfg.cross_attention_dim, generator=g).to(dev)
    noise = pipe.prepare_noise(batch, 30, 512, 512, torch.Generator().manual_seed(seed)).to(dev)

    def step():
        return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=30, guidance_scale=5.0, height=512, width=512,
                    output_type="uint8", noise=noise).images
    return step


def run_single(step, n):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


one = make(B, 1)
t1 = run_single(one, steps)
print(f"one graph of batch {B}: {B * steps / t1:.3f} images/s", flush=True)
del one
torch.cuda.empty_cache()

halves = [make(B // 2, 1), make(B // 2, 2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def worker(i, n):
    with torch.cuda.stream(streams[i]):
        for _ in range(n):
            halves[i]()
        streams[i].synchronize()


for i in range(2):                      # warm-up (eager pass + capture) one after the other
    worker(i, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(i, steps)) for i in range(2)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
t2 = time.perf_counter() - t0
print(f"two concurrent graphs of batch {B // 2} on two streams: {B * steps / t2:.3f} images/s", flush=True)
th0 = time.perf_counter()
worker(0, steps); worker(1, steps)
torch.cuda.synchronize()
t3 = time.perf_counter() - th0
print(f"the same two graphs one after the other: {B * steps / t3:.3f} images/s", flush=True)
