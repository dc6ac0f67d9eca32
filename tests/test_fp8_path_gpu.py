"""The fp8 path of BASELINE configs[4] (SD-2.1 768x768 v-prediction, "fp8 MFMA weight path"): `torch_dtype="fp8"` runs the 3x3 convs of
the UNet's ResnetBlock2Ds on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 activations from GroupNorm+SiLU with one fixed scale, e4m3 weights
with a scale per output channel), the rest of the graph in f16.

Stated tolerance (one CFG UNet forward, eps rel-RMS against the fp32 oracle): **<= 7.7e-2** = 1.5x the measured 5.1e-2 (full size,
96x96 latents, LoRA; reduced graph 5.6e-2).  That is the price of 3-bit mantissas on the conv operands of this synthetic-weight
network, not of the kernels: the oracle re-run on the CPU with EXACTLY the engine's quantisation emulated (oracle.fp8_quantize /
fp8_weights / ROUND_CONV_IN + the f16 rounding hook) is at the same distance from fp32 (5.10e-2 vs the engine's 5.12e-2; the test
requires agreement within 10 %).  Engine and emulation cannot agree element by element the way the f16 path does: an f16-level
difference upstream (1e-3) moves ~2 % of the e4m3 roundings by a whole 6 % step, so their mutual distance is 2.2e-2 = 0.44x the class
error (bounded at 0.6x)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _emulated(O, usd, ucfg, x, t, ctx, lora=None):
    wsd = {k: (v.half().float() if (v.ndim >= 2 and not k.startswith(("conv_in.", "time_embedding.")) and ".time_emb_proj." not in k) else v)
           for k, v in O.fp8_weights(usd).items()}
    for k, v in O.fp8_weights(usd).items():                       # the e4m3 weights are NOT re-rounded to f16 (exact in f16 anyway)
        if ".resnets." in k and k.endswith((".conv1.weight", ".conv2.weight")):
            wsd[k] = v
    O.ROUND = lambda kind, z: z.half().float()
    O.ROUND_CONV_IN = O.fp8_quantize                                  # (tensor, the layer's scale): oracle.fp8_act_scale = engine.fp8_act_scale
    try:
        with torch.no_grad():
            return O.unet_forward(wsd, ucfg, x, t, ctx, lora)
    finally:
        O.ROUND, O.ROUND_CONV_IN = None, None


def test_fp8_unet_forward_reduced_graph(lib):
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from oracle import sd21_oracle as O
    ucfg, vcfg = S.TINY_UNET, S.TINY_VAE
    usd, vsd = W.synth_unet(ucfg, 7), W.synth_vae(vcfg, 8)
    pipe = StableDiffusionPipeline(ucfg, vcfg, usd, vsd, torch_dtype="fp8").to(DEV)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 16, 16, generator=g)
    ctx = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g)
    got = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    with torch.no_grad():
        ref = O.unet_forward(usd, ucfg, x, 501, ctx)
    emu = _emulated(O, usd, ucfg, x, 501, ctx)
    e_emu = ((got - emu).norm() / emu.norm()).item()
    e_ref = ((got - ref).norm() / ref.norm()).item()
    e_cls = ((emu - ref).norm() / ref.norm()).item()
    print(f"[fp8] reduced-graph UNet forward: vs emulated-quantisation oracle {e_emu:.3e}; vs fp32 oracle {e_ref:.3e} (emulated oracle vs fp32: {e_cls:.3e})")
    assert e_ref < 8.4e-2 and abs(e_ref / e_cls - 1.0) < 0.1 and e_emu < 0.6 * e_cls      # measured 5.56e-2 / 0.98 / 0.44
    # a 2-step sampler call through the pipeline API (v-prediction scheduler mode of configs[4]) runs and stays finite
    from faceposegenerator_amd.scheduler import DDPMScheduler
    pipe.scheduler = DDPMScheduler(S.SchedulerConfig(prediction_type="v_prediction"))
    pe, ne = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g), torch.randn(1, 77, ucfg.cross_attention_dim, generator=g)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2, guidance_scale=5.0, height=128, width=128, output_type="np",
               generator=torch.Generator().manual_seed(3)).images
    assert out.shape == (1, 128, 128, 3) and (out >= 0).all() and (out <= 1).all()


def test_fp8_unet_forward_full_size_768(lib):
    """BASELINE configs[4] geometry: full SD-2.1 graph, 96x96 latents, rank-4 LoRA, one CFG forward; the oracle (fp32 and
    emulated-quantisation) runs on the host cores of the GPU box."""
    import os
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from oracle import sd21_oracle as O
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ucfg = S.SD21_UNET
    usd, vsd = W.synth_unet(ucfg, 1234), W.synth_vae(S.SD21_VAE, 1235)
    lora_raw = W.synth_lora(ucfg, 1)
    merged = O.merge_lora(usd, O.normalize_lora_keys(lora_raw))
    pipe = StableDiffusionPipeline(ucfg, S.SD21_VAE, usd, vsd, torch_dtype="fp8").to(DEV)
    pipe.load_lora_weights(lora_raw)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 4, 96, 96, generator=g)
    ctx = torch.randn(2, 77, 1024, generator=g)
    xin = torch.cat([x, x])
    got = pipe.unet(xin.to(DEV), 958, ctx.to(DEV), return_dict=False)[0].cpu()
    with torch.no_grad():
        ref = O.unet_forward(merged, ucfg, xin, 958, ctx)
    emu = _emulated(O, merged, ucfg, xin, 958, ctx)
    e_emu = ((got - emu).norm() / emu.norm()).item()
    e_ref = ((got - ref).norm() / ref.norm()).item()
    e_cls = ((emu - ref).norm() / ref.norm()).item()
    print(f"[fp8] full-size 96x96 CFG forward: vs emulated-quantisation oracle {e_emu:.3e}; vs fp32 oracle {e_ref:.3e} (emulated oracle vs fp32: {e_cls:.3e})")
    assert e_ref < 7.7e-2 and abs(e_ref / e_cls - 1.0) < 0.1 and e_emu < 0.6 * e_cls        # measured 5.12e-2 / 1.003 / 0.44
    del pipe
    torch.cuda.empty_cache()


def _psnr(a, b):
    import math
    import numpy as np
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d ** 2).mean())
    return 10 * math.log10(255.0 ** 2 / mse) if mse > 0 else float("inf")


def test_fp8_ten_step_vpred_trajectory_768(lib):
    """BASELINE configs[4] on the calibrated network: 96x96 latents, v-prediction, LoRA, 10 DDPM steps, CFG 5 — the fp8 engine against
    the committed fp32 oracle trajectory (tests/golden/sd21_config4_vpred.npz), beside the distance of the emulated-quantisation
    oracle stored in the same fixture: the engine must sit in that error class over the whole trajectory, not just one forward."""
    import os
    import numpy as np
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from faceposegenerator_amd.scheduler import DDPMScheduler
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sd21_config4_vpred.npz"))
    useed, vseed, lseed, batch, side, steps, eseed, nseed, calibrated = gold["meta"].tolist()
    import dataclasses
    ucfg = dataclasses.replace(S.SD21_UNET, prediction_type="v_prediction")
    usd, vsd = W.synth_unet(S.SD21_UNET, useed, calibrated=bool(calibrated)), W.synth_vae(S.SD21_VAE, vseed)
    assert np.allclose([float(usd[n].double().sum()) for n in ("conv_in.weight", "mid_block.resnets.0.conv1.weight",
                        "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_k.weight", "conv_out.weight", "conv_in.weight")],
                       gold["unet_fingerprint"], rtol=0, atol=1e-7)
    g = torch.Generator().manual_seed(eseed)
    pe, ne = torch.randn(batch, 77, 1024, generator=g), torch.randn(batch, 77, 1024, generator=g)
    gen = torch.Generator().manual_seed(nseed)
    noise = torch.stack([torch.randn((batch, 4, side, side), generator=gen, dtype=torch.float32) for _ in range(steps + 1)])
    assert np.array_equal(noise.flatten()[:4].numpy(), gold["noise_first4"])
    ref = gold["latents_per_step"]
    res = {}
    for dtype in ("fp8", "f16"):
        pipe = StableDiffusionPipeline(ucfg, S.SD21_VAE, usd, vsd, scheduler=DDPMScheduler(S.SchedulerConfig(prediction_type="v_prediction")),
                                       torch_dtype=dtype).to(DEV)
        pipe.load_lora_weights(W.synth_lora(S.SD21_UNET, lseed))
        eng = pipe._engine()
        sch = pipe.scheduler
        sch.set_timesteps(steps)
        ts = sch.timesteps.tolist()
        assert ts == gold["timesteps"].tolist()
        coefs = torch.tensor([list(sch.step_coefficients(t)) + [5.0] for t in ts], dtype=torch.float32)
        trace = []
        lat = eng.sample(pe, ne, noise.to(DEV), ts, coefs.to(DEV), vpred=True, use_graph=False, trace=trace)
        _, u8 = eng.decode_images(lat, chunk=1)
        per_step = []
        for i in range(steps):
            d = trace[i][1].cpu().numpy().astype(np.float64) - ref[i]
            per_step.append((float(np.sqrt((d ** 2).mean()) / ref[i].std()), float(np.abs(d).max())))
        res[dtype] = (per_step, _psnr(u8.cpu().numpy(), gold["image_u8"]))
        print(f"[{dtype}] configs[4] 96x96 v-prediction, {steps} steps: latents rel-RMS / max-abs per step vs fp32 oracle: " +
              ", ".join(f"{r:.2e}/{m:.2e}" for r, m in per_step) + f"; decoded 768x768 image PSNR {res[dtype][1]:.2f} dB")
        del pipe, eng
        torch.cuda.empty_cache()
    cls = [(float(r), float(m)) for r, m in gold["emulated_fp8_error_per_step"]]
    psnr_cls = float(gold["emulated_fp8_image_psnr"])
    print("[emulated-quantisation oracle] the same: " + ", ".join(f"{r:.2e}/{m:.2e}" for r, m in cls) + f"; image PSNR {psnr_cls:.2f} dB")
    (r8, m8), (rc, mc) = res["fp8"][0][-1], cls[-1]
    # the engine is in the emulated oracle's class at the END of the trajectory (within 1.5x either way) and its image within 3 dB
    assert r8 < 1.5 * rc and m8 < 2.0 * mc, (r8, rc, m8, mc)
    assert res["fp8"][1] > psnr_cls - 3.0
    assert res["f16"][0][-1][0] < 0.25 * r8            # and the f16 path is far inside it (what fp8 costs is visible, not hidden)


def test_fp8_per_block_error_attribution(lib):
    """One CFG forward of the calibrated configs[1] network at 64x64: every block's output, fp8 engine vs fp32 oracle, beside the oracle
    with the engine's quantisation emulated (e4m3 resnet-conv operands with the per-layer scales, f16 storage elsewhere)."""
    import os
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from oracle import sd21_oracle as O
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ucfg = S.SD21_UNET
    usd, vsd = W.synth_unet(ucfg, 1234, calibrated=True), W.synth_vae(S.SD21_VAE, 1235)
    lora_raw = W.synth_lora(ucfg, 1)
    merged = O.merge_lora(usd, O.normalize_lora_keys(lora_raw))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 64, 64, generator=g)
    ctx = torch.randn(2, 77, 1024, generator=g)
    xin = torch.cat([x, x])
    with torch.no_grad():
        ref = {"__blocks__": True}
        eps_ref = O.unet_forward(merged, ucfg, xin, 958, ctx, taps=ref)
    q = O.fp8_weights(merged)
    wsd = {k: (v.half().float() if (v.ndim >= 2 and not k.startswith(("conv_in.", "time_embedding.")) and ".time_emb_proj." not in k) else v) for k, v in q.items()}
    for k, v in q.items():
        if ".resnets." in k and k.endswith((".conv1.weight", ".conv2.weight")):
            wsd[k] = v
    emu = {"__blocks__": True}
    O.ROUND, O.ROUND_CONV_IN = (lambda kind, z: z.half().float()), O.fp8_quantize
    try:
        with torch.no_grad():
            eps_emu = O.unet_forward(wsd, ucfg, xin, 958, ctx, taps=emu)
    finally:
        O.ROUND, O.ROUND_CONV_IN = None, None
    pipe = StableDiffusionPipeline(ucfg, S.SD21_VAE, usd, vsd, torch_dtype="fp8").to(DEV)
    pipe.load_lora_weights(lora_raw)
    eng = pipe._engine()
    eng.taps = {}
    eps = pipe.unet(xin.to(DEV), 958, ctx.to(DEV), return_dict=False)[0].cpu()
    taps, eng.taps = eng.taps, None
    print("[fp8] step-0 CFG forward, rel-RMS error per block output: fp8 engine | emulated-quantisation oracle")
    bad = []
    for name in [k for k in ref if k not in ("__blocks__", "temb", "down_out", "mid_out", "up_out")]:
        r = ref[name]
        got = taps[name].cpu().reshape(r.shape[0], r.shape[2], r.shape[3], r.shape[1]).permute(0, 3, 1, 2)
        e_gpu, e_emu = float((got - r).norm() / r.norm()), float((emu[name] - r).norm() / r.norm())
        print(f"    {name:45s} {e_gpu:.3e} | {e_emu:.3e}")
        if e_gpu > 1.3 * e_emu + 1e-4:
            bad.append((name, e_gpu, e_emu))
    e_gpu, e_emu = float((eps - eps_ref).norm() / eps_ref.norm()), float((eps_emu - eps_ref).norm() / eps_ref.norm())
    print(f"    {'eps (conv_out)':45s} {e_gpu:.3e} | {e_emu:.3e}")
    assert not bad, f"blocks outside 1.3x the emulated fp8 error class: {bad}"
    assert e_gpu < 1.3 * e_emu
    del pipe, eng
    torch.cuda.empty_cache()
