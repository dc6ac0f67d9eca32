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
